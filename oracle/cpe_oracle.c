/*
 * cpe_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, fp64, single thread) of the
 * reference's full-trajectory-estimation hot path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product (libcpe.so, HIP) never does.
 *
 * Parity status of this oracle (see DESIGN.md "Oracle"):
 *   - projection, robust loss, uncertainty tables, relative angles: PINNED by golden vectors generated
 *     from the reference's own functions (tests/golden/misc_golden.npz, tools/gen_golden.py).
 *   - link parameters: PINNED (cheetah_params.py values exported by tools/export_skeleton_params.py).
 *   - forward kinematics / marker model / joint equalities: restated from SURVEY.md Appendix A (the
 *     FK library `physical_education` is absent from /root/reference and the stored .robot / fte.pickle
 *     files are refused by the safe loaders): pinned only through tests/golden/fk_csv_pin (2D files).
 *   - argmin of the full NLP vs the reference's IPOPT solutions: PARITY UNPINNED (IPOPT/Pyomo absent).
 *
 * What each function follows in the reference is cited at its definition.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/cpe.h"

#define NQ(s) (3 + 3 * (s)->n_links)

/* ------------------------------------------------------------------------------------------------ */
/* Rotation R = Rz(psi) Ry(theta) Rx(phi), row-major (Link3D.Rb_I in the .robot files) */
void cpo_rot(const double a[3], double R[9]) {
    double sf = sin(a[0]), cf = cos(a[0]), st = sin(a[1]), ct = cos(a[1]), sp = sin(a[2]), cp = cos(a[2]);
    R[0] = cp * ct; R[1] = sf * st * cp - sp * cf; R[2] = sf * sp + st * cf * cp;
    R[3] = sp * ct; R[4] = sf * sp * st + cf * cp; R[5] = -sf * cp + sp * st * cf;
    R[6] = -st;     R[7] = sf * ct;                R[8] = cf * ct;
}

/* dR[j] = dR/d(angle j), j = phi,theta,psi */
void cpo_drot(const double a[3], double dR[3][9]) {
    double sf = sin(a[0]), cf = cos(a[0]), st = sin(a[1]), ct = cos(a[1]), sp = sin(a[2]), cp = cos(a[2]);
    double* d = dR[0]; /* d/dphi */
    d[0] = 0; d[1] = cf * st * cp + sp * sf; d[2] = cf * sp - st * sf * cp;
    d[3] = 0; d[4] = cf * sp * st - sf * cp; d[5] = -cf * cp - sp * st * sf;
    d[6] = 0; d[7] = cf * ct;                d[8] = -sf * ct;
    d = dR[1]; /* d/dtheta */
    d[0] = -cp * st; d[1] = sf * ct * cp; d[2] = ct * cf * cp;
    d[3] = -sp * st; d[4] = sf * sp * ct; d[5] = sp * ct * cf;
    d[6] = -ct;      d[7] = -sf * st;     d[8] = -cf * st;
    d = dR[2]; /* d/dpsi */
    d[0] = -sp * ct; d[1] = -sf * st * sp - cp * cf; d[2] = sf * cp - st * cf * sp;
    d[3] = cp * ct;  d[4] = sf * cp * st - cf * sp;  d[5] = sf * sp + cp * st * cf;
    d[6] = 0; d[7] = 0; d[8] = 0;
}

static void matvec3(const double* R, const double* v, double* out) {
    for (int i = 0; i < 3; i++) out[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}

/* FK chain: origin_i = origin_parent + R_parent * attach_i (cheetah.py:32-38,109-200) */
void cpo_fk(const cpe_skeleton* s, const double* q, double* R /*[nl][9]*/, double* origin /*[nl][3]*/) {
    for (int i = 0; i < s->n_links; i++) {
        cpo_rot(q + 3 + 3 * i, R + 9 * i);
        if (s->parent[i] < 0) {
            for (int d = 0; d < 3; d++) origin[3 * i + d] = q[d];
        } else {
            int p = s->parent[i];
            double v[3];
            matvec3(R + 9 * p, s->attach[i], v);
            for (int d = 0; d < 3; d++) origin[3 * i + d] = origin[3 * p + d] + v[d];
        }
    }
}

/* marker model (acinoset_misc.py:1581-1659, order = get_markers() :1914-1940) */
void cpo_markers(const cpe_skeleton* s, const double* q, double* pos /*[L][3]*/) {
    double R[CPE_MAX_LINKS * 9], o[CPE_MAX_LINKS * 3];
    cpo_fk(s, q, R, o);
    for (int l = 0; l < s->n_markers; l++) {
        int k = s->marker_link[l];
        double v[3];
        matvec3(R + 9 * k, s->marker_off[l], v);
        for (int d = 0; d < 3; d++) pos[3 * l + d] = o[3 * k + d] + v[d];
    }
}

/* centre of mass (acinoset_misc.py:722-742) */
void cpo_com(const cpe_skeleton* s, const double* q, double* com) {
    double R[CPE_MAX_LINKS * 9], o[CPE_MAX_LINKS * 3], M = 0;
    cpo_fk(s, q, R, o);
    com[0] = com[1] = com[2] = 0;
    for (int i = 0; i < s->n_links; i++) {
        double v[3];
        matvec3(R + 9 * i, s->com[i], v);
        for (int d = 0; d < 3; d++) com[d] += s->mass[i] * (o[3 * i + d] + v[d]);
        M += s->mass[i];
    }
    for (int d = 0; d < 3; d++) com[d] /= M;
}

/* markers + dense Jacobian d pos / d q  ([L][3][nq]) */
void cpo_markers_jac(const cpe_skeleton* s, const double* q, double* pos, double* dpos) {
    int nq = NQ(s);
    double R[CPE_MAX_LINKS * 9], o[CPE_MAX_LINKS * 3];
    cpo_fk(s, q, R, o);
    memset(dpos, 0, sizeof(double) * s->n_markers * 3 * nq);
    for (int l = 0; l < s->n_markers; l++) {
        int k = s->marker_link[l];
        double v[3];
        matvec3(R + 9 * k, s->marker_off[l], v);
        for (int d = 0; d < 3; d++) {
            pos[3 * l + d] = o[3 * k + d] + v[d];
            dpos[(3 * l + d) * nq + d] = 1.0;
        }
        /* walk up the chain: link k carries vector vk (marker offset, then the attach of the child) */
        const double* vk = s->marker_off[l];
        int link = k;
        while (link >= 0) {
            double dR[3][9];
            cpo_drot(q + 3 + 3 * link, dR);
            for (int j = 0; j < 3; j++) {
                double w[3];
                matvec3(dR[j], vk, w);
                for (int d = 0; d < 3; d++) dpos[(3 * l + d) * nq + 3 + 3 * link + j] = w[d];
            }
            vk = s->attach[link];
            link = s->parent[link];
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* camera projection with optional 2x3 derivative G = d(u,v)/dp
 * fisheye: acinoset_misc.py:1663-1679 ; pinhole-radial: acinoset_misc.py:1682-1696 */
void cpo_project(const cpe_camera* c, const double p[3], double uv[2], double* G /*6 or NULL*/) {
    double X[3];
    for (int i = 0; i < 3; i++) X[i] = c->R[3 * i] * p[0] + c->R[3 * i + 1] * p[1] + c->R[3 * i + 2] * p[2] + c->t[i];
    double a = X[0] / X[2], b = X[1] / X[2];
    double r = sqrt(a * a + b * b);
    double g, dg_dr; /* x_p = a*g(r), y_p = b*g(r) */
    if (c->model == CPE_CAM_FISHEYE) {
        double th = atan(r), t2 = th * th;
        double thd = th * (1 + c->D[0] * t2 + c->D[1] * t2 * t2 + c->D[2] * t2 * t2 * t2 + c->D[3] * t2 * t2 * t2 * t2);
        double dthd = 1 + 3 * c->D[0] * t2 + 5 * c->D[1] * t2 * t2 + 7 * c->D[2] * t2 * t2 * t2 + 9 * c->D[3] * t2 * t2 * t2 * t2;
        double den = r + 1e-12;
        g = thd / den;
        dg_dr = (dthd / (1 + r * r) * den - thd) / (den * den);
    } else {
        double r2 = r * r;
        g = 1 + c->D[0] * r2 + c->D[1] * r2 * r2 + c->D[2] * r2 * r2 * r2;
        dg_dr = (2 * c->D[0] + 4 * c->D[1] * r2 + 6 * c->D[2] * r2 * r2) * r;
    }
    uv[0] = c->fx * a * g + c->cx;
    uv[1] = c->fy * b * g + c->cy;
    if (G) {
        double dr_da = r > 0 ? a / r : 0, dr_db = r > 0 ? b / r : 0;
        /* d(xp,yp)/d(a,b) */
        double xa = g + a * dg_dr * dr_da, xb = a * dg_dr * dr_db;
        double ya = b * dg_dr * dr_da, yb = g + b * dg_dr * dr_db;
        /* d(a,b)/dX */
        double iz = 1.0 / X[2];
        double dadX[3] = {iz, 0, -a * iz}, dbdX[3] = {0, iz, -b * iz};
        for (int k = 0; k < 3; k++) {
            /* d(a)/dp_k = sum_i dadX[i] R[i][k] */
            double dak = 0, dbk = 0;
            for (int i = 0; i < 3; i++) { dak += dadX[i] * c->R[3 * i + k]; dbk += dbdX[i] * c->R[3 * i + k]; }
            G[k] = c->fx * (xa * dak + xb * dbk);
            G[3 + k] = c->fy * (ya * dak + yb * dbk);
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* redescending loss (acinoset_misc.py:2001-2015) and its first two derivatives w.r.t. err.
 * out[0] = rho, out[1] = d rho/d err, out[2] = d2 rho / d err2 (derivative at err==0 taken as 0) */
static void sig3(double t, double x, double* s, double* s1, double* s2) {
    double v = 1.0 / (1.0 + exp(-(x - t)));
    *s = v; *s1 = v * (1 - v); *s2 = v * (1 - v) * (1 - 2 * v);
}
void cpo_loss(double err, double a, double b, double c, double out[3]) {
    double e = fabs(err);
    double sa, sa1, sa2, sb, sb1, sb2, sc, sc1, sc2;
    sig3(a, e, &sa, &sa1, &sa2); sig3(b, e, &sb, &sb1, &sb2); sig3(c, e, &sc, &sc1, &sc2);
    double lin = a * e - a * a / 2;
    double cb = c - b, u = (c - e) / cb;
    double k = a * b - a * a / 2 + (a * cb / 2) * (1 - u * u), k1 = a * (c - e) / cb, k2 = -a / cb;
    double K = a * b - a * a / 2 + a * cb / 2;
    double A = (1 - sa) / 2 * e * e;
    double A1 = -sa1 / 2 * e * e + (1 - sa) * e;
    double A2 = -sa2 / 2 * e * e - 2 * sa1 * e + (1 - sa);
    double B = (sa - sb) * lin, B1 = (sa1 - sb1) * lin + (sa - sb) * a, B2 = (sa2 - sb2) * lin + 2 * (sa1 - sb1) * a;
    double C = (sb - sc) * k, C1 = (sb1 - sc1) * k + (sb - sc) * k1;
    double C2 = (sb2 - sc2) * k + 2 * (sb1 - sc1) * k1 + (sb - sc) * k2;
    double D = sc * K, D1 = sc1 * K, D2 = sc2 * K;
    out[0] = A + B + C + D;
    double sgn = err > 0 ? 1.0 : (err < 0 ? -1.0 : 0.0);
    out[1] = (A1 + B1 + C1 + D1) * sgn;
    out[2] = A2 + B2 + C2 + D2;
}

/* curvature weight of one scalar residual in the Gauss-Newton block (PSD by construction):
 * mode 0: max(rho''(s), rho'(|s|)/|s|, 0) -- the true curvature where the loss is convex (|s| < a),
 *         the IRLS majoriser weight on the linear / redescending pieces;
 * mode 1: max(rho''(s), 0) (plain Newton clipped) */
static double curv_weight(const double L[3], double s, int mode) {
    double c2 = L[2] > 0 ? L[2] : 0.0;
    if (mode == 1) return c2;
    double as = fabs(s), irls = 0.0;
    if (as > 1e-12 && L[1] * s > 0) irls = fabs(L[1]) / as;
    return c2 > irls ? c2 : irls;
}

/* ------------------------------------------------------------------------------------------------ */
/* joint equalities (`angle_constraints` of the .robot files, cheetah.py:71-72,101,160-161).
 * c[nc], optional dense Jacobian Cq[nc][nq]. Returns nc. */
int cpo_constraints(const cpe_skeleton* s, const double* q, double* c, double* Cq) {
    int nq = NQ(s), nc = 0;
    for (int j = 0; j < s->n_joints; j++) nc += s->joint_kind[j] == CPE_JOINT_REVOLUTE_Y ? 2 : 1;
    if (Cq) memset(Cq, 0, sizeof(double) * nc * nq);
    int row = 0;
    for (int j = 0; j < s->n_joints; j++) {
        int p = s->joint_parent[j], ch = s->joint_child[j];
        double Rp[9], Rc[9], dRp[3][9], dRc[3][9];
        cpo_rot(q + 3 + 3 * p, Rp); cpo_rot(q + 3 + 3 * ch, Rc);
        cpo_drot(q + 3 + 3 * p, dRp); cpo_drot(q + 3 + 3 * ch, dRc);
        int cols[2] = {0, 2}; /* child x axis, child z axis */
        int first = s->joint_kind[j] == CPE_JOINT_REVOLUTE_Y ? 0 : 1;
        for (int t = first; t < 2; t++) {
            int cc = cols[t];
            double v = 0;
            for (int i = 0; i < 3; i++) v += Rp[3 * i + 1] * Rc[3 * i + cc];
            c[row] = v;
            if (Cq) {
                for (int a = 0; a < 3; a++) {
                    double dp = 0, dc = 0;
                    for (int i = 0; i < 3; i++) { dp += dRp[a][3 * i + 1] * Rc[3 * i + cc]; dc += Rp[3 * i + 1] * dRc[a][3 * i + cc]; }
                    Cq[row * nq + 3 + 3 * p + a] += dp;
                    Cq[row * nq + 3 + 3 * ch + a] += dc;
                }
            }
            row++;
        }
    }
    return nc;
}

/* lists of dependent / independent dofs: revolute -> child phi,psi ; hooke -> child phi */
int cpo_split_dofs(const cpe_skeleton* s, int* indep, int* dep) {
    int nq = NQ(s), nd = 0, ni = 0;
    char isdep[CPE_MAX_NQ];
    memset(isdep, 0, sizeof(isdep));
    for (int j = 0; j < s->n_joints; j++) {
        int ch = s->joint_child[j];
        if (s->joint_kind[j] == CPE_JOINT_REVOLUTE_Y) {
            dep[nd++] = 3 + 3 * ch; dep[nd++] = 3 + 3 * ch + 2;
            isdep[3 + 3 * ch] = isdep[3 + 3 * ch + 2] = 1;
        } else {
            dep[nd++] = 3 + 3 * ch; isdep[3 + 3 * ch] = 1;
        }
    }
    for (int p = 0; p < nq; p++) if (!isdep[p]) indep[ni++] = p;
    return ni;
}

/* closed-form solution of the joint equalities for the dependent angles (branch child.y = +parent.y,
 * the one reached from the reference's initial guess phi=theta=0, psi=heading, acinoset_opt.py:574-583) */
int cpo_project_dependents(const cpe_skeleton* s, double* q) {
    int clamped = 0; /* 1 if a revolute child sits in the gimbal band |cos(theta)| < |a_z| (no solution) */
    for (int j = 0; j < s->n_joints; j++) {
        int p = s->joint_parent[j], ch = s->joint_child[j];
        double Rp[9];
        cpo_rot(q + 3 + 3 * p, Rp);
        double ax = Rp[1], ay = Rp[4], az = Rp[7];
        double* ang = q + 3 + 3 * ch;
        double st = sin(ang[1]), ct = cos(ang[1]);
        if (s->joint_kind[j] == CPE_JOINT_REVOLUTE_Y) {
            double sphi = az / ct;
            if (sphi > 1) { sphi = 1; clamped = 1; }
            if (sphi < -1) { sphi = -1; clamped = 1; }
            double phi = asin(sphi), cphi = cos(phi);
            double psi = atan2(ay, ax) - atan2(cphi, sphi * st);
            double ref = q[3 + 3 * p + 2];
            psi += 2 * M_PI * round((ref - psi) / (2 * M_PI));
            ang[0] = phi; ang[2] = psi;
        } else {
            double sp = sin(ang[2]), cp = cos(ang[2]);
            double num = ax * st * cp + ay * st * sp + az * ct;
            double den = ay * cp - ax * sp;
            ang[0] = atan2(num, den);
        }
    }
    return clamped;
}

/* dense LU solve A X = B in place (n x n, nrhs); returns 0 on success */
static int lu_solve(int n, double* A, int nrhs, double* Bm) {
    for (int k = 0; k < n; k++) {
        int piv = k; double mx = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; i++) if (fabs(A[i * n + k]) > mx) { mx = fabs(A[i * n + k]); piv = i; }
        if (mx < 1e-300) return 1;
        if (piv != k) {
            for (int j = 0; j < n; j++) { double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            for (int j = 0; j < nrhs; j++) { double t = Bm[k * nrhs + j]; Bm[k * nrhs + j] = Bm[piv * nrhs + j]; Bm[piv * nrhs + j] = t; }
        }
        for (int i = k + 1; i < n; i++) {
            double f = A[i * n + k] / A[k * n + k];
            if (f == 0) continue;
            for (int j = k; j < n; j++) A[i * n + j] -= f * A[k * n + j];
            for (int j = 0; j < nrhs; j++) Bm[i * nrhs + j] -= f * Bm[k * nrhs + j];
        }
    }
    for (int k = n - 1; k >= 0; k--)
        for (int j = 0; j < nrhs; j++) {
            double v = Bm[k * nrhs + j];
            for (int i = k + 1; i < n; i++) v -= A[k * n + i] * Bm[i * nrhs + j];
            Bm[k * nrhs + j] = v / A[k * n + k];
        }
    return 0;
}

/* tangent basis of the constraint manifold: Z[nq][nu], rows of independent dofs = identity, rows of
 * dependent dofs = S = -(dc/dy)^-1 dc/du (implicit function theorem on cpo_constraints) */
int cpo_tangent_basis(const cpe_skeleton* s, const double* q, double* Z) {
    int nq = NQ(s), indep[CPE_MAX_NQ], dep[CPE_MAX_NQ];
    int nu = cpo_split_dofs(s, indep, dep), nd = nq - nu;
    double c[64], Cq[64 * CPE_MAX_NQ];
    int nc = cpo_constraints(s, q, c, Cq);
    if (nc != nd) return -1;
    double* Cy = (double*)malloc(sizeof(double) * nd * nd);
    double* Cu = (double*)malloc(sizeof(double) * nd * nu);
    for (int r = 0; r < nd; r++) {
        for (int k = 0; k < nd; k++) Cy[r * nd + k] = Cq[r * nq + dep[k]];
        for (int k = 0; k < nu; k++) Cu[r * nu + k] = -Cq[r * nq + indep[k]];
    }
    int bad = lu_solve(nd, Cy, nu, Cu);
    memset(Z, 0, sizeof(double) * nq * nu);
    for (int k = 0; k < nu; k++) Z[indep[k] * nu + k] = 1.0;
    for (int r = 0; r < nd; r++) for (int k = 0; k < nu; k++) Z[dep[r] * nu + k] = Cu[r * nu + k];
    free(Cy); free(Cu);
    return bad ? -2 : nu;
}

/* ------------------------------------------------------------------------------------------------ */
/* acceleration slack of the constant-acceleration model for one sequence (acinoset_misc.py:639-677
 * with the implicit-Euler collocation of make_pyomo_model, acinoset_opt.py:508; free dq0, ddq0 make the
 * slack of frames 1 and 2 vanish -- SURVEY A.5): eps_n = (q_n - 3q_{n-1} + 3q_{n-2} - q_{n-3})/h^2 */
void cpo_motion_eps(int nq, int N, double h, const double* q, double* eps) {
    memset(eps, 0, sizeof(double) * N * nq);
    for (int n = 3; n < N; n++)
        for (int p = 0; p < nq; p++)
            eps[n * nq + p] = (q[n * nq + p] - 3 * q[(n - 1) * nq + p] + 3 * q[(n - 2) * nq + p] - q[(n - 3) * nq + p]) / (h * h);
}

/* dq, ddq of the collocation with the gauge ddq0 = ddq1 = ddq2 of the stored solutions */
void cpo_derivatives(int nq, int N, double h, const double* q, double* dq, double* ddq) {
    memset(dq, 0, sizeof(double) * N * nq); memset(ddq, 0, sizeof(double) * N * nq);
    for (int n = 1; n < N; n++) for (int p = 0; p < nq; p++) dq[n * nq + p] = (q[n * nq + p] - q[(n - 1) * nq + p]) / h;
    for (int n = 2; n < N; n++) for (int p = 0; p < nq; p++) ddq[n * nq + p] = (dq[n * nq + p] - dq[(n - 1) * nq + p]) / h;
    if (N >= 3) for (int p = 0; p < nq; p++) { ddq[nq + p] = ddq[2 * nq + p]; ddq[p] = ddq[2 * nq + p]; }
    if (N >= 2) for (int p = 0; p < nq; p++) dq[p] = dq[nq + p] - h * ddq[nq + p];
}

/* ------------------------------------------------------------------------------------------------ */
/* metric-1 evaluation for one sequence: residual, dense Jacobian [C][L][2][nq], acceleration slack,
 * per-frame robust cost (acinoset_misc.py:269-288, 459-484, 639-677) */
void cpo_eval_resjac(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, int N,
                     const double* q, const double* meas, const double* weight,
                     double* r, double* Jdense, double* eps, double* cost) {
    int nq = NQ(s), L = s->n_markers;
    double* pos = (double*)malloc(sizeof(double) * L * 3);
    double* dpos = (double*)malloc(sizeof(double) * L * 3 * nq);
    for (int n = 0; n < N; n++) {
        cpo_markers_jac(s, q + n * nq, pos, dpos);
        double fc = 0;
        for (int c = 0; c < C; c++)
            for (int l = 0; l < L; l++) {
                double uv[2], G[6];
                cpo_project(&cams[c], pos + 3 * l, uv, G);
                size_t base = ((size_t)(n * C + c) * L + l);
                for (int d = 0; d < 2; d++) {
                    double e = uv[d] - meas[base * 2 + d];
                    if (r) r[base * 2 + d] = e;
                    if (Jdense)
                        for (int p = 0; p < nq; p++)
                            Jdense[(base * 2 + d) * nq + p] = G[3 * d] * dpos[(3 * l) * nq + p] + G[3 * d + 1] * dpos[(3 * l + 1) * nq + p] + G[3 * d + 2] * dpos[(3 * l + 2) * nq + p];
                    double Lo[3];
                    cpo_loss(cams[c].mult * weight[base] * e, o->loss_a, o->loss_b, o->loss_c, Lo);
                    fc += Lo[0];
                }
            }
        if (cost) cost[n] = fc;
    }
    if (eps) cpo_motion_eps(nq, N, o->h, q, eps);
    free(pos); free(dpos);
}

/* bench.py's all-cores CPU baseline: `reps` passes of cpo_eval_resjac over B sequences, sequences spread over OpenMP
 * threads, each thread writing into its own output buffers (what is timed is the evaluation, not one shared result
 * array).  Returns the number of threads used.  Test infrastructure like everything in this file. */
#include <omp.h>
int cpo_eval_resjac_batch(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, int B, int N,
                          const double* q, const double* meas, const double* weight, int reps, int threads, double* checksum) {
    int nq = NQ(s), L = s->n_markers, used = 1;
    double total = 0;
    if (threads < 1) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads) reduction(+ : total)
    {
#pragma omp single
        used = omp_get_num_threads();
        size_t nr = (size_t)N * C * L * 2;
        double* r = (double*)malloc(sizeof(double) * nr);
        double* J = (double*)malloc(sizeof(double) * nr * nq);
        double* eps = (double*)malloc(sizeof(double) * (size_t)N * nq);
        double* cost = (double*)malloc(sizeof(double) * N);
#pragma omp for schedule(dynamic, 1)
        for (int k = 0; k < B * reps; k++) {
            int b = k % B;
            cpo_eval_resjac(s, cams, C, o, N, q + (size_t)b * N * nq, meas + (size_t)b * nr, weight + (size_t)b * N * C * L, r, J, eps, cost);
            for (int n = 0; n < N; n++) total += cost[n];
        }
        free(r); free(J); free(eps); free(cost);
    }
    if (checksum) *checksum = total;
    return used;
}

/* ------------------------------------------------------------------------------------------------ */
/* relative angles x (acinoset_misc.py:508-528 numpy branch + mask :1699-1757), linear in q */
void cpo_relative_angles(const cpe_skeleton* s, const double* q, double* x /*[nu]*/) {
    int indep[CPE_MAX_NQ], dep[CPE_MAX_NQ];
    int nu = cpo_split_dofs(s, indep, dep);
    for (int k = 0; k < nu; k++) {
        int p = indep[k];
        x[k] = s->rel_ref[p] < 0 ? q[p] : s->rel_sign[p] * (q[p] - q[s->rel_ref[p]]);
    }
}

/* GMM pose prior value, gradient and PSD curvature w.r.t. x22 = x[6:] (acinoset_misc.py:680-714):
 * f = -log(sum_k w_k N(x;mu_k,Sigma_k) + 1e-12).  Hess approx = sum_k gamma_k P_k (responsibility-weighted). */
static double gmm_eval(const cpe_priors* pr, const double* x, double* grad, double* H) {
    int K = pr->gmm_k, D = pr->gmm_dim;
    double lp[CPE_MAX_GMM], v[CPE_MAX_GMM][CPE_NX], mx = -1e300;
    for (int k = 0; k < K; k++) {
        double qf = 0;
        for (int i = 0; i < D; i++) {
            double a = 0;
            for (int j = 0; j < D; j++) a += pr->gmm_P[k][i][j] * (x[j] - pr->gmm_mu[k][j]);
            v[k][i] = a; qf += a * (x[i] - pr->gmm_mu[k][i]);
        }
        lp[k] = pr->gmm_logw[k] - 0.5 * qf;
        if (lp[k] > mx) mx = lp[k];
    }
    double sum = 0;
    for (int k = 0; k < K; k++) sum += exp(lp[k] - mx);
    double S = sum * exp(mx) + 1e-12; /* mixture density + 1e-12 */
    double f = -log(S);
    if (grad) {
        for (int i = 0; i < D; i++) grad[i] = 0;
        if (H) for (int i = 0; i < D * D; i++) H[i] = 0;
        for (int k = 0; k < K; k++) {
            double gam = exp(lp[k]) / S;
            for (int i = 0; i < D; i++) grad[i] += gam * v[k][i];
            if (H) for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) H[i * D + j] += gam * pr->gmm_P[k][i][j];
        }
    }
    return f;
}
double cpo_gmm_cost(const cpe_priors* pr, const double* x22, double* grad) { return gmm_eval(pr, x22, grad, NULL); }

/* ------------------------------------------------------------------------------------------------ */
/* per-frame terms in reduced coordinates u (independent dofs): cost, gradient g[nu], PSD block Bm[nu][nu] */
/* ------------------------------------------------------------------------------------------------ */
/* Reduced coordinates of the solver.  The 28 coordinates u' are: the independent Euler
 * dofs of the trunk (base 6, bodyF 3, neck 3, theta/psi of both tails) and, for every leg link c, its
 * rotation alpha_c about the y axis of the body B the leg hangs from:  R_c = R_B * Ry(alpha_c).
 * This satisfies the two revolute equalities of every leg joint identically (child.y = body.y) and,
 * unlike the absolute Euler pitch theta_c, covers BOTH solution branches of those equalities smoothly
 * (real runs swing limbs through +-90 deg of pitch: tests/golden/fk_csv_pin.npz).  alpha_c sits at the
 * position of theta_c in the coordinate vector.  State of one frame: st[nq + nrev] = Euler q (54, kept
 * consistent by state_sync) followed by the nrev leg angles alpha. */
struct kin_s;
typedef struct {
    const cpe_skeleton* s; const cpe_camera* cams; int C; const cpe_options* o; const cpe_priors* pr;
    struct kin_s* kin;            /* physics-based model (cpo_solve_kinetic), NULL for the kinematic models */
    const double* tau;            /* shutter delays per camera (cpo_solve_shutter), NULL = off */
    double* rcb;                  /* [N][C][9]: per camera sum_l G^T rho' w (3) and sum_l G^T cw G (6) of the last evaluation with gradient */
    int nq, nu, indep[CPE_MAX_NQ], dep[CPE_MAX_NQ], u_of_q[CPE_MAX_NQ];
    int nrev, ns;                 /* revolute (leg) links, state size nq + nrev */
    int rev_joint[CPE_MAX_JOINTS];/* joint index of revolute r */
    int rev_body[CPE_MAX_JOINTS]; /* body link whose y axis the leg shares */
    int rev_of_u[CPE_NX];         /* u' index -> revolute index, or -1 for a plain Euler coordinate */
} ctx_t;

static void ctx_init(ctx_t* x, const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr) {
    x->s = s; x->cams = cams; x->C = C; x->o = o; x->pr = pr; x->nq = NQ(s); x->kin = NULL; x->tau = NULL; x->rcb = NULL;
    x->nu = cpo_split_dofs(s, x->indep, x->dep);
    for (int p = 0; p < x->nq; p++) x->u_of_q[p] = -1;
    for (int k = 0; k < x->nu; k++) { x->u_of_q[x->indep[k]] = k; x->rev_of_u[k] = -1; }
    x->nrev = 0;
    int body_of_link[CPE_MAX_LINKS];
    for (int i = 0; i < s->n_links; i++) body_of_link[i] = -1;
    for (int j = 0; j < s->n_joints; j++)
        if (s->joint_kind[j] == CPE_JOINT_REVOLUTE_Y) {
            int p = s->joint_parent[j], c = s->joint_child[j];
            int body = body_of_link[p] >= 0 ? body_of_link[p] : p;
            body_of_link[c] = body;
            x->rev_joint[x->nrev] = j; x->rev_body[x->nrev] = body;
            x->rev_of_u[x->u_of_q[3 + 3 * c + 1]] = x->nrev;
            x->nrev++;
        }
    x->ns = x->nq + x->nrev;
}

static void mat3mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static double wrap_pi(double a) { return a - 2 * M_PI * round(a / (2 * M_PI)); }

/* Euler q of the whole skeleton from the trunk coordinates (in st[0..nq)) and the leg angles alpha */
static void state_sync(const ctx_t* x, double* st) {
    const cpe_skeleton* s = x->s;
    for (int r = 0; r < x->nrev; r++) {
        int c = s->joint_child[x->rev_joint[r]], B = x->rev_body[r];
        double RB[9], Rc[9], al = st[x->nq + r], ca = cos(al), sa = sin(al);
        double Ry[9] = {ca, 0, sa, 0, 1, 0, -sa, 0, ca};
        cpo_rot(st + 3 + 3 * B, RB);
        mat3mul(RB, Ry, Rc);
        double* e = st + 3 + 3 * c;
        double sth = -Rc[6]; if (sth > 1) sth = 1; if (sth < -1) sth = -1;
        e[1] = asin(sth);                       /* principal pitch: the STATE's triple (FK, outputs, physics rows); the cost terms see theta_B + alpha, cost_view() */
        e[0] = atan2(Rc[7], Rc[8]);
        e[2] = atan2(Rc[3], Rc[0]);
        e[2] += 2 * M_PI * round((st[3 + 3 * B + 2] - e[2]) / (2 * M_PI));
    }
    /* hooke joints: phi in closed form (cpo_project_dependents handles exactly those when the revolute children are consistent) */
    for (int j = 0; j < s->n_joints; j++)
        if (s->joint_kind[j] == CPE_JOINT_HOOKE_YZ) {
            int p = s->joint_parent[j], ch = s->joint_child[j];
            double Rp[9]; cpo_rot(st + 3 + 3 * p, Rp);
            double ax = Rp[1], ay = Rp[4], az = Rp[7];
            double* ang = st + 3 + 3 * ch;
            double sth = sin(ang[1]), cth = cos(ang[1]), sp = sin(ang[2]), cp = cos(ang[2]);
            ang[0] = atan2(ax * sth * cp + ay * sth * sp + az * cth, ay * cp - ax * sp);
        }
}

/* state from a (possibly inconsistent) Euler q: alpha_c = angle of R_B^T R_c about y */
static void state_from_q(const ctx_t* x, const double* q, double* st) {
    const cpe_skeleton* s = x->s;
    memcpy(st, q, sizeof(double) * x->nq);
    for (int r = 0; r < x->nrev; r++) {
        int c = s->joint_child[x->rev_joint[r]], B = x->rev_body[r];
        double RB[9], Rc[9];
        cpo_rot(q + 3 + 3 * B, RB); cpo_rot(q + 3 + 3 * c, Rc);
        /* M = RB^T Rc ; alpha = atan2(M[0][2], M[0][0]) */
        double m00 = RB[0] * Rc[0] + RB[3] * Rc[3] + RB[6] * Rc[6], m02 = RB[0] * Rc[2] + RB[3] * Rc[5] + RB[6] * Rc[8];
        st[x->nq + r] = atan2(m02, m00);
    }
    state_sync(x, st);
}

static void state_add(const ctx_t* x, double* st, int k, double d) {
    if (x->rev_of_u[k] >= 0) st[x->nq + x->rev_of_u[k]] += d; else st[x->indep[k]] += d;
}

/* Z' = d(Euler q)/d(u') [nq][nu] by central differences of the explicit map state_sync (the HIP kernels
 * use the analytic form; the two are independent derivations) */
static void state_jacobian(const ctx_t* x, const double* st, double* Zp) {
    const double h = 1e-6;
    double a[CPE_MAX_NQ + CPE_MAX_JOINTS], b[CPE_MAX_NQ + CPE_MAX_JOINTS];
    for (int k = 0; k < x->nu; k++) {
        memcpy(a, st, sizeof(double) * x->ns); memcpy(b, st, sizeof(double) * x->ns);
        state_add(x, a, k, h); state_add(x, b, k, -h);
        state_sync(x, a); state_sync(x, b);
        for (int p = 0; p < x->nq; p++) {
            double d = a[p] - b[p];
            if (p >= 3) d = wrap_pi(d);
            Zp[p * x->nu + k] = d / (2 * h);
        }
    }
}

/* COST VIEW of a frame's Euler angles: the terms of the objective that act on the Euler pitch of a leg link -- constant-acceleration cost,
 * joint ranges, learned priors -- take theta_B + alpha_c (pitch of the body the leg hangs from + the leg angle about the body's y axis): the reference's
 * variable for an unrolled trunk, smooth through +-90 degrees (the state keeps the principal triple: forward kinematics, outputs, physics terms).
 * qc [nq], Zc [nq][nu] (if Z): copies of the state's Euler part / of Z' with the leg-pitch entries / rows replaced. */
static void cost_view(const ctx_t* x, const double* st, const double* Z, double* qc, double* Zc) {
    const cpe_skeleton* s = x->s; int nq = x->nq, nu = x->nu;
    memcpy(qc, st, sizeof(double) * nq);
    if (Z) memcpy(Zc, Z, sizeof(double) * nq * nu);
    for (int r = 0; r < x->nrev; r++) {
        int c = s->joint_child[x->rev_joint[r]], B = x->rev_body[r], p = 3 + 3 * c + 1, pb = 3 + 3 * B + 1;
        qc[p] = st[pb] + st[nq + r];
        if (Z) for (int k = 0; k < nu; k++) Zc[p * nu + k] = Z[pb * nu + k] + (x->rev_of_u[k] == r ? 1.0 : 0.0);
    }
}

/* cost terms for one frame; returns meas cost, *cb bound-penalty cost, *cp pose-prior cost.  qn must point into a STATE (Euler part | leg angles). */
static double frame_terms(const ctx_t* x, const double* qn, const double* Zp /* [nq][nu], needed iff g */, const double* meas, const double* weight,
                          const double* mu, double* g, double* Bm, double* cb, double* cp, double* viol_out,
                          const double* x1, const double* x2 /* base positions of frames n-1, n-2 (shutter delay, n >= 2), or NULL */,
                          double* gx /* [6] gradient parts for the base position of frames n-1, n-2 */, double* rc9 /* [C][9] */) {
    const cpe_skeleton* s = x->s; int nq = x->nq, nu = x->nu, L = s->n_markers;
    double pos[CPE_MAX_MARKERS * 3];
    double* dpos = NULL; const double* Z = Zp; double* dpu = NULL;
    int want = g != NULL;
    double qc[CPE_MAX_NQ];
    double* Zc = want ? (double*)malloc(sizeof(double) * nq * nu) : NULL;
    cost_view(x, qn, want ? Zp : NULL, qc, Zc);
    if (want) {
        dpos = (double*)malloc(sizeof(double) * L * 3 * nq);
        dpu = (double*)malloc(sizeof(double) * L * 3 * nu);
        cpo_markers_jac(s, qn, pos, dpos);
        for (int i = 0; i < L * 3; i++)
            for (int k = 0; k < nu; k++) {
                double a = 0;
                for (int p = 0; p < nq; p++) a += dpos[i * nq + p] * Z[p * nu + k];
                dpu[i * nu + k] = a;
            }
        memset(g, 0, sizeof(double) * nu); memset(Bm, 0, sizeof(double) * nu * nu);
    } else cpo_markers(s, qn, pos);
    double fm = 0, rho0[3];
    double Ju[CPE_NX + 8];
    cpo_loss(0.0, x->o->loss_a, x->o->loss_b, x->o->loss_c, rho0);
    if (gx) for (int i = 0; i < 6; i++) gx[i] = 0;
    if (rc9) for (int i = 0; i < 9 * x->C; i++) rc9[i] = 0;
    for (int c = 0; c < x->C; c++) {
        /* shutter delay (acinoset_misc.py:283-285): camera c sees the markers displaced by q'_base tau_c + q''_base tau_c^2 = alpha x_n + beta x_n-1 + gamma x_n-2 */
        double al = 0, be = 0, ga = 0, sft[3] = {0, 0, 0};
        if (x->tau && x1) {
            double ta = x->tau[c], ih = 1.0 / x->o->h;
            al = ta * ih + ta * ta * ih * ih; be = -ta * ih - 2 * ta * ta * ih * ih; ga = ta * ta * ih * ih;
            for (int d = 0; d < 3; d++) sft[d] = al * qn[d] + be * x1[d] + ga * x2[d];
        }
        for (int l = 0; l < L; l++) {
            double w = x->cams[c].mult * weight[c * L + l];
            if (w == 0.0) { fm += 2 * rho0[0]; continue; } /* the reference still adds rho(0) for both coordinates */
            double uv[2], G[6], pt[3] = {pos[3 * l] + sft[0], pos[3 * l + 1] + sft[1], pos[3 * l + 2] + sft[2]};
            cpo_project(&x->cams[c], pt, uv, want ? G : NULL);
            for (int d = 0; d < 2; d++) {
                double e = uv[d] - meas[(c * L + l) * 2 + d], sres = w * e, Lo[3];
                cpo_loss(sres, x->o->loss_a, x->o->loss_b, x->o->loss_c, Lo);
                fm += Lo[0];
                if (!want) continue;
                double gs = Lo[1] * w, cw = curv_weight(Lo, sres, x->o->curvature) * w * w;
                for (int k = 0; k < nu; k++)
                    Ju[k] = G[3 * d] * dpu[(3 * l) * nu + k] + G[3 * d + 1] * dpu[(3 * l + 1) * nu + k] + G[3 * d + 2] * dpu[(3 * l + 2) * nu + k]
                            + (k < 3 ? al * G[3 * d + k] : 0.0);            /* d P / d x_n = (1 + alpha) I: reduced coordinates 0..2 are the base position */
                for (int k = 0; k < nu; k++) {
                    g[k] += gs * Ju[k];
                    if (Ju[k] != 0.0) for (int m = 0; m < nu; m++) Bm[k * nu + m] += cw * Ju[k] * Ju[m];
                }
                if (gx) for (int j = 0; j < 3; j++) { gx[j] += be * gs * G[3 * d + j]; gx[3 + j] += ga * gs * G[3 * d + j]; }
                if (rc9) {
                    double* r9 = rc9 + 9 * c;
                    for (int j = 0; j < 3; j++) r9[j] += gs * G[3 * d + j];
                    r9[3] += cw * G[3 * d] * G[3 * d]; r9[4] += cw * G[3 * d] * G[3 * d + 1]; r9[5] += cw * G[3 * d] * G[3 * d + 2];
                    r9[6] += cw * G[3 * d + 1] * G[3 * d + 1]; r9[7] += cw * G[3 * d + 1] * G[3 * d + 2]; r9[8] += cw * G[3 * d + 2] * G[3 * d + 2];
                }
            }
        }
    }
    /* angle bounds (cheetah.py:306-352) by an augmented Lagrangian: per bound and side a multiplier mu >= 0,
     * psi = (max(0, mu + kappa*viol_signed)^2 - mu^2) / (2 kappa); mu == NULL means all multipliers zero */
    double fb = 0, vmax = 0;
    for (int b = 0; b < s->n_bounds; b++) {
        int ia = s->bound_a[b], ib = s->bound_b[b];
        double kp = x->o->bound_penalty;
        double v = qc[ia] - (ib >= 0 ? qc[ib] : 0.0);
        double mu_up = mu ? mu[2 * b] : 0.0, mu_lo = mu ? mu[2 * b + 1] : 0.0;
        double t_up = mu_up + kp * (v - s->bound_up[b]), t_lo = mu_lo + kp * (s->bound_lo[b] - v);
        if (v - s->bound_up[b] > vmax) vmax = v - s->bound_up[b];
        if (s->bound_lo[b] - v > vmax) vmax = s->bound_lo[b] - v;
        double pu = t_up > 0 ? t_up : 0, pl = t_lo > 0 ? t_lo : 0;
        fb += (pu * pu - mu_up * mu_up + pl * pl - mu_lo * mu_lo) / (2 * kp);
        if (want && (pu > 0 || pl > 0)) {
            /* v is a difference of Euler angles; its gradient w.r.t. the reduced coordinates is a difference of rows of Z' */
            double gv = pu - pl, hv = kp * ((pu > 0) + (pl > 0)), dv[CPE_NX];
            for (int k = 0; k < nu; k++) dv[k] = Zc[ia * nu + k] - (ib >= 0 ? Zc[ib * nu + k] : 0.0);
            for (int k = 0; k < nu; k++) {
                if (dv[k] == 0.0) continue;
                g[k] += gv * dv[k];
                for (int m = 0; m < nu; m++) Bm[k * nu + m] += hv * dv[k] * dv[m];
            }
        }
    }
    if (viol_out) *viol_out = vmax;
    /* GMM pose prior on x[6:] */
    double fp = 0;
    if (x->pr && x->pr->gmm_k > 0) {
        double xr[CPE_NX], gr[CPE_NX], Hx[CPE_NX * CPE_NX];
        int D = x->pr->gmm_dim, off = nu - D;
        cpo_relative_angles(s, qc, xr);
        fp = gmm_eval(x->pr, xr + off, want ? gr : NULL, want ? Hx : NULL);
        if (want) {
            /* x_i = sign_i (q_p - q_ref(p)); d x_i / d u' = sign_i (Z'[p] - Z'[ref]) */
            double Xp[CPE_NX * CPE_NX];
            for (int i = 0; i < D; i++) {
                int pi = x->indep[off + i];
                double si = s->rel_ref[pi] < 0 ? 1.0 : s->rel_sign[pi];
                for (int k = 0; k < nu; k++) Xp[i * nu + k] = si * (Zc[pi * nu + k] - (s->rel_ref[pi] < 0 ? 0.0 : Zc[s->rel_ref[pi] * nu + k]));
            }
            for (int k = 0; k < nu; k++) {
                double a = 0;
                for (int i = 0; i < D; i++) a += Xp[i * nu + k] * gr[i];
                g[k] += a;
            }
            for (int i = 0; i < D; i++)
                for (int j = 0; j < D; j++) {
                    double hv = Hx[i * D + j];
                    if (hv == 0.0) continue;
                    for (int k = 0; k < nu; k++) {
                        double a = Xp[i * nu + k] * hv;
                        if (a == 0.0) continue;
                        for (int m = 0; m < nu; m++) Bm[k * nu + m] += a * Xp[j * nu + m];
                    }
                }
        }
    }
    if (cb) *cb = fb; if (cp) *cp = fp;
    if (want) { free(dpos); free(dpu); free(Zc); }
    return fm;
}

/* banded symmetric storage: A(i,j), j<=i, i-j<=kd stored at ab[i*(kd+1) + (i-j)] */
#define AB(i, j) ab[(size_t)(i) * (kd + 1) + ((i) - (j))]

static int band_cholesky(int n, int kd, double* ab) {
    for (int j = 0; j < n; j++) {
        double d = AB(j, j);
        int k0 = j - kd > 0 ? j - kd : 0;
        for (int k = k0; k < j; k++) d -= AB(j, k) * AB(j, k);
        if (!(d > 0)) return j + 1;
        d = sqrt(d); AB(j, j) = d;
        int i1 = j + kd < n - 1 ? j + kd : n - 1;
        for (int i = j + 1; i <= i1; i++) {
            double v = AB(i, j);
            int kk = i - kd > k0 ? i - kd : k0;
            for (int k = kk; k < j; k++) v -= AB(i, k) * AB(j, k);
            AB(i, j) = v / d;
        }
    }
    return 0;
}
static void band_solve(int n, int kd, const double* ab, double* x) {
    for (int i = 0; i < n; i++) {
        double v = x[i];
        int k0 = i - kd > 0 ? i - kd : 0;
        for (int k = k0; k < i; k++) v -= AB(i, k) * x[k];
        x[i] = v / AB(i, i);
    }
    for (int i = n - 1; i >= 0; i--) {
        double v = x[i];
        int k1 = i + kd < n - 1 ? i + kd : n - 1;
        for (int k = i + 1; k <= k1; k++) v -= AB(k, i) * x[k];
        x[i] = v / AB(i, i);
    }
}

/* whole-sequence evaluation in reduced coordinates. u -> q (dependents projected). Fills cost terms,
 * and, if g != NULL, gradient g[N*nu] and band matrix ab (without damping). */
typedef struct { double meas, model, pose, motion, bound, total, maxviol; } costs_t;

void cpo_eom_rows(const cpe_skeleton* s, const cpe_eom_options* o, const double* q, const double* dq, const double* ddq, double* E);
void cpo_dyn_forces(const cpe_skeleton* s, const cpe_dyn_options* o, const double* q, const double* tau, const double* lam,
                    const double* grf, double* Q);
#include "cpe_oracle_kinetic.inc"

static void seq_eval(const ctx_t* x, int N, int kd, double* st /* [N][ns] states, synced in place */, const double* meas, const double* weight,
                     const double* mu /* [N][nb][2] or NULL */, costs_t* ct, double* g, double* ab) {
    const cpe_skeleton* s = x->s; int nq = x->nq, nu = x->nu, ns = x->ns, L = s->n_markers, C = x->C;
    int n_tot = N * nu;
    memset(ct, 0, sizeof(*ct));
    double* gB = g ? (double*)malloc(sizeof(double) * (nu + nu * nu)) : NULL;
    double* Zall = g ? (double*)malloc(sizeof(double) * (size_t)N * nq * nu) : NULL;   /* Z'_n = d q_n / d u'_n */
    double* Qcv = (double*)malloc(sizeof(double) * (size_t)N * nq);                     /* cost view of every frame (cost_view): what the motion terms act on */
    double* Zcv = g ? (double*)malloc(sizeof(double) * (size_t)N * nq * nu) : NULL;
    if (g) { memset(g, 0, sizeof(double) * n_tot); memset(ab, 0, sizeof(double) * (size_t)n_tot * (kd + 1)); }
    for (int n = 0; n < N; n++) {
        double* sn = st + (size_t)n * ns;
        state_sync(x, sn);
        double* Zn = g ? Zall + (size_t)n * nq * nu : NULL;
        if (g) state_jacobian(x, sn, Zn);
        cost_view(x, sn, Zn, Qcv + (size_t)n * nq, g ? Zcv + (size_t)n * nq * nu : NULL);
        double cb, cp, vm;
        double gx6[6];
        int sd = x->tau != NULL && n >= 2;                       /* the displacement acts from node 2 on (include/cpe.h, cpe_solve_shutter) */
        double fm = frame_terms(x, sn, Zn, meas + (size_t)n * C * L * 2, weight + (size_t)n * C * L,
                                mu ? mu + (size_t)n * s->n_bounds * 2 : NULL, g ? gB : NULL, g ? gB + nu : NULL, &cb, &cp, &vm,
                                sd ? sn - ns : NULL, sd ? sn - 2 * ns : NULL, (g && sd) ? gx6 : NULL, (g && x->rcb) ? x->rcb + (size_t)n * C * 9 : NULL);
        if (g && sd) for (int j = 0; j < 3; j++) { g[(n - 1) * nu + j] += gx6[j]; g[(n - 2) * nu + j] += gx6[3 + j]; }
        ct->meas += fm; ct->bound += cb; ct->pose += cp;
        if (vm > ct->maxviol) ct->maxviol = vm;
        if (g) {
            for (int k = 0; k < nu; k++) {
                g[n * nu + k] += gB[k];
                for (int m = 0; m <= k; m++) AB(n * nu + k, n * nu + m) += gB[nu + k * nu + m];
            }
        }
    }
#define Q(n, p) Qcv[(size_t)(n) * nq + (p)]
    /* constant-acceleration model on the EULER angles (acinoset_misc.py:639-677): sum_{n>=3} w_p eps_{n,p}^2,
     * eps = third difference / h^2; Gauss-Newton through Z' (exactly quadratic for the trunk coordinates) */
    double ih2 = 1.0 / (x->o->h * x->o->h);
    static const double d3[4] = {-1, 3, -3, 1}; /* coefficients of q_{n-3..n} */
    for (int n = 3; n < N; n++)
        for (int p = 0; p < nq; p++) {
            double w = s->motion_w[p];
            if (w == 0) continue;
            double e = 0;
            for (int t = 0; t < 4; t++) e += d3[t] * Q(n - 3 + t, p);
            e *= ih2;
            ct->model += w * e * e;
            if (!g) continue;
            for (int t = 0; t < 4; t++) {
                const double* Za = Zcv + ((size_t)(n - 3 + t) * nq + p) * nu;
                for (int k = 0; k < nu; k++) {
                    if (Za[k] == 0.0) continue;
                    int ia = (n - 3 + t) * nu + k;
                    g[ia] += 2 * w * e * d3[t] * ih2 * Za[k];
                    for (int t2 = 0; t2 <= t; t2++) {
                        const double* Zb = Zcv + ((size_t)(n - 3 + t2) * nq + p) * nu;
                        for (int k2 = 0; k2 < nu; k2++) {
                            int ib = (n - 3 + t2) * nu + k2;
                            if (ib > ia || Zb[k2] == 0.0) continue;
                            AB(ia, ib) += 2 * w * d3[t] * d3[t2] * ih2 * ih2 * Za[k] * Zb[k2];
                        }
                    }
                }
            }
        }
    /* linear autoregressive motion prior on x (acinoset_misc.py:291-336): for n >= window
     * slack = x_n - (coef.[x_{n-w};...;x_{n-1}] + b), cost sum_p lr_w[p] slack_p^2 ; x = relative angles of q */
    if (x->pr && x->pr->lr_window > 0) {
        int W = x->pr->lr_window;
        double* xs = (double*)malloc(sizeof(double) * N * nu);
        double* Xp = g ? (double*)malloc(sizeof(double) * (size_t)N * nu * nu) : NULL;     /* d x_n / d u'_n */
        for (int n = 0; n < N; n++) {
            cpo_relative_angles(s, Qcv + (size_t)n * nq, xs + n * nu);
            if (g)
                for (int i = 0; i < nu; i++) {
                    int pi = x->indep[i];
                    double si = s->rel_ref[pi] < 0 ? 1.0 : s->rel_sign[pi];
                    const double* Za = Zcv + ((size_t)n * nq + pi) * nu;
                    const double* Zr = s->rel_ref[pi] < 0 ? NULL : Zcv + ((size_t)n * nq + s->rel_ref[pi]) * nu;
                    for (int k = 0; k < nu; k++) Xp[((size_t)n * nu + i) * nu + k] = si * (Za[k] - (Zr ? Zr[k] : 0.0));
                }
        }
        double* Ku = g ? (double*)malloc(sizeof(double) * (W + 1) * nu) : NULL;          /* d slack_p / d u'_{n-W+t} */
        for (int n = W; n < N; n++)
            for (int p = 0; p < nu; p++) {
                double w = x->pr->lr_w[p];
                if (w == 0) continue;
                double sl = xs[n * nu + p] - x->pr->lr_b[p];
                for (int t = 0; t < W; t++) for (int j = 0; j < nu; j++) sl -= x->pr->lr_coef[p][t * nu + j] * xs[(n - W + t) * nu + j];
                ct->motion += w * sl * sl;
                if (!g) continue;
                for (int t = 0; t <= W; t++)
                    for (int k = 0; k < nu; k++) {
                        double a = 0;
                        const double* Xa = Xp + (size_t)(n - W + t) * nu * nu;
                        if (t == W) a = Xa[p * nu + k];
                        else for (int j = 0; j < nu; j++) a -= x->pr->lr_coef[p][t * nu + j] * Xa[j * nu + k];
                        Ku[t * nu + k] = a;
                    }
                for (int t = 0; t <= W; t++) for (int k = 0; k < nu; k++) {
                    double kj = Ku[t * nu + k];
                    if (kj == 0) continue;
                    int ia = (n - W + t) * nu + k;
                    g[ia] += 2 * w * sl * kj;
                    for (int t2 = 0; t2 <= t; t2++) for (int k2 = 0; k2 < nu; k2++) {
                        int ib = (n - W + t2) * nu + k2;
                        if (ib > ia) continue;
                        double kj2 = Ku[t2 * nu + k2];
                        if (kj2 != 0) AB(ia, ib) += 2 * w * kj * kj2;
                    }
                }
            }
        free(xs); if (Xp) free(Xp); if (Ku) free(Ku);
    }
#undef Q
    if (x->kin) kin_seq_terms(x, N, kd, st, ct, g, ab);
    ct->total = ct->meas + ct->model + ct->pose + ct->motion + ct->bound;
    if (gB) free(gB);
    if (Zall) free(Zall);
    free(Qcv); if (Zcv) free(Zcv);
}

/* Levenberg-Marquardt over the whole trajectory in reduced coordinates (stands where IPOPT is called,
 * acinoset_opt.py:611-617).  Same algorithm as the HIP product (DESIGN.md "Solver"). */
#define CPO_DIAG_FLOOR(N) ((N) < 4 ? 0.1 : 1e-12)
typedef struct { const double* tau; double* rcb; double* mu_keep; } shutter_t;
static cpe_status solve_impl2(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o,
                              const cpe_priors* pr, int N, const double* q_init, const double* meas,
                              const double* weight, double* q, double* dq, double* ddq, double* positions,
                              double* meas_err, cpe_stats* st, kin_t* K, const shutter_t* SH);
static cpe_status solve_impl(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o,
                             const cpe_priors* pr, int N, const double* q_init, const double* meas,
                             const double* weight, double* q, double* dq, double* ddq, double* positions,
                             double* meas_err, cpe_stats* st, kin_t* K) {
    return solve_impl2(s, cams, C, o, pr, N, q_init, meas, weight, q, dq, ddq, positions, meas_err, st, K, NULL);
}
static cpe_status solve_impl2(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o,
                              const cpe_priors* pr, int N, const double* q_init, const double* meas,
                              const double* weight, double* q, double* dq, double* ddq, double* positions,
                              double* meas_err, cpe_stats* st, kin_t* K, const shutter_t* SH) {
    ctx_t x; ctx_init(&x, s, cams, C, o, pr);
    x.kin = K;
    if (SH) { x.tau = SH->tau; x.rcb = SH->rcb; }
    int nq = x.nq, nu = x.nu, ns = x.ns, L = s->n_markers;
    int bw = 3; if (pr && pr->lr_window > bw) bw = pr->lr_window;
    int kd = (bw + 1) * nu - 1, n_tot = N * nu;
    double* qc = (double*)malloc(sizeof(double) * N * ns);      /* states [N][ns]: Euler q + leg angles */
    double* qt = (double*)malloc(sizeof(double) * N * ns);
    double* g = (double*)malloc(sizeof(double) * n_tot);
    double* ab = (double*)malloc(sizeof(double) * (size_t)n_tot * (kd + 1));
    double* abf = (double*)malloc(sizeof(double) * (size_t)n_tot * (kd + 1));
    double* abm = K ? (double*)malloc(sizeof(double) * (size_t)n_tot * (kd + 1)) : NULL;
    const double* abp = K ? abm : ab;                   /* matrix of the quadratic model whose decrease is `pred` */
    double* dl = (double*)malloc(sizeof(double) * n_tot);
    for (int n = 0; n < N; n++) state_from_q(&x, q_init + (size_t)n * nq, qc + (size_t)n * ns);
    costs_t cc, ctr;
    double* mu = (SH && SH->mu_keep) ? SH->mu_keep : (double*)calloc((size_t)N * s->n_bounds * 2 + 1, sizeof(double));
    seq_eval(&x, N, kd, qc, meas, weight, mu, &cc, g, ab);
    const double NU0 = 8.0;   /* first rejection multiplies lambda by 8, then 16, 32, ... (Nielsen's rule starts at 2: measured 6 % more iterations) */
    double lam = o->lambda0, nu_f = NU0;
    int it = 0, status = CPE_MAX_ITER, outer = 0;
    if (!isfinite(cc.total)) status = CPE_NUMERICAL;
    while (status == CPE_MAX_ITER && it < o->max_iter) {
        it++;
        int inner_done = 0;
        /* (H + lam diag(H)) dl = -g */
        memcpy(abf, ab, sizeof(double) * (size_t)n_tot * (kd + 1));
        if (K) {            /* physics terms: Schur complement of the node forces, damped in force space with the same lambda */
            kin_add_schur(&x, N, kd, lam, abf);
            memcpy(abm, abf, sizeof(double) * (size_t)n_tot * (kd + 1));        /* the model matrix of this iteration (for pred) */
        }
        /* Marquardt scaling lam * diag(H) with a FLOOR on the diagonal for sequences without a motion term (N < 4: no third difference exists): a
         * coordinate that no measurement of its frame sees (weights zero, outliers in the flat part of the loss) is tied to nothing there, its diagonal
         * is ~0 and scaling alone leaves it undamped -- steps of 30 rad in such coordinates made sequences of 2 and 3 frames creep to the iteration
         * limit (round 2's three expected failures).  From 4 frames on the floor is the old guard against an exact zero (a larger one slows the
         * weakly observed directions next to the Euler pole: 200 instead of 38 iterations).  Same rule in csrc/cpe_solver.hip.inc */
        for (int i = 0; i < n_tot; i++) { double d = abf[(size_t)i * (kd + 1)]; abf[(size_t)i * (kd + 1)] = d + lam * fmax(d, CPO_DIAG_FLOOR(N)); }
        if (band_cholesky(n_tot, kd, abf)) { lam *= 10; if (lam > 1e12) { status = CPE_NUMERICAL; } continue; }
        for (int i = 0; i < n_tot; i++) dl[i] = -g[i];
        band_solve(n_tot, kd, abf, dl);
        /* predicted reduction: -g.dl - 0.5 dl.H.dl */
        double gd = 0, dHd = 0, maxstep = 0;
        for (int i = 0; i < n_tot; i++) {
            gd += g[i] * dl[i];
            double hv = 0;
            int k0 = i - kd > 0 ? i - kd : 0, k1 = i + kd < n_tot - 1 ? i + kd : n_tot - 1;
            for (int k = k0; k <= i; k++) hv += abp[(size_t)i * (kd + 1) + (i - k)] * dl[k];
            for (int k = i + 1; k <= k1; k++) hv += abp[(size_t)k * (kd + 1) + (k - i)] * dl[k];
            dHd += dl[i] * hv;
            if (fabs(dl[i]) > maxstep) maxstep = fabs(dl[i]);
        }
        double pred = -gd - 0.5 * dHd;
        memcpy(qt, qc, sizeof(double) * N * ns);
        for (int n = 0; n < N; n++) for (int k = 0; k < nu; k++) state_add(&x, qt + (size_t)n * ns, k, dl[n * nu + k]);
        if (K) K->write_try = 1;
        seq_eval(&x, N, kd, qt, meas, weight, mu, &ctr, NULL, NULL);
        if (K) K->write_try = 0;
        double act = cc.total - ctr.total;
        double gain = pred > 0 ? act / pred : -1;
        if (getenv("CPO_DEBUG")) fprintf(stderr, "it %3d cost %.10f trial %.10f pred %.3e act %.3e gain %.3f lam %.2e step %.2e viol %.2e\n", it, cc.total, ctr.total, pred, act, gain, lam, maxstep, cc.maxviol);
        if (isfinite(ctr.total) && act > 0 && gain > 1e-4) {
            double rel = act / (fabs(cc.total) + 1e-30);
            memcpy(qc, qt, sizeof(double) * N * ns);
            if (K) { double* t_ = K->f_cur; K->f_cur = K->f_try; K->f_try = t_; }      /* the trial's node forces become the warm start */
            seq_eval(&x, N, kd, qc, meas, weight, mu, &cc, g, ab);
            double f = 1 - (2 * gain - 1) * (2 * gain - 1) * (2 * gain - 1);
            lam *= f > 1.0 / 3 ? f : 1.0 / 3; nu_f = NU0;
            if (lam < 1e-12) lam = 1e-12;
            if (maxstep < o->tol_step || rel < o->tol_cost) inner_done = 1;
        } else {
            lam *= nu_f; nu_f *= 2;
            if (maxstep < o->tol_step * 1e-2 || lam > 1e14) inner_done = 1; /* step collapsed: stationary to tolerance */
        }
        if (inner_done) {
            if (cc.maxviol > o->bound_tol && outer < o->max_outer) {
                /* multiplier update mu <- max(0, mu + kappa * violation), then re-baseline the cost */
                outer++;
                for (int n = 0; n < N; n++)
                    for (int b = 0; b < s->n_bounds; b++) {
                        int ia = s->bound_a[b], ib = s->bound_b[b];
                        double qv[CPE_MAX_NQ];
                        cost_view(&x, qc + (size_t)n * ns, NULL, qv, NULL);          /* the ranges act on the cost view (leg pitch = theta_B + alpha) */
                        double v = qv[ia] - (ib >= 0 ? qv[ib] : 0.0);
                        double* m2 = mu + ((size_t)n * s->n_bounds + b) * 2;
                        double t_up = m2[0] + o->bound_penalty * (v - s->bound_up[b]), t_lo = m2[1] + o->bound_penalty * (s->bound_lo[b] - v);
                        m2[0] = t_up > 0 ? t_up : 0; m2[1] = t_lo > 0 ? t_lo : 0;
                    }
                if (K) K->update_mu = 1;          /* force / height / slip multipliers: updated inside the evaluation, node by node */
                seq_eval(&x, N, kd, qc, meas, weight, mu, &cc, g, ab);
                if (lam > 1e-3) lam = 1e-3;
                nu_f = NU0;
            } else status = CPE_OK;
        }
    }
    if (K) seq_eval(&x, N, kd, qc, meas, weight, mu, &cc, NULL, NULL);      /* node forces and statistics of the FINAL iterate (the last evaluation may have been a rejected trial) */
    /* outputs as CheetahEstimator.save writes them (acinoset_opt.py:289-361) */
    for (int n = 0; n < N; n++) memcpy(q + (size_t)n * nq, qc + (size_t)n * ns, sizeof(double) * nq);
    if (dq && ddq) cpo_derivatives(nq, N, o->h, q, dq, ddq);
    double maxc = 0;
    for (int n = 0; n < N; n++) {
        double pos[CPE_MAX_MARKERS * 3], cv[64];
        cpo_markers(s, q + (size_t)n * nq, pos);
        if (positions) memcpy(positions + (size_t)n * L * 3, pos, sizeof(double) * L * 3);
        int nc = cpo_constraints(s, q + (size_t)n * nq, cv, NULL);
        for (int i = 0; i < nc; i++) if (fabs(cv[i]) > maxc) maxc = fabs(cv[i]);
        if (meas_err)
            for (int c = 0; c < C; c++) for (int l = 0; l < L; l++) {
                double uv[2], pt[3] = {pos[3 * l], pos[3 * l + 1], pos[3 * l + 2]};
                if (x.tau && n >= 2) {
                    double ta = x.tau[c], ih = 1.0 / o->h, al = ta * ih + ta * ta * ih * ih, be = -ta * ih - 2 * ta * ta * ih * ih, ga = ta * ta * ih * ih;
                    for (int d = 0; d < 3; d++) pt[d] += al * q[(size_t)n * nq + d] + be * q[(size_t)(n - 1) * nq + d] + ga * q[(size_t)(n - 2) * nq + d];
                }
                cpo_project(&cams[c], pt, uv, NULL);
                size_t b = ((size_t)(n * C + c) * L + l) * 2;
                meas_err[b] = uv[0] - meas[b]; meas_err[b + 1] = uv[1] - meas[b + 1];
            }
    }
    if (st) {
        st->status = status; st->iterations = it; st->lambda = lam; st->max_constraint = maxc;
        st->max_bound_violation = cc.maxviol; st->outer = outer;
        st->cost_meas = cc.meas; st->cost_model = cc.model; st->cost_pose = cc.pose; st->cost_motion = cc.motion;
        st->cost = o->cost_scale * (cc.meas + cc.model + cc.pose + cc.motion);
    }
    free(qc); free(qt); free(g); free(ab); free(abf); free(dl); if (!(SH && SH->mu_keep)) free(mu); if (abm) free(abm);
    return status;
}

/* Anderson mixing of the delay fixed-point iteration tau <- tau + step(tau) (memory CPO_AA_MEM): the plain iteration contracts
 * slowly in the direction where all delays move together and the trajectory shifts in time to make up for it */
#define CPO_AA_MEM 4
typedef struct { int k; double X[CPO_AA_MEM + 1][CPE_MAX_CAMS], F[CPO_AA_MEM + 1][CPE_MAX_CAMS], fn_prev; } anderson_t;
static void anderson_next(anderson_t* A, int n, const double* x, const double* f, double* xn) {
    double fn = 0;
    for (int i = 0; i < n; i++) fn += f[i] * f[i];
    if (A->k > 0 && fn > A->fn_prev) A->k = 0;                      /* residual grew: drop the history */
    A->fn_prev = fn;
    int slot = A->k % (CPO_AA_MEM + 1), mk = A->k < CPO_AA_MEM ? A->k : CPO_AA_MEM;
    for (int i = 0; i < n; i++) { A->X[slot][i] = x[i]; A->F[slot][i] = f[i]; }
    for (int i = 0; i < n; i++) xn[i] = x[i] + f[i];
    if (mk > n) mk = n;
    if (mk > 0) {
        double dX[CPO_AA_MEM][CPE_MAX_CAMS], dF[CPO_AA_MEM][CPE_MAX_CAMS], M[CPO_AA_MEM][CPO_AA_MEM], r[CPO_AA_MEM], tr = 0;
        for (int j = 0; j < mk; j++) {
            int s1 = (A->k - j) % (CPO_AA_MEM + 1), s0 = (A->k - j - 1) % (CPO_AA_MEM + 1);
            for (int i = 0; i < n; i++) { dX[j][i] = A->X[s1][i] - A->X[s0][i]; dF[j][i] = A->F[s1][i] - A->F[s0][i]; }
        }
        for (int a = 0; a < mk; a++) {
            for (int b = 0; b < mk; b++) { double v = 0; for (int i = 0; i < n; i++) v += dF[a][i] * dF[b][i]; M[a][b] = v; }
            double v = 0; for (int i = 0; i < n; i++) v += dF[a][i] * f[i];
            r[a] = v; tr += M[a][a];
        }
        for (int a = 0; a < mk; a++) M[a][a] += 1e-10 * tr / mk + 1e-300;
        int ok = 1;                                                  /* Cholesky of the mk x mk normal matrix */
        for (int j = 0; j < mk && ok; j++) {
            double d = M[j][j];
            for (int k = 0; k < j; k++) d -= M[j][k] * M[j][k];
            if (!(d > 0)) { ok = 0; break; }
            d = sqrt(d); M[j][j] = d;
            for (int i = j + 1; i < mk; i++) { double v = M[i][j]; for (int k = 0; k < j; k++) v -= M[i][k] * M[j][k]; M[i][j] = v / d; }
        }
        if (ok) {
            for (int i = 0; i < mk; i++) { double v = r[i]; for (int k = 0; k < i; k++) v -= M[i][k] * r[k]; r[i] = v / M[i][i]; }
            for (int i = mk - 1; i >= 0; i--) { double v = r[i]; for (int k = i + 1; k < mk; k++) v -= M[k][i] * r[k]; r[i] = v / M[i][i]; }
            for (int j = 0; j < mk; j++) for (int i = 0; i < n; i++) xn[i] -= r[j] * (dX[j][i] + dF[j][i]);
        }
    }
    A->k++;
}

/* shutter-delay estimation (include/cpe.h, cpe_solve_shutter): trajectory solve with the delays fixed, then one Newton step per
 * delay with the trajectory fixed (same sums as k_shutter_step), the sequence of delays accelerated by Anderson mixing */
cpe_status cpo_solve_shutter(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr, int N,
                             const double* q_init, const double* meas, const double* weight, double tau_bound, int max_rounds, double tol_tau,
                             double* q, double* dq, double* ddq, double* positions, double* meas_err, double* tau, cpe_stats* st, int* rounds_out) {
    int nq = NQ(s), iters = 0, round = 0;
    double* rcb = (double*)calloc((size_t)N * C * 9 + 1, sizeof(double));
    double* mu = (double*)calloc((size_t)N * s->n_bounds * 2 + 1, sizeof(double));
    double* qi = (double*)malloc(sizeof(double) * (size_t)N * nq);
    memcpy(qi, q_init, sizeof(double) * (size_t)N * nq);
    for (int c = 0; c < C; c++) tau[c] = 0;
    shutter_t SH = {tau, rcb, mu};
    anderson_t AA; AA.k = 0; AA.fn_prev = 0;
    cpe_status rc = CPE_OK;
    for (; round < max_rounds; round++) {
        rc = solve_impl2(s, cams, C, o, pr, N, qi, meas, weight, q, NULL, NULL, NULL, NULL, st, NULL, &SH);
        iters += st->iterations;
        memcpy(qi, q, sizeof(double) * (size_t)N * nq);
        double worst = 0, ih = 1.0 / o->h, step[CPE_MAX_CAMS], tn[CPE_MAX_CAMS];
        step[0] = 0;
        for (int c = 1; c < C; c++) {
            double ta = tau[c], g = 0, hh = 0;
            for (int n = 2; n < N; n++) {
                const double* r = rcb + ((size_t)n * C + c) * 9;
                double t3[3];
                for (int d = 0; d < 3; d++) {
                    double x0 = q[(size_t)n * nq + d], x1 = q[(size_t)(n - 1) * nq + d], x2 = q[(size_t)(n - 2) * nq + d];
                    t3[d] = (x0 - x1) * ih + 2 * ta * (x0 - 2 * x1 + x2) * ih * ih;
                }
                g += t3[0] * r[0] + t3[1] * r[1] + t3[2] * r[2];
                hh += t3[0] * (r[3] * t3[0] + r[4] * t3[1] + r[5] * t3[2]) + t3[1] * (r[4] * t3[0] + r[6] * t3[1] + r[7] * t3[2]) + t3[2] * (r[5] * t3[0] + r[7] * t3[1] + r[8] * t3[2]);
            }
            step[c] = hh > 0 ? -g / hh : 0.0;
            if (fabs(step[c]) > worst) worst = fabs(step[c]);
            if (getenv("CPO_DEBUG")) fprintf(stderr, "round %d cam %d tau %.6g g %.6g h %.6g\n", round, c, ta, g, hh);
        }
        if (worst == 0.0) break;                                       /* no node carries a displacement (N < 3), or nothing to move */
        anderson_next(&AA, C, tau, step, tn);
        /* the plain step underestimates the distance to the fixed point by the contraction factor (the trajectory has not followed yet):
         * the test is on the mixed update */
        worst = 0;
        for (int c = 1; c < C; c++) {
            double v = tn[c] > tau_bound ? tau_bound : (tn[c] < -tau_bound ? -tau_bound : tn[c]);
            if (fabs(v - tau[c]) > worst) worst = fabs(v - tau[c]);
            tau[c] = v;
        }
        if (worst < tol_tau) { round++; break; }
    }
    rc = solve_impl2(s, cams, C, o, pr, N, qi, meas, weight, q, dq, ddq, positions, meas_err, st, NULL, &SH);
    st->iterations += iters;
    if (rounds_out) *rounds_out = round;
    free(rcb); free(mu); free(qi);
    return rc;
}

cpe_status cpo_solve(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o,
                     const cpe_priors* pr, int N, const double* q_init, const double* meas,
                     const double* weight, double* q, double* dq, double* ddq, double* positions,
                     double* meas_err, cpe_stats* st) {
    return solve_impl(s, cams, C, o, pr, N, q_init, meas, weight, q, dq, ddq, positions, meas_err, st, NULL);
}

/* tests: cpo_kinetic_nodes also copies the Jacobian [nrow][84] of node `node` to J (NULL switches it off); numeric: fourth-order differences instead of the closed form */
static double* g_dbgJ = NULL; static int g_dbgJ_node = -1, g_numeric_jacobian = 0;
void cpo_set_debug_jacobian(double* J, int node) { g_dbgJ = J; g_dbgJ_node = node; }
void cpo_set_numeric_jacobian(int on) { g_numeric_jacobian = on; }      /* applies to every later cpo_kinetic_* / cpo_solve_kinetic* call */

void cpo_default_kinetic_options(cpe_kinetic_options* o, double fps, int kinetic_dataset) {
    /* o->dyn (inertias, feet, motors) is the caller's */
    o->w_slack = 10e3; o->w_torque = 1.0; o->w_smooth = 0.1 / (fps * fps); o->friction = 0.8; o->force_max = 5.0; o->grfz_min = 0.01;
    o->foot_height_tol = kinetic_dataset ? 0.03 : 0.1; o->foot_height_min = 0.0; o->ground_height = 0.0; o->slip_max = 1.0; o->zvel_max = kinetic_dataset ? 1.0 : 0.0;
    o->slack_lo = -2.0; o->slack_hi = 2.0; o->kappa_slack = 1e6;
    o->reg_force = 1e-4; o->kappa_force = 1e5; o->kappa_height = 1e6; o->kappa_slip = 1e2; o->lm_force_damping = 10.0; o->lm_wall_damping = 10.0; o->inner_iterations = 30; o->_pad = 0;
}

/* objective of the physics-based model at a point (multipliers zero), its reduced gradient and, optionally, the band matrix:
 * for finite-difference checks of kin_term (tests) */
double cpo_kinetic_objective(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr,
                             const cpe_kinetic_options* ko, int N, double* q, const double* meas, const double* weight, const int32_t* stance,
                             double* g /*[N*nu] or NULL*/, double* Hband /*[N*nu][4 nu] or NULL*/, double* terms /*[8] or NULL*/) {
    ctx_t x; ctx_init(&x, s, cams, C, o, pr);
    kin_t K; memset(&K, 0, sizeof(K));
    K.ko = ko; K.stance = stance; K.N = N; K.nm = ko->dyn.n_motors; K.nf = ko->dyn.n_feet; K.h = o->h; K.numeric_jacobian = g_numeric_jacobian;
    for (int j = 0; j < s->n_joints; j++) K.nc += s->joint_kind[j] == CPE_JOINT_REVOLUTE_Y ? 2 : 1;
    K.nlat = K.nm + K.nc + 3 * K.nf;
    double M = 0; for (int i = 0; i < s->n_links; i++) M += s->mass[i];
    K.Mg = M * ko->dyn.eom.gravity;
    K.f_cur = (double*)calloc((size_t)N * K.nlat + 1, sizeof(double)); K.f_try = (double*)calloc((size_t)N * K.nlat + 1, sizeof(double));
    K.mu = (double*)calloc((size_t)N * K.nf * KIN_MU + 1, sizeof(double)); K.mu_slack = (double*)calloc((size_t)N * CPE_MAX_NQ * 2 + 1, sizeof(double));
    K.pHuu = (double*)malloc(sizeof(double) * (size_t)N * KIN_NC3 * KIN_NC3); K.pHfu = (double*)malloc(sizeof(double) * (size_t)N * CPE_KIN_MAXLAT * KIN_NC3);
    K.pHff = (double*)malloc(sizeof(double) * (size_t)N * CPE_KIN_MAXLAT * CPE_KIN_MAXLAT); K.pna = (int*)calloc(N + 1, sizeof(int));
    x.kin = &K;
    int kd = 4 * x.nu - 1;
    costs_t ct;
    double* st = (double*)malloc(sizeof(double) * (size_t)N * x.ns);
    for (int n = 0; n < N; n++) state_from_q(&x, q + (size_t)n * x.nq, st + (size_t)n * x.ns);
    double* ab = g ? (Hband ? Hband : (double*)malloc(sizeof(double) * (size_t)N * x.nu * (kd + 1))) : NULL;
    seq_eval(&x, N, kd, st, meas, weight, NULL, &ct, g, ab);
    for (int n = 0; n < N; n++) memcpy(q + (size_t)n * x.nq, st + (size_t)n * x.ns, sizeof(double) * x.nq);
    if (terms) { terms[0] = ct.meas; terms[1] = ct.model; terms[2] = ct.pose; terms[3] = ct.bound; terms[4] = K.torque; terms[5] = K.energy; terms[6] = K.eom; terms[7] = K.al; }
    if (g && ab) kin_add_schur(&x, N, kd, 0.0, ab);           /* exact variable-projection Gauss-Newton matrix */
    if (g && !Hband) free(ab);
    free(st); free(K.f_cur); free(K.f_try); free(K.mu); free(K.mu_slack); free(K.pHuu); free(K.pHfu); free(K.pHff); free(K.pna);
    return ct.total;
}


/* per-node quantities of one evaluation of the physics terms (multipliers zero, cold start), in cpe_eval_kinetic_nodes' layout:
 * f [N][64], stat [N][8], g [N][84], Huu [N][84][84], Hfu [N][64][84], Hff [N][64][64], meta [N][65] */
void cpo_kinetic_nodes(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_kinetic_options* ko, int N,
                       const double* q, const int32_t* stance, double* f, double* stat, double* g, double* Huu, double* Hfu, double* Hff, int32_t* meta) {
    ctx_t x; ctx_init(&x, s, cams, C, o, NULL);
    kin_t K; memset(&K, 0, sizeof(K));
    K.ko = ko; K.stance = stance; K.N = N; K.nm = ko->dyn.n_motors; K.nf = ko->dyn.n_feet; K.h = o->h; K.numeric_jacobian = g_numeric_jacobian;
    for (int j = 0; j < s->n_joints; j++) K.nc += s->joint_kind[j] == CPE_JOINT_REVOLUTE_Y ? 2 : 1;
    K.nlat = K.nm + K.nc + 3 * K.nf;
    double M = 0; for (int i = 0; i < s->n_links; i++) M += s->mass[i];
    K.Mg = M * ko->dyn.eom.gravity;
    K.f_cur = (double*)calloc((size_t)N * K.nlat + 1, sizeof(double)); K.f_try = (double*)calloc((size_t)N * K.nlat + 1, sizeof(double));
    K.mu = (double*)calloc((size_t)N * K.nf * KIN_MU + 1, sizeof(double)); K.mu_slack = (double*)calloc((size_t)N * CPE_MAX_NQ * 2 + 1, sizeof(double));
    K.pHuu = (double*)malloc(sizeof(double) * (size_t)N * KIN_NC3 * KIN_NC3); K.pHfu = (double*)malloc(sizeof(double) * (size_t)N * CPE_KIN_MAXLAT * KIN_NC3);
    K.pHff = (double*)malloc(sizeof(double) * (size_t)N * CPE_KIN_MAXLAT * CPE_KIN_MAXLAT); K.pna = (int*)calloc(N + 1, sizeof(int));
    x.kin = &K;
    K.dbgJ = g_dbgJ; K.dbgJ_node = g_dbgJ_node;
    double* st = (double*)malloc(sizeof(double) * (size_t)N * x.ns);
    for (int n = 0; n < N; n++) state_from_q(&x, q + (size_t)n * x.nq, st + (size_t)n * x.ns);
    int nc3 = KIN_NC3;
    double* gT = (double*)malloc(sizeof(double) * nc3); double* HT = (double*)malloc(sizeof(double) * nc3 * nc3);
    for (int n = 0; n < N; n++) {
        kin_node_t R; memset(&R, 0, sizeof(R));
        memset(gT, 0, sizeof(double) * nc3);
        int na = 0, idx[CPE_KIN_MAXLAT];
        if (n >= 2) {
            kin_term(&x, &K, n, st + (size_t)n * x.ns, st + (size_t)(n - 1) * x.ns, st + (size_t)(n - 2) * x.ns, gT, HT, &R);
            na = K.pna[n];
            for (int m = 0; m < K.nm + K.nc; m++) idx[m] = m;
            int a = K.nm + K.nc;
            for (int k = 0; k < K.nf; k++) if (stance[(size_t)n * K.nf + k]) for (int d = 0; d < 3; d++) idx[a++] = K.nm + K.nc + 3 * k + d;
        }
        if (f) { for (int i = 0; i < 64; i++) f[(size_t)n * 64 + i] = i < K.nlat && n >= 2 ? K.f_cur[(size_t)n * K.nlat + i] : 0.0; }
        if (stat) { double* d = stat + (size_t)n * 8; d[0] = R.eom; d[1] = R.torque; d[2] = R.reg; d[3] = R.energy; d[4] = R.al; d[5] = R.max_slack; d[6] = R.max_base; d[7] = R.max_viol; }
        if (g) memcpy(g + (size_t)n * nc3, gT, sizeof(double) * nc3);
        if (meta) { meta[(size_t)n * 65] = na; for (int i = 0; i < na; i++) meta[(size_t)n * 65 + 1 + i] = idx[i]; }
        if (n < 2) continue;
        if (Huu) memcpy(Huu + (size_t)n * nc3 * nc3, K.pHuu + (size_t)n * nc3 * nc3, sizeof(double) * nc3 * nc3);
        if (Hfu) for (int i = 0; i < na; i++) memcpy(Hfu + ((size_t)n * 64 + i) * nc3, K.pHfu + (size_t)n * CPE_KIN_MAXLAT * nc3 + (size_t)i * nc3, sizeof(double) * nc3);
        if (Hff) for (int i = 0; i < na; i++) for (int j = 0; j < na; j++) Hff[((size_t)n * 64 + i) * 64 + j] = K.pHff[(size_t)n * CPE_KIN_MAXLAT * CPE_KIN_MAXLAT + (size_t)i * na + j];
    }
    free(gT); free(HT); free(st); free(K.f_cur); free(K.f_try); free(K.mu); free(K.mu_slack); free(K.pHuu); free(K.pHfu); free(K.pHff); free(K.pna);
}

/* physics-based trajectory model: include/cpe.h, cpe_solve_kinetic (estimate_kinetics, acinoset_opt.py:693-963) */
cpe_status cpo_solve_kinetic_fixed(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr,
                                   const cpe_kinetic_options* ko, int N, const double* q_init, const double* meas, const double* weight,
                                   const int32_t* stance, const double* grf_fixed, double* q, double* dq, double* ddq, double* positions, double* meas_err,
                                   double* tau, double* lam, double* grf, double* slack, cpe_stats* st, cpe_kinetic_stats* kst);
cpe_status cpo_solve_kinetic(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr,
                             const cpe_kinetic_options* ko, int N, const double* q_init, const double* meas, const double* weight,
                             const int32_t* stance, double* q, double* dq, double* ddq, double* positions, double* meas_err,
                             double* tau, double* lam, double* grf, double* slack, cpe_stats* st, cpe_kinetic_stats* kst) {
    return cpo_solve_kinetic_fixed(s, cams, C, o, pr, ko, N, q_init, meas, weight, stance, NULL, q, dq, ddq, positions, meas_err, tau, lam, grf, slack, st, kst);
}
/* with prescribed net foot forces grf_fixed [N][nf][3] (include/cpe.h, cpe_solve_kinetic_fixed), or NULL */
static cpe_status solve_kinetic_impl(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr,
                                   const cpe_kinetic_options* ko, int N, const double* q_init, const double* meas, const double* weight,
                                   const int32_t* stance, const double* grf_fixed, const double* tau_box, const double* grf_box, double* q, double* dq, double* ddq, double* positions, double* meas_err,
                                   double* tau, double* lam, double* grf, double* slack, cpe_stats* st, cpe_kinetic_stats* kst);
cpe_status cpo_solve_kinetic_fixed(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr,
                                   const cpe_kinetic_options* ko, int N, const double* q_init, const double* meas, const double* weight,
                                   const int32_t* stance, const double* grf_fixed, double* q, double* dq, double* ddq, double* positions, double* meas_err,
                                   double* tau, double* lam, double* grf, double* slack, cpe_stats* st, cpe_kinetic_stats* kst) {
    return solve_kinetic_impl(s, cams, C, o, pr, ko, N, q_init, meas, weight, stance, grf_fixed, NULL, NULL, q, dq, ddq, positions, meas_err, tau, lam, grf, slack, st, kst);
}
/* with the net force of every stance foot boxed: grf_box [N][nf][3][2] = (lower, upper) of (z, x, y) (include/cpe.h, cpe_solve_kinetic_force_box:
 * estimate_kinetics(joint_estimation=False, fix_grf=False), acinoset_opt.py:838-850, lets the forces move within 20 % of a profile) */
cpe_status cpo_solve_kinetic_force_box(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr,
                                   const cpe_kinetic_options* ko, int N, const double* q_init, const double* meas, const double* weight,
                                   const int32_t* stance, const double* grf_box, double* q, double* dq, double* ddq, double* positions, double* meas_err,
                                   double* tau, double* lam, double* grf, double* slack, cpe_stats* st, cpe_kinetic_stats* kst) {
    return solve_kinetic_impl(s, cams, C, o, pr, ko, N, q_init, meas, weight, stance, NULL, NULL, grf_box, q, dq, ddq, positions, meas_err, tau, lam, grf, slack, st, kst);
}
/* with every torque boxed: tau_box [N][n_motors][2] = (lower, upper) (include/cpe.h, cpe_solve_kinetic_bounded: the reference's module-level
 * estimate_grf, acinoset_opt.py:966-1048, bounds the torques to +-10 % of a previous solve) */
cpe_status cpo_solve_kinetic_bounded(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr,
                                   const cpe_kinetic_options* ko, int N, const double* q_init, const double* meas, const double* weight,
                                   const int32_t* stance, const double* tau_box, double* q, double* dq, double* ddq, double* positions, double* meas_err,
                                   double* tau, double* lam, double* grf, double* slack, cpe_stats* st, cpe_kinetic_stats* kst) {
    return solve_kinetic_impl(s, cams, C, o, pr, ko, N, q_init, meas, weight, stance, NULL, tau_box, NULL, q, dq, ddq, positions, meas_err, tau, lam, grf, slack, st, kst);
}
static cpe_status solve_kinetic_impl(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr,
                                   const cpe_kinetic_options* ko, int N, const double* q_init, const double* meas, const double* weight,
                                   const int32_t* stance, const double* grf_fixed, const double* tau_box, const double* grf_box, double* q, double* dq, double* ddq, double* positions, double* meas_err,
                                   double* tau, double* lam, double* grf, double* slack, cpe_stats* st, cpe_kinetic_stats* kst) {
    int nq = NQ(s);
    for (int p = 0; p < nq; p++) if (s->motion_w[p] != 0.0) return CPE_BAD_ARG;      /* the physics replaces the constant-acceleration cost */
    kin_t K; memset(&K, 0, sizeof(K));
    K.ko = ko; K.stance = stance; K.N = N; K.nm = ko->dyn.n_motors; K.nf = ko->dyn.n_feet; K.h = o->h; K.numeric_jacobian = g_numeric_jacobian; K.grf_fix = grf_fixed;
    for (int j = 0; j < s->n_joints; j++) K.nc += s->joint_kind[j] == CPE_JOINT_REVOLUTE_Y ? 2 : 1;
    K.nlat = K.nm + K.nc + 3 * K.nf;
    if (K.nlat > CPE_KIN_MAXLAT) return CPE_BAD_ARG;
    double M = 0; for (int i = 0; i < s->n_links; i++) M += s->mass[i];
    K.Mg = M * ko->dyn.eom.gravity;
    K.f_cur = (double*)calloc((size_t)N * K.nlat + 1, sizeof(double)); K.f_try = (double*)calloc((size_t)N * K.nlat + 1, sizeof(double));
    K.mu = (double*)calloc((size_t)N * K.nf * KIN_MU + 1, sizeof(double)); K.mu_slack = (double*)calloc((size_t)N * CPE_MAX_NQ * 2 + 1, sizeof(double));
    K.grf_box = grf_box; K.tau_box = tau_box; K.mu_tau = (double*)calloc((size_t)N * K.nm * 2 + 1, sizeof(double));
    K.pHuu = (double*)malloc(sizeof(double) * (size_t)N * KIN_NC3 * KIN_NC3); K.pHfu = (double*)malloc(sizeof(double) * (size_t)N * CPE_KIN_MAXLAT * KIN_NC3);
    K.pHff = (double*)malloc(sizeof(double) * (size_t)N * CPE_KIN_MAXLAT * CPE_KIN_MAXLAT); K.pna = (int*)calloc(N + 1, sizeof(int));
    cpe_status rc = solve_impl(s, cams, C, o, pr, N, q_init, meas, weight, q, dq, ddq, positions, meas_err, st, &K);
    /* node forces of the final iterate; slack = the residual of the equations of motion there */
    ctx_t x; ctx_init(&x, s, cams, C, o, pr);
    for (int n = 0; n < N; n++) {
        const double* f = K.f_cur + (size_t)n * K.nlat;
        double tq[CPE_MAX_MOTORS], lN[2 * CPE_MAX_JOINTS], g5[20], rho[KIN_MAXROWS];
        kin_split_forces(&K, f, tq, lN, g5);
        if (tau) for (int m = 0; m < K.nm; m++) tau[(size_t)n * K.nm + m] = n >= 2 ? tq[m] : 0.0;
        if (lam) for (int r = 0; r < K.nc; r++) lam[(size_t)n * K.nc + r] = n >= 2 ? f[K.nm + r] : 0.0;
        if (grf) for (int k = 0; k < 5 * K.nf; k++) grf[(size_t)n * 5 * K.nf + k] = n >= 2 ? g5[k] : 0.0;
        if (slack) {
            if (n >= 2) { kin_rho(&x, &K, q + (size_t)n * nq, q + (size_t)(n - 1) * nq, q + (size_t)(n - 2) * nq, f, rho); memcpy(slack + (size_t)n * nq, rho, sizeof(double) * nq); }
            else memset(slack + (size_t)n * nq, 0, sizeof(double) * nq);
        }
    }
    if (kst) {
        kst->cost_torque = K.torque; kst->cost_energy = K.energy; kst->cost_eom = K.eom; kst->max_slack = K.max_slack;
        kst->max_base_rows = K.max_base; kst->max_violation = K.max_viol; kst->inner_max = K.inner_max; kst->_pad = 0;
    }
    free(K.f_cur); free(K.f_try); free(K.mu); free(K.mu_slack); free(K.mu_tau); free(K.pHuu); free(K.pHfu); free(K.pHff); free(K.pna);
    return rc;
}

/* bench.py's multi-thread CPU solve leg: B independent sequences, one cpo_solve each, spread over OpenMP threads (the way a
 * CPU user of the reference would parallelise run_dataset.py's loop over sequences).  Returns the number of threads used;
 * iterations[b] (optional) receives each solve's LM iteration count.  Outputs other than q are not kept. */
int cpo_solve_batch(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr, int B, int N,
                    const double* q_init, const double* meas, const double* weight, double* q, int threads, int* iterations) {
    int nq = NQ(s), L = s->n_markers, used = 1;
    if (threads < 1) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
    {
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; b++) {
            cpe_stats st;
            cpo_solve(s, cams, C, o, pr, N, q_init + (size_t)b * N * nq, meas + (size_t)b * N * C * L * 2, weight + (size_t)b * N * C * L,
                      q + (size_t)b * N * nq, NULL, NULL, NULL, NULL, &st);
            if (iterations) iterations[b] = st.iterations;
        }
    }
    return used;
}

/* reduced gradient / cost at a point (used by tests to check stationarity and by finite-difference checks) */
double cpo_objective(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o,
                     const cpe_priors* pr, int N, double* q /* made consistent with the joint equalities in place */,
                     const double* meas, const double* weight, double* g /*[N*nu] or NULL*/,
                     double* Hband /* [N*nu][kd+1] or NULL */, double* terms /*[5] or NULL*/) {
    ctx_t x; ctx_init(&x, s, cams, C, o, pr);
    int bw = 3; if (pr && pr->lr_window > bw) bw = pr->lr_window;
    int kd = (bw + 1) * x.nu - 1;
    costs_t ct;
    double* ab = NULL;
    double* st = (double*)malloc(sizeof(double) * (size_t)N * x.ns);
    for (int n = 0; n < N; n++) state_from_q(&x, q + (size_t)n * x.nq, st + (size_t)n * x.ns);
    if (g) ab = Hband ? Hband : (double*)malloc(sizeof(double) * (size_t)N * x.nu * (kd + 1));
    seq_eval(&x, N, kd, st, meas, weight, NULL, &ct, g, ab);
    for (int n = 0; n < N; n++) memcpy(q + (size_t)n * x.nq, st + (size_t)n * x.ns, sizeof(double) * x.nq);
    if (g && !Hband) free(ab);
    free(st);
    if (terms) { terms[0] = ct.meas; terms[1] = ct.model; terms[2] = ct.pose; terms[3] = ct.motion; terms[4] = ct.bound; }
    return ct.total;
}

/* cpo_objective with per-camera shutter delays: f, reduced gradient g [N*nu] and d f / d tau [C] (tests: finite differences) */
double cpo_objective_shutter(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr, int N, double* q,
                             const double* meas, const double* weight, const double* tau, double* g, double* gtau) {
    ctx_t x; ctx_init(&x, s, cams, C, o, pr);
    int bw = 3; if (pr && pr->lr_window > bw) bw = pr->lr_window;
    int kd = (bw + 1) * x.nu - 1;
    costs_t ct;
    double* st = (double*)malloc(sizeof(double) * (size_t)N * x.ns);
    double* ab = g ? (double*)malloc(sizeof(double) * (size_t)N * x.nu * (kd + 1)) : NULL;
    double* rcb = g ? (double*)calloc((size_t)N * C * 9 + 1, sizeof(double)) : NULL;
    x.tau = tau; x.rcb = rcb;
    for (int n = 0; n < N; n++) state_from_q(&x, q + (size_t)n * x.nq, st + (size_t)n * x.ns);
    seq_eval(&x, N, kd, st, meas, weight, NULL, &ct, g, ab);
    for (int n = 0; n < N; n++) memcpy(q + (size_t)n * x.nq, st + (size_t)n * x.ns, sizeof(double) * x.nq);
    if (g && gtau)
        for (int c = 0; c < C; c++) {
            double acc = 0, ih = 1.0 / o->h;
            for (int n = 2; n < N; n++)
                for (int d = 0; d < 3; d++) {
                    double x0 = st[(size_t)n * x.ns + d], x1 = st[(size_t)(n - 1) * x.ns + d], x2 = st[(size_t)(n - 2) * x.ns + d];
                    acc += ((x0 - x1) * ih + 2 * tau[c] * (x0 - 2 * x1 + x2) * ih * ih) * rcb[((size_t)n * C + c) * 9 + d];
                }
            gtau[c] = acc;
        }
    free(st); if (ab) free(ab); if (rcb) free(rcb);
    return ct.total;
}

/* per-frame terms in reduced coordinates at Euler q_n (made consistent first): measurement + bounds (+ pose prior)
 * cost[3] = {meas, bound, pose}; g[nu]; B[nu][nu]; Gam[nq][nu] = d Euler / d coordinates (optional) */
void cpo_frame_normal(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, const cpe_priors* pr,
                      double* qn, const double* meas, const double* weight, double* g, double* Bm, double* cost, double* Zout) {
    ctx_t x; ctx_init(&x, s, cams, C, o, pr);
    double st[CPE_MAX_NQ + CPE_MAX_JOINTS];
    state_from_q(&x, qn, st);
    double* Zp = (double*)malloc(sizeof(double) * x.nq * x.nu);
    state_jacobian(&x, st, Zp);
    double cb, cp, vm;
    cost[0] = frame_terms(&x, st, Zp, meas, weight, NULL, g, Bm, &cb, &cp, &vm, NULL, NULL, NULL, NULL);
    cost[1] = cb; cost[2] = cp;
    memcpy(qn, st, sizeof(double) * x.nq);
    if (Zout) memcpy(Zout, Zp, sizeof(double) * x.nq * x.nu);
    free(Zp);
}

/* one reduced coordinate of every frame moved by d (tests: finite differences of cpo_objective) */
void cpo_move_coordinate(const cpe_skeleton* s, int N, double* q, int n, int k, double d) {
    ctx_t x; ctx_init(&x, s, NULL, 0, NULL, NULL);
    double st[CPE_MAX_NQ + CPE_MAX_JOINTS];
    (void)N;
    state_from_q(&x, q + (size_t)n * x.nq, st);
    state_add(&x, st, k, d);
    state_sync(&x, st);
    memcpy(q + (size_t)n * x.nq, st, sizeof(double) * x.nq);
}

/* ------------------------------------------------------------------------------------------------ */
/* per-frame ground-reaction-force fit (acinoset_opt.py:176-270); see include/cpe.h        */
/* ------------------------------------------------------------------------------------------------ */
static void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
/* body-frame angular velocity / acceleration of a link from its ZYX Euler angles, rates and accelerations */
static void body_rates(const double* a, const double* da, const double* dda, double* w, double* al, double Jw[3][3]) {
    double sf = sin(a[0]), cf = cos(a[0]), st = sin(a[1]), ct = cos(a[1]);
    /* columns of d w / d(phi', theta', psi') */
    double c0[3] = {1, 0, 0}, c1[3] = {0, cf, -sf}, c2[3] = {-st, sf * ct, cf * ct};
    double d1[3] = {0, -sf * da[0], -cf * da[0]};
    double d2[3] = {-ct * da[1], cf * da[0] * ct - sf * st * da[1], -sf * da[0] * ct - cf * st * da[1]};
    for (int k = 0; k < 3; k++) {
        w[k] = c0[k] * da[0] + c1[k] * da[1] + c2[k] * da[2];
        al[k] = c0[k] * dda[0] + c1[k] * dda[1] + c2[k] * dda[2] + d1[k] * da[1] + d2[k] * da[2];
        if (Jw) { Jw[k][0] = c0[k]; Jw[k][1] = c1[k]; Jw[k][2] = c2[k]; }
    }
}
/* rows 0-5 of the equations of motion, E[6], and the force matrix A[6][5 n_feet] (columns per foot: z, +x, +y, -x, -y),
 * both divided by M g */
void cpo_grf_terms(const cpe_skeleton* s, const cpe_grf_options* o, const double* q, const double* dq, const double* ddq,
                   double* E, double* A) {
    int nl = s->n_links, nf = o->n_feet, nv = 5 * nf;
    double R[CPE_MAX_LINKS * 9], w[CPE_MAX_LINKS][3], al[CPE_MAX_LINKS][3], oacc[CPE_MAX_LINKS][3], first[CPE_MAX_LINKS][3];
    double Jw[3][3], M = 0, g = o->gravity;
    int root = -1;
    for (int i = 0; i < nl; i++) {
        cpo_rot(q + 3 + 3 * i, R + 9 * i);
        body_rates(q + 3 + 3 * i, dq + 3 + 3 * i, ddq + 3 + 3 * i, w[i], al[i], s->parent[i] < 0 ? Jw : NULL);
        M += s->mass[i];
        if (s->parent[i] < 0) root = i;
    }
    double dRr[3][9];
    cpo_drot(q + 3 + 3 * root, dRr);
    for (int k = 0; k < 6; k++) E[k] = 0;
    for (int k = 0; k < 6 * nv; k++) A[k] = 0;
    /* acceleration of a body-fixed vector v of link i: R (w x (w x v) + alpha x v) */
#define POINT_ACC(i, v, out) do { double t1[3], t2[3], t3[3]; cross3(w[i], v, t1); cross3(w[i], t1, t2); cross3(al[i], v, t3); \
        for (int d_ = 0; d_ < 3; d_++) t2[d_] += t3[d_]; matvec3(R + 9 * (i), t2, out); } while (0)
    for (int i = 0; i < nl; i++) {                       /* parents precede children in the link order */
        int p = s->parent[i];
        if (p < 0) { for (int d = 0; d < 3; d++) { oacc[i][d] = ddq[d]; first[i][d] = s->com[i][d]; } }
        else {
            double a3[3]; POINT_ACC(p, s->attach[i], a3);
            for (int d = 0; d < 3; d++) { oacc[i][d] = oacc[p][d] + a3[d]; first[i][d] = p == root ? s->attach[i][d] : first[p][d]; }
        }
        double ac[3]; POINT_ACC(i, s->com[i], ac);
        double f[3] = {s->mass[i] * (oacc[i][0] + ac[0]), s->mass[i] * (oacc[i][1] + ac[1]), s->mass[i] * (oacc[i][2] + ac[2] + g)};
        for (int d = 0; d < 3; d++) E[d] += f[d];
        for (int a = 0; a < 3; a++) { double v[3]; matvec3(dRr[a], first[i], v); E[3 + a] += f[0] * v[0] + f[1] * v[1] + f[2] * v[2]; }
    }
    {   /* rotation of the root link itself */
        const double* I = o->root_inertia;
        double Iw[3] = {I[0] * w[root][0], I[1] * w[root][1], I[2] * w[root][2]}, wIw[3];
        cross3(w[root], Iw, wIw);
        double tq[3] = {I[0] * al[root][0] + wIw[0], I[1] * al[root][1] + wIw[1], I[2] * al[root][2] + wIw[2]};
        for (int a = 0; a < 3; a++) E[3 + a] += tq[0] * Jw[0][a] + tq[1] * Jw[1][a] + tq[2] * Jw[2][a];
    }
    static const double D[5][3] = {{0, 0, 1}, {1, 0, 0}, {0, 1, 0}, {-1, 0, 0}, {0, -1, 0}};
    for (int f = 0; f < nf; f++) {
        int lk = s->marker_link[o->foot_marker[f]];
        for (int k = 0; k < 5; k++) {
            int col = 5 * f + k;
            for (int d = 0; d < 3; d++) A[d * nv + col] = D[k][d];
            for (int a = 0; a < 3; a++) {
                double v[3]; matvec3(dRr[a], lk == root ? s->marker_off[o->foot_marker[f]] : first[lk], v);
                A[(3 + a) * nv + col] = v[0] * D[k][0] + v[1] * D[k][1] + v[2] * D[k][2];
            }
        }
    }
    for (int k = 0; k < 6; k++) E[k] /= M * g;
#undef POINT_ACC
}

/* all nq rows of d/dt dL/dq' - dL/dq (include/cpe.h, cpe_eom_rows) */
void cpo_eom_rows(const cpe_skeleton* s, const cpe_eom_options* o, const double* q, const double* dq, const double* ddq, double* E) {
    int nl = s->n_links;
    double R[CPE_MAX_LINKS * 9], w[CPE_MAX_LINKS][3], al[CPE_MAX_LINKS][3], oacc[CPE_MAX_LINKS][3], f[CPE_MAX_LINKS][3], Fs[CPE_MAX_LINKS][3];
    double Jw[CPE_MAX_LINKS][3][3], g = o->gravity;
    for (int i = 0; i < nl; i++) {
        cpo_rot(q + 3 + 3 * i, R + 9 * i);
        body_rates(q + 3 + 3 * i, dq + 3 + 3 * i, ddq + 3 + 3 * i, w[i], al[i], Jw[i]);
    }
#define POINT_ACC(i, v, out) do { double t1[3], t2[3], t3[3]; cross3(w[i], v, t1); cross3(w[i], t1, t2); cross3(al[i], v, t3); \
        for (int d_ = 0; d_ < 3; d_++) t2[d_] += t3[d_]; matvec3(R + 9 * (i), t2, out); } while (0)
    for (int i = 0; i < nl; i++) {
        int p = s->parent[i];
        if (p < 0) for (int d = 0; d < 3; d++) oacc[i][d] = ddq[d];
        else { double a3[3]; POINT_ACC(p, s->attach[i], a3); for (int d = 0; d < 3; d++) oacc[i][d] = oacc[p][d] + a3[d]; }
        double ac[3]; POINT_ACC(i, s->com[i], ac);
        for (int d = 0; d < 3; d++) { f[i][d] = s->mass[i] * (oacc[i][d] + ac[d] + (d == 2 ? g : 0.0)); Fs[i][d] = f[i][d]; }
    }
#undef POINT_ACC
    for (int i = nl - 1; i >= 0; i--) if (s->parent[i] >= 0) for (int d = 0; d < 3; d++) Fs[s->parent[i]][d] += Fs[i][d];
    for (int i = 0; i < nl; i++) {
        if (s->parent[i] < 0) for (int d = 0; d < 3; d++) E[d] = Fs[i][d];
        double dR[3][9];
        cpo_drot(q + 3 + 3 * i, dR);
        const double* I = o->link_inertia[i];
        double Iw[3] = {I[0] * w[i][0], I[1] * w[i][1], I[2] * w[i][2]}, wIw[3];
        cross3(w[i], Iw, wIw);
        double tq[3] = {I[0] * al[i][0] + wIw[0], I[1] * al[i][1] + wIw[1], I[2] * al[i][2] + wIw[2]};
        for (int a = 0; a < 3; a++) {
            double v[3], e;
            matvec3(dR[a], s->com[i], v);
            e = f[i][0] * v[0] + f[i][1] * v[1] + f[i][2] * v[2];
            for (int c = 0; c < nl; c++)
                if (s->parent[c] == i) { matvec3(dR[a], s->attach[c], v); e += Fs[c][0] * v[0] + Fs[c][1] * v[1] + Fs[c][2] * v[2]; }
            e += tq[0] * Jw[i][0][a] + tq[1] * Jw[i][1][a] + tq[2] * Jw[i][2][a];
            E[3 + 3 * i + a] = e;
        }
    }
}

/* generalised forces Q[nq] of the physics-based model (include/cpe.h, cpe_eom_residual); tau / lam / grf may be NULL */
void cpo_dyn_forces(const cpe_skeleton* s, const cpe_dyn_options* o, const double* q, const double* tau, const double* lam,
                    const double* grf, double* Q) {
    int nl = s->n_links, nq = NQ(s);
    double R[CPE_MAX_LINKS * 9], dR[CPE_MAX_LINKS][3][9], M = 0, g = o->eom.gravity;
    for (int i = 0; i < nl; i++) { cpo_rot(q + 3 + 3 * i, R + 9 * i); cpo_drot(q + 3 + 3 * i, dR[i]); M += s->mass[i]; }
    for (int p = 0; p < nq; p++) Q[p] = 0;
    static const double D[5][3] = {{0, 0, 1}, {1, 0, 0}, {0, 1, 0}, {-1, 0, 0}, {0, -1, 0}};
    if (grf)
        for (int f = 0; f < o->n_feet; f++) {
            double F[3] = {0, 0, 0};
            for (int k = 0; k < 5; k++) for (int d = 0; d < 3; d++) F[d] += M * g * grf[5 * f + k] * D[k][d];
            for (int d = 0; d < 3; d++) Q[d] += F[d];
            /* chain of the foot marker: marker link, then up through the parents; vector on each link */
            int l = o->foot_marker[f], k = s->marker_link[l];
            const double* v = s->marker_off[l];
            while (k >= 0) {
                for (int a = 0; a < 3; a++) { double t[3]; matvec3(dR[k][a], v, t); Q[3 + 3 * k + a] += F[0] * t[0] + F[1] * t[1] + F[2] * t[2]; }
                v = s->attach[k]; k = s->parent[k];
            }
        }
    if (tau)
        for (int m = 0; m < o->n_motors; m++) {
            int a1 = o->motor_first[m], a2 = o->motor_second[m], ax = o->motor_axis[m];
            double T[3];
            for (int d = 0; d < 3; d++) T[d] = M * g * tau[m] * R[9 * a1 + 3 * d + ax];
            for (int side = 0; side < 2; side++) {
                int i = side == 0 ? a2 : a1; double sg = side == 0 ? 1.0 : -1.0;
                double w[3], al[3], Jw[3][3], zero[3] = {0, 0, 0};
                body_rates(q + 3 + 3 * i, zero, zero, w, al, Jw);
                for (int a = 0; a < 3; a++) {
                    double col[3] = {Jw[0][a], Jw[1][a], Jw[2][a]}, wc[3];
                    matvec3(R + 9 * i, col, wc);                                   /* d w_world / d q'_a */
                    Q[3 + 3 * i + a] += sg * (T[0] * wc[0] + T[1] * wc[1] + T[2] * wc[2]);
                }
            }
        }
    if (lam) {
        double c[64], Cq[64 * CPE_MAX_NQ];
        int nc = cpo_constraints(s, q, c, Cq);
        for (int r = 0; r < nc; r++) for (int p = 0; p < nq; p++) Q[p] += Cq[r * nq + p] * lam[r];
    }
}
void cpo_eom_residual(const cpe_skeleton* s, const cpe_dyn_options* o, const double* q, const double* dq, const double* ddq,
                      const double* tau, const double* lam, const double* grf, double* res) {
    double Q[CPE_MAX_NQ];
    cpo_eom_rows(s, &o->eom, q, dq, ddq, res);
    cpo_dyn_forces(s, o, q, tau, lam, grf, Q);
    for (int p = 0; p < NQ(s); p++) res[p] -= Q[p];
}

/* projection of y0[5] = (z, x0..x3) onto {0 <= . <= fmax, sum x <= mu z}: clamp(y0 - lam n), n = (-mu, 1, 1, 1, 1), lam >= 0 by bisection */
static void grf_project(double* y, double mu, double fmax) {
    double y0[5]; memcpy(y0, y, sizeof(y0));
    double lo = 0, hi = 0;
#define CL(v) ((v) < 0 ? 0 : ((v) > fmax ? fmax : (v)))
#define GFUN(l) (CL(y0[1] - (l)) + CL(y0[2] - (l)) + CL(y0[3] - (l)) + CL(y0[4] - (l)) - mu * CL(y0[0] + mu * (l)))
    if (GFUN(0.0) > 0) {
        hi = y0[1];                                   /* at lam = max_k x_k every x is clamped to 0: g <= 0 */
        for (int k = 2; k < 5; k++) if (y0[k] > hi) hi = y0[k];
        for (int it = 0; it < 50; it++) { double mid = 0.5 * (lo + hi); if (GFUN(mid) > 0) lo = mid; else hi = mid; }
    }
    y[0] = CL(y0[0] + mu * hi);
    for (int k = 1; k < 5; k++) y[k] = CL(y0[k] - hi);
#undef GFUN
#undef CL
}

/* one frame: forces y[5 n_feet] (z, +x, +y, -x, -y per foot), residual res[6] (units of M g) */
void cpo_grf_fit_frame(const cpe_skeleton* s, const cpe_grf_options* o, const double* q, const double* dq, const double* ddq,
                       const int32_t* contact, double* y, double* res) {
    int nf = o->n_feet, nv = 5 * nf;
    double E[6], A[6 * 20], v[20], yp[20], gr[20];
    cpo_grf_terms(s, o, q, dq, ddq, E, A);
    double L = o->regularisation; int any = 0;
    for (int f = 0; f < nf; f++) if (contact[f]) { any = 1; for (int k = 0; k < 5; k++) for (int r = 0; r < 6; r++) L += A[r * nv + 5 * f + k] * A[r * nv + 5 * f + k]; }
    for (int c = 0; c < nv; c++) { y[c] = 0; v[c] = 0; }
    if (any)
        for (int it = 0; it < o->iterations; it++) {
            double r6[6];
            for (int r = 0; r < 6; r++) { double a = E[r]; for (int c = 0; c < nv; c++) a -= A[r * nv + c] * v[c]; r6[r] = a; }
            for (int c = 0; c < nv; c++) { double a = o->regularisation * v[c]; for (int r = 0; r < 6; r++) a -= A[r * nv + c] * r6[r]; gr[c] = a; }
            memcpy(yp, y, sizeof(double) * nv);
            for (int f = 0; f < nf; f++) {
                if (!contact[f]) { for (int k = 0; k < 5; k++) y[5 * f + k] = 0; continue; }
                for (int k = 0; k < 5; k++) y[5 * f + k] = v[5 * f + k] - gr[5 * f + k] / L;
                grf_project(y + 5 * f, o->friction_ratio, o->force_max);
            }
            double beta = (double)it / (double)(it + 3);
            for (int c = 0; c < nv; c++) v[c] = y[c] + beta * (y[c] - yp[c]);
        }
    if (res) for (int r = 0; r < 6; r++) { double a = E[r]; for (int c = 0; c < nv; c++) a -= A[r * nv + c] * y[c]; res[r] = a; }
}

void cpo_default_options(cpe_options* o) {
    o->h = 1.0 / 120; o->loss_a = 3; o->loss_b = 10; o->loss_c = 20; o->cost_scale = 1e-3;
    o->bound_penalty = 1e4; o->bound_tol = 1e-6; o->lambda0 = 1e-4; o->tol_step = 1e-8; o->tol_cost = 1e-9;
    o->max_iter = 200; o->curvature = 0; o->max_outer = 8;
}
