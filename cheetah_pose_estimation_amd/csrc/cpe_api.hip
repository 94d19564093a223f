// cpe_api.hip -- C ABI (include/cpe.h) over the HIP kernels.  gfx950 only, no CPU fallback:
// every entry point fails with CPE_NO_DEVICE / CPE_HIP_ERROR when the GPU path is unavailable.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

#include "cpe_kernels.hip"

static thread_local std::string g_err;
static cpe_status fail(cpe_status s, const std::string& m) { g_err = m; return s; }
#define HIPCHK(x)                                                                                         \
    do {                                                                                                  \
        hipError_t e_ = (x);                                                                              \
        if (e_ != hipSuccess) return fail(CPE_HIP_ERROR, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct cpe_handle {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    DevModel hm;                 // host copy
    DevModel* dm = nullptr;      // device copy
    cpe_options opts;
    // solver workspace
    size_t ws_frames = 0; int ws_B = 0;
    double *qbuf = nullptr, *gbuf = nullptr, *Bbuf = nullptr, *costbuf = nullptr, *Lbuf = nullptr, *zbuf = nullptr,
           *gtbuf = nullptr, *dgbuf = nullptr, *cmax = nullptr, *mu = nullptr, *gambuf = nullptr;
    SeqState* st = nullptr;
    int* flag = nullptr;
    int* act = nullptr;          // [ws_B] sequences of the current launch window (written by k_build_act)
    int* n_act = nullptr;        // device word: entries of act in use
    int* poll_host = nullptr;    // pinned host memory: two snapshots of n_act, read without draining the stream
    hipEvent_t poll_ev[2] = {nullptr, nullptr};
    hipEvent_t order_ev = nullptr;   // cpe_stream_wait / cpe_stream_signal
    // per-kernel device time of the last cpe_solve (cpe_profile_enable / cpe_profile_get)
    bool prof = false;
    double prof_ms[CPE_PROFILE_SLOTS] = {0};
    int64_t prof_n[CPE_PROFILE_SLOTS] = {0};
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_spans;
    cpe_eom_options* eom = nullptr;   // device copy of the last cpe_eom_rows options
    cpe_dyn_options* dyn = nullptr;   // device copy of the last cpe_eom_residual options
    // learned priors (config 3)
    DevPriors* pri = nullptr;    // device copy, nullptr without priors
    int gmm_k = 0, gmm_dim = 0, lr_window = 0;
    double* Hlr = nullptr;       // [2][F][pb][nu*nu] off-diagonal Gauss-Newton blocks of the autoregressive prior
    // physics-based model (cpe_solve_kinetic): device options and workspace
    DevKin* dk = nullptr; DevKin hk;
    size_t kws_frames = 0;
    double *kmut = nullptr;       // multipliers of the torque boxes [F][2 CPE_MAX_MOTORS] (cpe_solve_kinetic_bounded)
    double *kmus = nullptr;       // multipliers of the box on the residual [F][2 CPE_MAX_NQ] (bound_eom_error)
    double *fbuf = nullptr, *kmu = nullptr, *Jbuf = nullptr, *Abuf = nullptr, *pieces = nullptr, *gTb = nullptr, *dstat = nullptr, *slackb = nullptr,
           *Tbuf = nullptr, *gk = nullptr, *Bk = nullptr, *Hk = nullptr;
    int* pmeta = nullptr;
    int pb = 3;                  // half-bandwidth of the normal equations in frames (4 with a window-4 motion prior)
};

struct DevBuf {
    double* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, sizeof(double) * (n ? n : 1)); }
};

static double host_rho0(double a, double b, double c) {
    // rho(0) of acinoset_misc.py:2001-2015
    auto sg = [](double t) { return 1.0 / (1.0 + std::exp(t)); };   // s(t, 0) = 1/(1+e^{t})
    const double sa = sg(a), sb = sg(b), sc = sg(c);
    const double cb = c - b, w = c / cb;
    return (sa - sb) * (-0.5 * a * a) + (sb - sc) * (a * b - 0.5 * a * a + 0.5 * a * cb * (1.0 - w * w)) +
           sc * (a * b - 0.5 * a * a + 0.5 * a * cb);
}

static cpe_status build_model(const cpe_skeleton* s, const cpe_camera* cams, int C, const cpe_options* o, DevModel& m) {
    memset(&m, 0, sizeof(m));
    if (s->n_links < 1 || s->n_links > CPE_MAX_LINKS || s->n_markers < 1 || s->n_markers > CPE_MAX_MARKERS ||
        s->n_joints < 0 || s->n_joints > CPE_MAX_JOINTS || s->n_bounds < 0 || s->n_bounds > CPE_MAX_BOUNDS ||
        C < 1 || C > CPE_MAX_CAMS)
        return fail(CPE_BAD_ARG, "skeleton / camera counts out of range");
    const int nl = s->n_links, nq = 3 + 3 * nl, L = s->n_markers;
    m.nl = nl; m.L = L; m.C = C; m.nq = nq; m.nj = s->n_joints; m.nb = s->n_bounds;
    m.h = o->h; m.ih2 = 1.0 / (o->h * o->h);
    m.loss_a = o->loss_a; m.loss_b = o->loss_b; m.loss_c = o->loss_c;
    m.bound_penalty = o->bound_penalty; m.curvature = o->curvature;
    m.rho0 = host_rho0(o->loss_a, o->loss_b, o->loss_c);
    if (!(o->h > 0)) return fail(CPE_BAD_ARG, "options.h must be > 0");
    double mt = 0;
    for (int i = 0; i < nl; i++) {
        if (s->parent[i] >= i) return fail(CPE_BAD_ARG, "links must be ordered parents first");
        m.parent[i] = s->parent[i];
        for (int d = 0; d < 3; d++) { m.attach[i][d] = s->attach[i][d]; m.com[i][d] = s->com[i][d]; }
        m.mass[i] = s->mass[i]; mt += s->mass[i];
    }
    m.inv_total_mass = mt > 0 ? 1.0 / mt : 0.0;
    for (int c = 0; c < C; c++) m.cam[c] = cams[c];

    // dependent / independent split
    for (int p = 0; p < nq; p++) { m.u_of_q[p] = -1; m.dep_of_q[p] = -1; }
    int ndep = 0;
    std::vector<int> joint_of_child(nl, -1);
    for (int j = 0; j < s->n_joints; j++) {
        const int p = s->joint_parent[j], c = s->joint_child[j], k = s->joint_kind[j];
        if (p < 0 || p >= nl || c <= p || c >= nl || joint_of_child[c] >= 0) return fail(CPE_BAD_ARG, "bad joint");
        m.joint_parent[j] = p; m.joint_child[j] = c; m.joint_kind[j] = k;
        joint_of_child[c] = j;
        m.joint_dep0[j] = ndep;
        m.dep_joint[ndep] = j;
        m.dep_of_q[3 + 3 * c] = ndep++;
        if (k == CPE_JOINT_REVOLUTE_Y) { m.dep_joint[ndep] = j; m.dep_of_q[3 + 3 * c + 2] = ndep++; }
        else if (k != CPE_JOINT_HOOKE_YZ) return fail(CPE_BAD_ARG, "unknown joint kind");
        if (ndep > CPE_MAX_DEP) return fail(CPE_BAD_ARG, "too many dependent angles");
    }
    m.ndep = ndep;
    int nu = 0;
    for (int p = 0; p < nq; p++)
        if (m.dep_of_q[p] < 0) { if (nu >= CPE_NX) return fail(CPE_BAD_ARG, "more than 28 independent dofs"); m.indep[nu] = p; m.u_of_q[p] = nu++; }
    m.nu = nu;
    if (nu != CPE_NX) return fail(CPE_BAD_ARG, "the solver is built for exactly 28 independent dofs");
    for (int p = 0; p < nq; p++)
        if (m.dep_of_q[p] >= 0 && s->motion_w[p] != 0.0) return fail(CPE_BAD_ARG, "motion weight on a dependent angle");
    for (int k = 0; k < nu; k++) {
        const int p = m.indep[k];
        m.motion_w_u[k] = s->motion_w[p];
        m.rel_sign_u[k] = s->rel_ref[p] < 0 ? 1.0 : s->rel_sign[p];
        m.rel_ref_u[k] = s->rel_ref[p] < 0 ? -1 : m.u_of_q[s->rel_ref[p]];
        if (s->rel_ref[p] >= 0 && m.rel_ref_u[k] < 0) return fail(CPE_BAD_ARG, "relative angle references a dependent dof");
    }
    // joint bodies and S column layout
    m.n_srow = 0;
    for (int r = 0; r < CPE_MAX_DEP; r++) m.srow[r] = -1;
    for (int j = 0; j < s->n_joints; j++) {
        const int p = m.joint_parent[j], c = m.joint_child[j], r = m.joint_dep0[j];
        if (m.joint_kind[j] == CPE_JOINT_REVOLUTE_Y) {
            int body = p;
            const int jp = joint_of_child[p];
            if (jp >= 0) {
                if (m.joint_kind[jp] != CPE_JOINT_REVOLUTE_Y) return fail(CPE_BAD_ARG, "revolute joint below a hooke joint is not supported");
                body = m.joint_body[jp];
            }
            m.joint_body[j] = body;
            for (int a = 0; a < 3; a++)
                if (m.u_of_q[3 + 3 * body + a] < 0) return fail(CPE_BAD_ARG, "revolute chain must hang off a link with independent angles");
            if (m.u_of_q[3 + 3 * c + 1] < 0) return fail(CPE_BAD_ARG, "theta of a revolute child must be independent");
            for (int rr = r; rr < r + 2; rr++) {
                m.scol_n[rr] = 4; m.dep_level[rr] = 0;
                for (int a = 0; a < 3; a++) m.scol[rr][a] = m.u_of_q[3 + 3 * body + a];
                m.scol[rr][3] = m.u_of_q[3 + 3 * c + 1];
            }
        } else {
            m.joint_body[j] = p;
            m.srow[r] = m.n_srow++;
            int n = 0;
            auto add = [&](int col, int kind, int ang, int chain) -> bool {
                for (int e = 0; e < n; e++)
                    if (m.scol[r][e] == col) { if (chain >= 0) m.hk_chain[r][e] = (int16_t)chain; else { m.hk_kind[r][e] = (int8_t)kind; m.hk_ang[r][e] = (int8_t)ang; } return true; }
                if (n >= CPE_MAX_SCOL) return false;
                m.scol[r][n] = col; m.hk_kind[r][n] = (int8_t)kind; m.hk_ang[r][n] = (int8_t)ang; m.hk_chain[r][n] = (int16_t)chain; n++;
                return true;
            };
            if (m.u_of_q[3 + 3 * c + 1] < 0 || m.u_of_q[3 + 3 * c + 2] < 0) return fail(CPE_BAD_ARG, "theta/psi of a hooke child must be independent");
            bool ok = add(m.u_of_q[3 + 3 * c + 1], 0, 1, -1) && add(m.u_of_q[3 + 3 * c + 2], 0, 2, -1);
            m.dep_level[r] = 0;
            for (int a = 0; a < 3 && ok; a++) {
                const int pq = 3 + 3 * p + a;
                if (m.u_of_q[pq] >= 0) ok = add(m.u_of_q[pq], 1, a, -1);
                else {
                    if (a != 0) return fail(CPE_BAD_ARG, "hooke parent with dependent theta/psi is not supported");
                    const int pr = m.dep_of_q[pq];
                    if (m.dep_level[pr] != 0 || m.joint_kind[m.dep_joint[pr]] != CPE_JOINT_HOOKE_YZ)
                        return fail(CPE_BAD_ARG, "hooke chains deeper than two are not supported");
                    m.dep_level[r] = 1;
                    for (int e = 0; e < m.scol_n[pr] && ok; e++) ok = add(m.scol[pr][e], 2, 0, m.srow[pr] * CPE_MAX_SCOL + e);
                }
            }
            if (!ok) return fail(CPE_BAD_ARG, "too many columns in a hooke row");
            m.scol_n[r] = n;
        }
    }
    // revolute (leg) links: R_c = R_body Ry(alpha_c)
    std::vector<int> rev_of_link(nl, -1);
    m.nrev = 0;
    for (int p = 0; p < nq; p++) m.euler_rev[p] = -1;
    for (int k = 0; k < nu; k++) { m.rev_of_u[k] = -1; m.ucoord_src[k] = m.indep[k]; m.bodyang_j[k] = -1; m.bodyang_legs_n[k] = 0; }
    for (int j = 0; j < s->n_joints; j++)
        if (m.joint_kind[j] == CPE_JOINT_REVOLUTE_Y) {
            const int r = m.nrev++, c = m.joint_child[j], Bk = m.joint_body[j];
            rev_of_link[c] = r;
            m.rev_child[r] = c; m.rev_body[r] = Bk; m.rev_u[r] = m.u_of_q[3 + 3 * c + 1];
            m.rev_of_u[m.rev_u[r]] = r; m.euler_rev[3 + 3 * c + 1] = r;
            for (int a = 0; a < 3; a++) {
                const int kb = m.u_of_q[3 + 3 * Bk + a];
                m.rev_body_u[r][a] = kb; m.bodyang_j[kb] = a;
                m.bodyang_legs[kb][m.bodyang_legs_n[kb]++] = r;
            }
        }
    m.ns = nq + m.nrev;
    for (int k = 0; k < nu; k++) {
        const int r = m.rev_of_u[k];
        m.ucost[k] = r < 0 ? m.indep[k] : ((3 + 3 * m.rev_body[r] + 1) | (nq + r + 1) << 8);      // trunk: its Euler angle; leg: theta_B + alpha_r
    }
    if (m.nrev > LM_MAX_REV) return fail(CPE_BAD_ARG, "more than 12 leg links");
    for (int k = 0; k < nu; k++) if (m.bodyang_legs_n[k] > LM_MAX_LEGS) return fail(CPE_BAD_ARG, "more than 6 leg links on one body");
    for (int r = 0; r < m.nrev; r++) m.ucoord_src[m.rev_u[r]] = nq + r;
    m.n_trunk = 0;
    for (int i = 0; i < nl; i++) { m.trunk_slot[i] = -1; if (rev_of_link[i] < 0) { m.trunk_slot[i] = m.n_trunk; m.trunk_link[m.n_trunk++] = i; } }
    // marker chains and Jacobian slots
    int S = 0, mct = 0, ss = 0, sv = 0;
    for (int l = 0; l < L; l++) {
        int chain[CPE_MAX_LINKS], n = 0, k = s->marker_link[l];
        if (k < 0 || k >= nl) return fail(CPE_BAD_ARG, "bad marker link");
        while (k >= 0) { chain[n++] = k; k = s->parent[k]; }
        if (n > CPE_MAX_CHAIN) return fail(CPE_BAD_ARG, "marker chain too long");
        m.chain_len[l] = n;
        for (int i = 0; i < n; i++) {
            const int link = chain[n - 1 - i];
            m.chain_link[l][i] = link;
            const double* v = i == n - 1 ? s->marker_off[l] : s->attach[chain[n - 2 - i]];
            for (int d = 0; d < 3; d++) m.chain_vec[l][i][d] = v[d];
        }
        m.slot_off[l] = S;
        if (S + 3 + 3 * n > CPE_MAX_SLOTS) return fail(CPE_BAD_ARG, "too many Jacobian slots");
        for (int d = 0; d < 3; d++) { m.slot_marker[S] = l; m.slot_dof[S] = d; m.slot_cpos[S] = -1; m.slot_ang[S] = 0; S++; }
        for (int i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) { m.slot_marker[S] = l; m.slot_dof[S] = 3 + 3 * m.chain_link[l][i] + a; m.slot_cpos[S] = i; m.slot_ang[S] = a; S++; }
        // ---- solver side: reduced columns of this marker.  Leg links are rotations of their body about its
        // y axis (R_c = R_B Ry(alpha_c)); everything the legs contribute is a body matrix times a body-frame
        // vector that depends on the alphas ("dynamic vectors").
        int nc = 0;
        auto col_index = [&](int col) -> int {
            for (int e = 0; e < nc; e++) if (m.mcol[l][e] == col) return e;
            if (nc >= CPE_MAX_MCOL) return -1;
            m.mcol[l][nc] = col; m.term_n[l][nc] = 0; return nc++;
        };
        auto add_term = [&](int col, int slot, int sidx) -> bool {
            const int e = col_index(col);
            if (e < 0 || m.term_n[l][e] >= CPE_MAX_TERMS) return false;
            m.term_slot[l][e][m.term_n[l][e]] = (int16_t)slot; m.term_s[l][e][m.term_n[l][e]] = (int16_t)sidx; m.term_n[l][e]++;
            return true;
        };
        auto new_slot = [&](int moff, int vdyn, const double* v) -> int {
            if (ss >= CPE_MAX_SLOTS) return -1;
            m.ss_moff[ss] = moff; m.ss_vdyn[ss] = vdyn;
            for (int d = 0; d < 3; d++) m.ss_vec[ss][d] = v ? v[d] : 0.0;
            return ss++;
        };
        bool ok = true;
        const double ex[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        for (int d = 0; d < 3 && ok; d++) { const int sl = new_slot(-1, -1, ex[d]); ok = sl >= 0 && add_term(m.u_of_q[d], sl, -1); }
        int n_trunk_chain = 0;
        while (n_trunk_chain < n && rev_of_link[m.chain_link[l][n_trunk_chain]] < 0) n_trunk_chain++;
        m.pc_len[l] = n_trunk_chain; m.pw_id[l] = -1; m.pw_body[l] = 0;
        for (int i = 0; i < n_trunk_chain && ok; i++) {
            const int link = m.chain_link[l][i];
            m.pc_link[l][i] = link;
            for (int d = 0; d < 3; d++) m.pc_vec[l][i][d] = m.chain_vec[l][i][d];
            for (int a = 0; a < 3 && ok; a++) {
                const int sl = new_slot(9 * (4 * m.trunk_slot[link] + 1 + a), -1, m.chain_vec[l][i]);
                const int p = 3 + 3 * link + a;
                if (sl < 0) { ok = false; break; }
                if (m.u_of_q[p] >= 0) ok = add_term(m.u_of_q[p], sl, -1);
                else { const int r = m.dep_of_q[p]; for (int e = 0; e < m.scol_n[r] && ok; e++) ok = m.srow[r] >= 0 && add_term(m.scol[r][e], sl, m.srow[r] * CPE_MAX_SCOL + e); }
            }
        }
        if (n_trunk_chain < n && ok) {
            const int Bk = m.chain_link[l][n_trunk_chain - 1];
            const int nleg = n - n_trunk_chain;
            if (nleg > 3 || sv + 1 + nleg > CPE_MAX_SDYN) return fail(CPE_BAD_ARG, "leg chain too long");
            const int wid = sv++;
            m.sv_kind[wid] = 0; m.sv_cnt[wid] = nleg;
            for (int i = 0; i < nleg; i++) {
                const int link = m.chain_link[l][n_trunk_chain + i];
                if (rev_of_link[link] < 0 || m.rev_body[rev_of_link[link]] != Bk) return fail(CPE_BAD_ARG, "leg links must follow their body in the marker chain");
                m.sv_rev[wid][i] = rev_of_link[link];
                for (int d = 0; d < 3; d++) m.sv_vec[wid][i][d] = m.chain_vec[l][n_trunk_chain + i][d];
            }
            m.pw_id[l] = wid; m.pw_body[l] = Bk;
            for (int a = 0; a < 3 && ok; a++) {           // body angles act on the whole leg vector
                const int sl = new_slot(9 * (4 * m.trunk_slot[Bk] + 1 + a), wid, nullptr);
                ok = sl >= 0 && add_term(m.u_of_q[3 + 3 * Bk + a], sl, -1);
            }
            for (int i = 0; i < nleg && ok; i++) {        // alpha of each leg link
                const int did = sv++;
                m.sv_kind[did] = 1; m.sv_cnt[did] = 1; m.sv_rev[did][0] = m.sv_rev[wid][i];
                for (int d = 0; d < 3; d++) m.sv_vec[did][0][d] = m.sv_vec[wid][i][d];
                const int sl = new_slot(9 * (4 * m.trunk_slot[Bk]), did, nullptr);
                ok = sl >= 0 && add_term(m.rev_u[m.sv_rev[wid][i]], sl, -1);
            }
        }
        if (!ok) return fail(CPE_BAD_ARG, "marker depends on too many reduced dofs");
        m.mcol_n[l] = nc; m.mcol_off[l] = mct;
        for (int e = 0; e < nc; e++) { m.mc_marker[mct] = (int16_t)l; m.mc_j[mct] = (int16_t)e; mct++; }
    }
    m.ss_n = ss; m.sv_n = sv;
    m.slot_off[L] = S;
    // Row alignment: J is [C][S][2] doubles, so a camera row is 16 S bytes.  S = 270 (the reference's 24 markers) makes every
    // other row start 32 B off a 64-byte line and measured 9 % slower than S = 276; pad S to a multiple of 4 with slots
    // of marker 0 and a dof outside its chain (structurally zero: the kernel stores 0 there, as the dense Jacobian has).
    for (int d = s->n_links * 3 + 2; (S & 3) && d >= 3; d--) {
        bool in_chain = false;
        for (int i = 0; i < m.chain_len[0]; i++) in_chain |= (d - 3) / 3 == m.chain_link[0][i];
        if (in_chain) continue;
        if (S >= CPE_MAX_SLOTS) return fail(CPE_BAD_ARG, "too many Jacobian slots");
        m.slot_marker[S] = 0; m.slot_dof[S] = d; m.slot_cpos[S] = -2; m.slot_ang[S] = 0; S++;
    }
    m.S = S; m.mcol_off[L] = mct; m.mc_total = mct;
    if (S > 5 * WAVE) return fail(CPE_BAD_ARG, "more than 320 Jacobian slots");
    {   // flat term list of the reduced marker columns (see cpe_model.h)
        m.tl_n = 0;
        for (int l = 0; l < L; l++)
            for (int e = 0; e < m.mcol_n[l]; e++)
                for (int k = 0; k < m.term_n[l][e]; k++) {
                    if (m.tl_n >= CPE_MAX_TL) return fail(CPE_BAD_ARG, "too many marker-column terms");
                    const int sl = m.term_slot[l][e][k], si = m.term_s[l][e][k], item = m.mcol_off[l] + e;
                    if (item >= 1024 || si + 1 >= 1024 || m.ss_vdyn[sl] + 1 >= 1024) return fail(CPE_BAD_ARG, "marker-column term out of range");
                    auto& T = m.tl[m.tl_n++];
                    T.w0 = item | ((si + 1) << 10) | ((m.ss_vdyn[sl] + 1) << 20);
                    T.moff = m.ss_moff[sl];
                    for (int d = 0; d < 3; d++) T.v[d] = m.ss_vec[sl][d];
                }
    }
    {   // gather lists for H and g (see cpe_model.h)
        const int nu_ = m.nu;
        std::vector<std::vector<uint32_t>> by_entry(nu_ * nu_);
        for (int l = 0; l < L; l++)
            for (int i = 0; i < m.mcol_n[l]; i++)
                for (int j = 0; j <= i; j++) {
                    int a = m.mcol[l][i], b = m.mcol[l][j];
                    if (a < b) std::swap(a, b);
                    by_entry[a * nu_ + b].push_back((uint32_t)(m.mcol_off[l] + i) | ((uint32_t)(m.mcol_off[l] + j) << 8) | ((uint32_t)l << 16));
                }
        std::vector<int> order;
        for (int e = 0; e < nu_ * nu_; e++) if (!by_entry[e].empty()) order.push_back(e);
        for (auto& w : m.h_covered) w = 0;
        for (int e : order) { const int a = e / nu_, b = e % nu_; for (int t : {a * nu_ + b, b * nu_ + a}) m.h_covered[t >> 5] |= 1u << (t & 31); }
        std::sort(order.begin(), order.end(), [&](int x, int y) { return by_entry[x].size() != by_entry[y].size() ? by_entry[x].size() > by_entry[y].size() : x < y; });
        int load[64] = {0};
        for (int e : order) {
            int best = 0;
            for (int ln = 1; ln < 64; ln++) if (load[ln] < load[best]) best = ln;
            if (load[best] + (int)by_entry[e].size() > CPE_MAX_HG) return fail(CPE_BAD_ARG, "normal-matrix gather list too long");
            const uint32_t ab = (uint32_t)(((e / nu_) << 5) | (e % nu_));
            for (size_t k = 0; k < by_entry[e].size(); k++) m.hg_code[load[best] + k][best] = by_entry[e][k] | (k == 0 ? 1u << 21 : 0u) | (ab << 22);
            load[best] += (int)by_entry[e].size();
        }
        m.hg_max = 0;
        for (int ln = 0; ln < 64; ln++) { m.hg_cnt[ln] = load[ln]; if (load[ln] > m.hg_max) m.hg_max = load[ln]; }
        for (int k = 0; k < nu_; k++) m.gg_cnt[k] = 0;
        for (int l = 0; l < L; l++)
            for (int i = 0; i < m.mcol_n[l]; i++) {
                const int a = m.mcol[l][i];
                if (m.gg_cnt[a] >= CPE_MAX_GG) return fail(CPE_BAD_ARG, "gradient gather list too long");
                m.gg_code[m.gg_cnt[a]++][a] = (uint16_t)((m.mcol_off[l] + i) | (l << 8));
            }
    }
    {   // per-lane tables of k_frame_normal with resolved indices (see cpe_model.h)
        m.hk_n = 0;
        for (int r = 0; r < m.ndep; r++) {
            const int j = m.dep_joint[r];
            if (m.joint_kind[j] != CPE_JOINT_HOOKE_YZ || m.srow[r] < 0) continue;
            for (int jc = 0; jc < m.scol_n[r]; jc++) {
                if (m.hk_n >= 64) return fail(CPE_BAD_ARG, "more than 64 entries in the hooke rows of S");
                const int ps = m.trunk_slot[m.joint_parent[j]], cs = m.trunk_slot[m.joint_child[j]];
                if (ps < 0 || cs < 0 || ps >= 64 || cs >= 64) return fail(CPE_BAD_ARG, "hooke joint between leg links");
                m.hk_w[m.hk_n][0] = ps | (cs << 6) | ((m.hk_kind[r][jc] & 3) << 12) | ((m.hk_ang[r][jc] & 3) << 14) | ((m.dep_level[r] & 1) << 16);
                m.hk_w[m.hk_n][1] = (m.hk_chain[r][jc] + 1) | ((m.srow[r] * CPE_MAX_SCOL + jc) << 16);
                m.hk_n++;
            }
        }
        m.hj_n = 0;
        for (int j = 0; j < s->n_joints; j++)
            if (m.joint_kind[j] == CPE_JOINT_HOOKE_YZ) {
                if (m.hj_n >= 64) return fail(CPE_BAD_ARG, "more than 64 hooke joints");
                m.fn_lane[m.hj_n++][0] = m.joint_parent[j] | (m.joint_child[j] << 8) | ((m.dep_of_q[3 + 3 * m.joint_parent[j]] >= 0 ? 1 : 0) << 16);
            }
        for (int t = 0; t < 4 * m.n_trunk && t < 64; t++) m.fn_lane[t][1] = m.trunk_link[t >> 2];     // (nrev <= 16 and sv_n <= 48 by the table sizes)
        for (int r = 0; r < m.nrev; r++) m.fn_lane[r][2] = m.rev_body[r] | (m.rev_child[r] << 8) | (m.trunk_slot[m.rev_body[r]] << 16);
        for (int t = 0; t < m.sv_n; t++) m.fn_lane[t][3] = m.sv_kind[t] | (m.sv_cnt[t] << 2) | (m.sv_rev[t][0] << 4) | (m.sv_rev[t][1] << 10) | (m.sv_rev[t][2] << 16);
        for (int l = 0; l < L; l++) {
            for (int i = 0; i < m.pc_len[l]; i++) m.pc_off[l][i] = 36 * m.trunk_slot[m.pc_link[l][i]];
            m.pw_off[l] = m.pw_id[l] >= 0 ? 36 * m.trunk_slot[m.pw_body[l]] : 0;
        }
    }
    for (int bnd = 0; bnd < s->n_bounds; bnd++) {
        const int a = s->bound_a[bnd], bb = s->bound_b[bnd];
        if (a < 0 || a >= nq || bb >= nq || m.u_of_q[a] < 0 || (bb >= 0 && m.u_of_q[bb] < 0))
            return fail(CPE_BAD_ARG, "bounds must act on independent dofs");
        m.bound_ua[bnd] = m.u_of_q[a]; m.bound_ub[bnd] = bb < 0 ? -1 : m.u_of_q[bb];
        m.bound_lo[bnd] = s->bound_lo[bnd]; m.bound_up[bnd] = s->bound_up[bnd];
        // (state index of the value, + 1 of the second addend or 0) for each side: a leg pitch is theta_B + alpha (ucost)
        const int ca = m.ucost[m.u_of_q[a]], cb = bb < 0 ? 0 : m.ucost[m.u_of_q[bb]];
        m.bnd_q[bnd] = (ca & 255) | (bb < 0 ? 0 : ((cb & 255) + 1) << 8) | ((ca >> 8) & 255) << 16 | (bb < 0 ? 0 : ((cb >> 8) & 255)) << 24;
    }
    return CPE_OK;
}

// the plain variant of k_frame_normal (no Gaussian-mixture prior, no shutter delay) writes H straight to HBM and needs 6 KB less LDS per wave
#define FRAME_NORMAL(plain) ((plain) ? k_frame_normal<true> : k_frame_normal<false>)
static size_t lds_fk(const DevModel& m) { return sizeof(double) * (m.nq + 6 * m.nl + 36 * m.nl + 3 * m.L + 23 * m.C); }
static size_t lds_normal(const DevModel& m, int gmm_k = 0, int gmm_dim = 0, bool shutter = false) {
    // g (and, unless the plain variant writes H straight to HBM, H | g) overlay the S rows, which are dead once Dp is built
    const bool plain = gmm_k == 0 && !shutter;
    size_t ov = CPE_MAX_SCOL * m.n_srow, hg = plain ? m.nu : m.nu * m.nu + m.nu;
    size_t camov = 6 * m.nl + 2 * m.nrev + 36 * m.n_trunk;   // sin / cos tables + trunk rotations, later the staged cameras (sizeof(cpe_camera) = 22 doubles each)
    if (camov < (size_t)22 * m.C) camov = (size_t)22 * m.C;
    size_t n = m.ns + camov + 3 * m.L + 3 * m.sv_n + GAM_STRIDE * m.nrev + (ov > hg ? ov : hg) +
               9 * m.L + 3 * m.mc_total;
    if (gmm_k > 0) n += CPE_NX + gmm_k * gmm_dim + CPE_MAX_GMM + CPE_NX;          // x | P_k (x - mu_k) | log p_k | gradient in x
    if (shutter) n += 15 * m.C + 6 * m.L + 6;          // shift | coefficients | rc | Mc per camera, M1 per marker, M2
    return sizeof(double) * n;
}

extern "C" {

const char* cpe_last_error(void) { return g_err.c_str(); }

void cpe_default_options(cpe_options* o) {
    o->h = 1.0 / 120.0; o->loss_a = 3.0; o->loss_b = 10.0; o->loss_c = 20.0; o->cost_scale = 1e-3;
    o->bound_penalty = 1e4; o->bound_tol = 1e-6; o->lambda0 = 1e-4; o->tol_step = 1e-8; o->tol_cost = 1e-9; o->max_iter = 200;
    o->curvature = 0; o->max_outer = 8; o->_pad = 0;
}

void cpe_destroy(cpe_handle* h);
static cpe_status create_impl(cpe_handle* h, const cpe_skeleton* skel, const cpe_camera* cams, int32_t n_cams, const cpe_options* opts,
                              const cpe_priors* priors, bool use_pri) {
    const int device = h->device;
    cpe_status s = build_model(skel, cams, n_cams, opts, h->hm);
    if (s != CPE_OK) return s;
    HIPCHK(hipSetDevice(device));
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        // dynamic LDS above 64 KiB needs an explicit opt-in per kernel
        // dynamic LDS above 64 KiB needs an explicit opt-in per kernel
        const void* ks[] = {(const void*)&k_resjac<true, 4, 3, 2>, (const void*)&k_resjac<false, 4, 3, 2>,
                            (const void*)&k_resjac<true, 4, 4, 2>, (const void*)&k_resjac<false, 4, 4, 2>};
        for (const void* k : ks) HIPCHK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&h->poll_ev[i], hipEventDisableTiming));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->poll_host), 2 * sizeof(int), hipHostMallocDefault));
    HIPCHK(hipMalloc(&h->n_act, sizeof(int)));
    HIPCHK(hipMalloc(&h->dm, sizeof(DevModel)));
    HIPCHK(hipMemcpy(h->dm, &h->hm, sizeof(DevModel), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&h->flag, sizeof(int)));
    if (use_pri) {
        if (h->hm.nu != CPE_NX) return fail(CPE_BAD_ARG, "learned priors need the 28 relative angles of the reference's skeleton");
        if (priors->gmm_k > 0 && priors->gmm_dim > h->hm.nu) return fail(CPE_BAD_ARG, "pose prior dimension exceeds the number of relative angles");
        std::vector<DevPriors> hp(1);
        DevPriors& P = hp[0];
        memset(&P, 0, sizeof(P));
        P.p = *priors;
        const int W = priors->lr_window, nu = CPE_NX;
        if (W > 0) {
            // K_t[p][j]: lag blocks of slack = sum_t K_t x_{n-W+t} - b
            auto K = [&](int t, int p2, int j) -> double { return t == W ? (p2 == j ? 1.0 : 0.0) : -priors->lr_coef[p2][t * nu + j]; };
            for (int ta = 0; ta <= W; ta++)
                for (int tb = 0; tb <= ta; tb++)
                    for (int i = 0; i < nu; i++)
                        for (int j = 0; j < nu; j++) {
                            double a = 0.0;
                            for (int p2 = 0; p2 < nu; p2++) a += K(ta, p2, i) * priors->lr_w[p2] * K(tb, p2, j);
                            P.lr_PK[ta][tb][i * nu + j] = 2.0 * a;
                        }
            for (int k = 0; k <= W; k++)
                for (int ta = k; ta <= W; ta++)
                    for (int e = 0; e < nu * nu; e++) P.lr_HI[k][e] += P.lr_PK[ta][ta - k][e];
            for (int f = 0; f < W * nu; f++) {
                bool any = false;
                for (int p2 = 0; p2 < nu; p2++) { P.lr_coefT[f][p2] = priors->lr_coef[p2][f]; any = any || priors->lr_coef[p2][f] != 0.0; }
                if (any) P.lr_feat[P.lr_nf++] = (uint8_t)f;
            }
        }
        // X' = dx/du: x_k = sign_k (c_k - c_ref), c = the reduced coordinate itself or, for a leg link, theta_B + alpha_c (cost pitch)
        const DevModel& hm = h->hm;
        for (int i = 0; i < nu; i++)
            for (int side = 0; side < 2; side++) {
                const int kk = side == 0 ? i : hm.rel_ref_u[i];
                if (kk < 0) continue;
                const double sgn = side == 0 ? hm.rel_sign_u[i] : -hm.rel_sign_u[i];
                P.Xc[i][kk] += sgn;
                const int r = hm.rev_of_u[kk];
                if (r >= 0) P.Xc[i][hm.rev_body_u[r][1]] += sgn;
            }
        // out = X'^T A X' (A, out row-major nu x nu), fixed summation order
        auto to_u = [&](const double* A, double* out) {
            std::vector<double> T((size_t)nu * nu);
            for (int r2 = 0; r2 < nu; r2++)
                for (int j = 0; j < nu; j++) { double a = 0.0; for (int c = 0; c < nu; c++) a += A[r2 * nu + c] * P.Xc[c][j]; T[(size_t)r2 * nu + j] = a; }
            for (int i = 0; i < nu; i++)
                for (int j = 0; j < nu; j++) { double a = 0.0; for (int r2 = 0; r2 < nu; r2++) a += P.Xc[r2][i] * T[(size_t)r2 * nu + j]; out[i * nu + j] = a; }
        };
        if (W > 0) {
            for (int ta = 0; ta <= W; ta++) for (int tb = 0; tb <= ta; tb++) to_u(P.lr_PK[ta][tb], P.lr_PKu[ta][tb]);
            for (int k = 0; k <= W; k++) to_u(P.lr_HI[k], P.lr_HIu[k]);
        }
        if (priors->gmm_k > 0) {
            const int D = priors->gmm_dim, off = nu - D;
            std::vector<double> A((size_t)nu * nu);
            for (int k = 0; k < priors->gmm_k; k++) {
                std::fill(A.begin(), A.end(), 0.0);
                for (int i = 0; i < D; i++)
                    for (int j = 0; j < D; j++) { A[(size_t)(off + i) * nu + off + j] = priors->gmm_P[k][i][j]; P.gmm_PT[k][j][i] = priors->gmm_P[k][i][j]; }
                to_u(A.data(), P.gmm_Q[k]);
            }
        }
        HIPCHK(hipMalloc(&h->pri, sizeof(DevPriors)));
        HIPCHK(hipMemcpy(h->pri, &P, sizeof(DevPriors), hipMemcpyHostToDevice));
        h->gmm_k = priors->gmm_k; h->gmm_dim = priors->gmm_dim; h->lr_window = W;
        h->pb = W > 3 ? W : 3;
    }
    return CPE_OK;
}

cpe_status cpe_create(const cpe_skeleton* skel, const cpe_camera* cams, int32_t n_cams, const cpe_options* opts,
                      const cpe_priors* priors, int32_t device, cpe_handle** out) {
    if (!skel || !cams || !opts || !out) return fail(CPE_BAD_ARG, "null argument");
    const bool use_pri = priors && (priors->gmm_k > 0 || priors->lr_window > 0);
    if (use_pri) {
        if (priors->gmm_k < 0 || priors->gmm_k > CPE_MAX_GMM || priors->gmm_dim < 0 || priors->gmm_dim > CPE_NX || (priors->gmm_k > 0 && priors->gmm_dim < 1))
            return fail(CPE_BAD_ARG, "pose prior: component count / dimension out of range");
        if (priors->lr_window < 0 || priors->lr_window > CPE_MAX_WINDOW) return fail(CPE_BAD_ARG, "motion prior: window out of range");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(CPE_NO_DEVICE, "no HIP device: this library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(CPE_BAD_ARG, "device index out of range");
    cpe_handle* h = new cpe_handle();
    h->device = device; h->opts = *opts;
    const cpe_status s = create_impl(h, skel, cams, n_cams, opts, priors, use_pri);
    if (s != CPE_OK) { cpe_destroy(h); return s; }      // releases whatever was allocated before the failure
    *out = h;
    return CPE_OK;
}

static void free_kws(cpe_handle* h) {
    void* ptrs[] = {h->kmut, h->kmus, h->fbuf, h->kmu, h->Jbuf, h->Abuf, h->pieces, h->gTb, h->dstat, h->slackb, h->Tbuf, h->gk, h->Bk, h->Hk, h->pmeta};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    h->kmut = h->kmus = h->fbuf = h->kmu = h->Jbuf = h->Abuf = h->pieces = h->gTb = h->dstat = h->slackb = h->Tbuf = h->gk = h->Bk = h->Hk = nullptr; h->pmeta = nullptr;
    h->kws_frames = 0;
}

static void free_ws(cpe_handle* h) {
    free_kws(h);
    void* ptrs[] = {h->qbuf, h->gbuf, h->Bbuf, h->costbuf, h->Lbuf, h->zbuf, h->gtbuf, h->dgbuf, h->cmax, h->mu, h->gambuf, h->st, h->Hlr, h->act};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    h->qbuf = h->gbuf = h->Bbuf = h->costbuf = h->Lbuf = h->zbuf = h->gtbuf = h->dgbuf = h->cmax = h->mu = h->gambuf = h->Hlr = nullptr; h->st = nullptr; h->act = nullptr;
    h->ws_frames = 0; h->ws_B = 0;
}

void cpe_destroy(cpe_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    free_ws(h);
    if (h->dm) (void)hipFree(h->dm);
    if (h->flag) (void)hipFree(h->flag);
    if (h->pri) (void)hipFree(h->pri);
    if (h->eom) (void)hipFree(h->eom);
    if (h->dyn) (void)hipFree(h->dyn);
    if (h->dk) (void)hipFree(h->dk);
    if (h->n_act) (void)hipFree(h->n_act);
    if (h->poll_host) (void)hipHostFree(h->poll_host);
    for (int i = 0; i < 2; i++) if (h->poll_ev[i]) (void)hipEventDestroy(h->poll_ev[i]);
    if (h->order_ev) (void)hipEventDestroy(h->order_ev);
    for (auto& sp : h->prof_spans) { (void)hipEventDestroy(sp.second.first); (void)hipEventDestroy(sp.second.second); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

void* cpe_stream(cpe_handle* h) { return h ? (void*)h->stream : nullptr; }
cpe_status cpe_synchronize(cpe_handle* h) {
    if (!h) return fail(CPE_BAD_ARG, "null handle");
    HIPCHK(hipStreamSynchronize(h->stream));
    return CPE_OK;
}

// ---- ordering against the caller's stream (e.g. torch's current stream) ---------------------------------------------
cpe_status cpe_stream_wait(cpe_handle* h, void* other) {
    if (!h) return fail(CPE_BAD_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->order_ev, (hipStream_t)other));
    HIPCHK(hipStreamWaitEvent(h->stream, h->order_ev, 0));
    return CPE_OK;
}
cpe_status cpe_stream_signal(cpe_handle* h, void* other) {
    if (!h) return fail(CPE_BAD_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->order_ev, h->stream));
    HIPCHK(hipStreamWaitEvent((hipStream_t)other, h->order_ev, 0));
    return CPE_OK;
}

// ---- per-kernel device time of cpe_solve ------------------------------------------------------------------------------
cpe_status cpe_profile_enable(cpe_handle* h, int32_t on) {
    if (!h) return fail(CPE_BAD_ARG, "null handle");
    h->prof = on != 0;
    for (int i = 0; i < CPE_PROFILE_SLOTS; i++) { h->prof_ms[i] = 0.0; h->prof_n[i] = 0; }
    return CPE_OK;
}
cpe_status cpe_profile_get(cpe_handle* h, double* ms, int64_t* launches) {
    if (!h || !ms || !launches) return fail(CPE_BAD_ARG, "null argument");
    for (int i = 0; i < CPE_PROFILE_SLOTS; i++) { ms[i] = h->prof_ms[i]; launches[i] = h->prof_n[i]; }
    return CPE_OK;
}
static void prof_begin(cpe_handle* h, int id) {
    if (!h->prof) return;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    (void)hipEventRecord(a, h->stream);
    h->prof_spans.push_back({id, {a, b}});
}
static void prof_end(cpe_handle* h) {
    if (!h->prof || h->prof_spans.empty()) return;
    (void)hipEventRecord(h->prof_spans.back().second.second, h->stream);
}
static void prof_collect(cpe_handle* h) {       // after the stream has been synchronised
    for (auto& sp : h->prof_spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.second.first, sp.second.second) == hipSuccess) { h->prof_ms[sp.first] += ms; h->prof_n[sp.first]++; }
        (void)hipEventDestroy(sp.second.first); (void)hipEventDestroy(sp.second.second);
    }
    h->prof_spans.clear();
}

int32_t cpe_jacobian_slots(const cpe_handle* h) { return h ? h->hm.S : 0; }
cpe_status cpe_jacobian_layout(const cpe_handle* h, int32_t* slot_marker, int32_t* slot_dof) {
    if (!h || !slot_marker || !slot_dof) return fail(CPE_BAD_ARG, "null argument");
    for (int s = 0; s < h->hm.S; s++) { slot_marker[s] = h->hm.slot_marker[s]; slot_dof[s] = h->hm.slot_dof[s]; }
    return CPE_OK;
}
int32_t cpe_num_independent(const cpe_handle* h) { return h ? h->hm.nu : 0; }
cpe_status cpe_independent_dofs(const cpe_handle* h, int32_t* dofs) {
    if (!h || !dofs) return fail(CPE_BAD_ARG, "null argument");
    for (int k = 0; k < h->hm.nu; k++) dofs[k] = h->hm.indep[k];
    return CPE_OK;
}

cpe_status cpe_eval_resjac(cpe_handle* h, int32_t B, int32_t N, const double* q, const double* meas, const double* weight,
                           double* r, double* J, double* eps, double* cost) {
    if (!h || !q || !meas || !r || !J || !eps) return fail(CPE_BAD_ARG, "null argument");
    if (cost && !weight) return fail(CPE_BAD_ARG, "cost requested without weights");
    if (B < 0 || N < 0) return fail(CPE_BAD_ARG, "negative size");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    if (F > 0x7fffffffULL) return fail(CPE_BAD_ARG, "too many frames for one launch");
    HIPCHK(hipSetDevice(h->device));
    const DevModel& m = h->hm;
    if (m.C * m.L > RJ_MAXPASS * WAVE) return fail(CPE_BAD_ARG, "cpe_eval_resjac supports at most 256 (camera, marker) pairs");
    constexpr int NW = 4;                                   // waves (= frames in flight) per workgroup
    const size_t lds = sizeof(double) * (((rj_shared_doubles(m.C, m.L, m.S) + 1) & ~1) + (size_t)NW * rj_wave_doubles(m.C, m.L, m.nq, m.nl));
    int wg_per_cu = (int)((160 * 1024) / lds);
    if (wg_per_cu < 1) return fail(CPE_BAD_ARG, "model too large for the LDS of one workgroup");
    // 2 workgroups (8 waves) per CU: the kernel needs ~200 VGPRs; capping it at 168 for 3 waves/SIMD spills
    // and measures 25 % slower (profiles/r01_resjac_ablation.md)
    constexpr int variant = 2;
    if (wg_per_cu > variant) wg_per_cu = variant;
    long grid = (long)h->n_cu * wg_per_cu;
    const long need = (long)((F + NW - 1) / NW);
    if (grid > need) grid = need;
    const int npass = (m.C * m.L + WAVE - 1) / WAVE;
#define RJ_LAUNCH(COST, NP, OC) hipLaunchKernelGGL((k_resjac<COST, NW, NP, OC>), dim3((unsigned)grid), dim3(WAVE * NW), lds, h->stream, h->dm, N, (long)F, q, meas, weight, r, J, eps, cost)
    if (cost) { if (npass <= 3) RJ_LAUNCH(true, 3, 2); else RJ_LAUNCH(true, 4, 2); }
    else { if (npass <= 3) RJ_LAUNCH(false, 3, 2); else RJ_LAUNCH(false, 4, 2); }
#undef RJ_LAUNCH
    HIPCHK(hipGetLastError());
    return CPE_OK;
}

cpe_status cpe_project_joints(cpe_handle* h, int32_t B, int32_t N, double* q) {
    if (!h || !q) return fail(CPE_BAD_ARG, "null argument");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemsetAsync(h->flag, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_project, dim3((unsigned)F), dim3(WAVE), sizeof(double) * (h->hm.nq + 6 * h->hm.nl), h->stream, h->dm, q, h->flag);
    HIPCHK(hipGetLastError());
    int fl = 0;
    HIPCHK(hipMemcpyAsync(&fl, h->flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return fl ? CPE_NUMERICAL : CPE_OK;
}

cpe_status cpe_grf_fit(cpe_handle* h, const cpe_grf_options* opt, int32_t B, int32_t N, const double* q, const double* dq,
                       const double* ddq, const int32_t* contact, double* grfz, double* grfxy, double* residual) {
    if (!h || !opt || !q || !dq || !ddq || !contact || !grfz || !grfxy) return fail(CPE_BAD_ARG, "null argument");
    if (B < 0 || N < 0) return fail(CPE_BAD_ARG, "negative size");
    if (opt->n_feet < 1 || opt->n_feet > 4 || opt->iterations < 1 || !(opt->gravity > 0) || !(opt->force_max > 0) || !(opt->friction_ratio >= 0))
        return fail(CPE_BAD_ARG, "grf options out of range");
    for (int f = 0; f < opt->n_feet; f++)
        if (opt->foot_marker[f] < 0 || opt->foot_marker[f] >= h->hm.L) return fail(CPE_BAD_ARG, "foot marker index out of range");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_grf, dim3((unsigned)((F + GRF_PACK - 1) / GRF_PACK)), dim3(WAVE), 0, h->stream, h->dm, *opt, F, q, dq, ddq, contact, grfz, grfxy, residual);
    HIPCHK(hipGetLastError());
    return CPE_OK;
}

cpe_status cpe_eom_rows(cpe_handle* h, const cpe_eom_options* opt, int32_t B, int32_t N, const double* q, const double* dq,
                        const double* ddq, double* rows) {
    if (!h || !opt || !q || !dq || !ddq || !rows) return fail(CPE_BAD_ARG, "null argument");
    if (B < 0 || N < 0) return fail(CPE_BAD_ARG, "negative size");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    HIPCHK(hipSetDevice(h->device));
    if (!h->eom) HIPCHK(hipMalloc(&h->eom, sizeof(cpe_eom_options)));
    HIPCHK(hipMemcpyAsync(h->eom, opt, sizeof(cpe_eom_options), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_eom, dim3((unsigned)F), dim3(WAVE), 0, h->stream, h->dm, h->eom, F, q, dq, ddq, rows);
    HIPCHK(hipGetLastError());
    return CPE_OK;
}

cpe_status cpe_eom_residual(cpe_handle* h, const cpe_dyn_options* opt, int32_t B, int32_t N, const double* q, const double* dq,
                            const double* ddq, const double* tau, const double* lambda, const double* grf, double* residual) {
    if (!h || !opt || !q || !dq || !ddq || !residual) return fail(CPE_BAD_ARG, "null argument");
    if (opt->n_feet < 0 || opt->n_feet > 4 || opt->n_motors < 0 || opt->n_motors > 32) return fail(CPE_BAD_ARG, "dynamics options out of range");
    for (int f = 0; f < opt->n_feet; f++) if (opt->foot_marker[f] < 0 || opt->foot_marker[f] >= h->hm.L) return fail(CPE_BAD_ARG, "foot marker index out of range");
    for (int m = 0; m < opt->n_motors; m++)
        if (opt->motor_first[m] < 0 || opt->motor_first[m] >= h->hm.nl || opt->motor_second[m] < 0 || opt->motor_second[m] >= h->hm.nl ||
            opt->motor_axis[m] < 0 || opt->motor_axis[m] > 2) return fail(CPE_BAD_ARG, "motor definition out of range");
    cpe_status s = cpe_eom_rows(h, &opt->eom, B, N, q, dq, ddq, residual);
    if (s != CPE_OK) return s;
    const size_t F = (size_t)B * N;
    if (F == 0 || (!tau && !lambda && !grf)) return CPE_OK;
    if (!h->dyn) HIPCHK(hipMalloc(&h->dyn, sizeof(cpe_dyn_options)));
    HIPCHK(hipMemcpyAsync(h->dyn, opt, sizeof(cpe_dyn_options), hipMemcpyHostToDevice, h->stream));
    int n_con = 0;
    for (int j = 0; j < h->hm.nj; j++) n_con += h->hm.joint_kind[j] == CPE_JOINT_REVOLUTE_Y ? 2 : 1;
    hipLaunchKernelGGL(k_dyn_forces, dim3((unsigned)F), dim3(WAVE), 0, h->stream, h->dm, h->dyn, F, q, tau, lambda, grf, n_con, residual);
    HIPCHK(hipGetLastError());
    return CPE_OK;
}

cpe_status cpe_forward_kinematics(cpe_handle* h, int32_t B, int32_t N, const double* q, double* positions, double* com) {
    if (!h || !q || !positions) return fail(CPE_BAD_ARG, "null argument");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_fk, dim3((unsigned)F), dim3(WAVE), lds_fk(h->hm), h->stream, h->dm, q, positions, com);
    HIPCHK(hipGetLastError());
    return CPE_OK;
}

cpe_status cpe_marker_velocities(cpe_handle* h, int32_t B, int32_t N, const double* q, const double* dq, double* velocities) {
    if (!h) return fail(CPE_BAD_ARG, "null argument");
    if (B < 0 || N < 0) return fail(CPE_BAD_ARG, "negative size");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    if (!q || !dq || !velocities) return fail(CPE_BAD_ARG, "null argument");
    if (F > 0x7fffffffULL) return fail(CPE_BAD_ARG, "too many frames for one launch");
    HIPCHK(hipSetDevice(h->device));
    const size_t lds = sizeof(double) * (2 * h->hm.nq + 6 * h->hm.nl + 36 * h->hm.nl);
    hipLaunchKernelGGL(k_marker_vel, dim3((unsigned)F), dim3(WAVE), lds, h->stream, h->dm, q, dq, velocities);
    HIPCHK(hipGetLastError());
    return CPE_OK;
}

cpe_status cpe_reproject(cpe_handle* h, int32_t B, int32_t N, const double* positions, double* uv) {
    if (!h) return fail(CPE_BAD_ARG, "null argument");
    if (B < 0 || N < 0) return fail(CPE_BAD_ARG, "negative size");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    if (!positions || !uv) return fail(CPE_BAD_ARG, "null argument");
    if (F > 0x7fffffffULL) return fail(CPE_BAD_ARG, "too many frames for one launch");
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_reproject, dim3((unsigned)F), dim3(WAVE), 0, h->stream, h->dm, positions, uv);
    HIPCHK(hipGetLastError());
    return CPE_OK;
}

cpe_status cpe_triangulate(cpe_handle* h, int32_t n, const int32_t* cam_a, const int32_t* cam_b, const double* uv_a, const double* uv_b,
                           double depth, double* xyz) {
    if (!h) return fail(CPE_BAD_ARG, "null argument");
    if (n < 0) return fail(CPE_BAD_ARG, "negative size");
    if (n == 0) return CPE_OK;                      // empty input: the arrays may be null
    if (!cam_a || !cam_b || !uv_a || !uv_b || !xyz) return fail(CPE_BAD_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    // the camera indices select entries of the handle's camera table: check them on the host before any lane dereferences one
    std::vector<int32_t> ia(n), ib(n);
    HIPCHK(hipMemcpyAsync(ia.data(), cam_a, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(ib.data(), cam_b, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; i++)
        if (ia[i] < 0 || ia[i] >= h->hm.C || ib[i] >= h->hm.C) return fail(CPE_BAD_ARG, "camera index out of range");
    hipLaunchKernelGGL(k_triangulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->dm, n, cam_a, cam_b, uv_a, uv_b, depth, xyz);
    HIPCHK(hipGetLastError());
    return CPE_OK;
}

cpe_status cpe_tensorise_dlc(cpe_handle* h, int32_t N, int32_t n_slots, int32_t slot, const double* table, int32_t rows, int32_t parts,
                             int32_t first_row, const int32_t* part_of_marker, const double* inv_sigma, double thresh, double* meas, double* weight) {
    if (!h) return fail(CPE_BAD_ARG, "null argument");
    if (N < 0 || rows < 0 || parts <= 0) return fail(CPE_BAD_ARG, "bad size");
    if (n_slots <= 0 || slot < 0 || slot >= n_slots) return fail(CPE_BAD_ARG, "camera slot out of range");
    if (N == 0) return CPE_OK;
    if (!part_of_marker || !inv_sigma || !meas || !weight || (!table && rows > 0)) return fail(CPE_BAD_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    const long total = (long)N * h->hm.L;
    hipLaunchKernelGGL(k_tensorise_dlc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, N, h->hm.L, n_slots, slot, table, rows, parts,
                       first_row, part_of_marker, inv_sigma, thresh, meas, weight);
    HIPCHK(hipGetLastError());
    return CPE_OK;
}

static cpe_status ensure_ws(cpe_handle* h, int B, int N) {
    const size_t F = (size_t)B * N;
    if (F <= h->ws_frames && B <= h->ws_B) return CPE_OK;
    free_ws(h);
    const int nq = h->hm.nq, nu = h->hm.nu;
    HIPCHK(hipMalloc(&h->qbuf, sizeof(double) * 2 * F * h->hm.ns));
    HIPCHK(hipMalloc(&h->gambuf, sizeof(double) * 2 * F * (size_t)(GAM_STRIDE * (h->hm.nrev > 0 ? h->hm.nrev : 1))));
    HIPCHK(hipMalloc(&h->gbuf, sizeof(double) * 2 * F * nu));
    HIPCHK(hipMalloc(&h->Bbuf, sizeof(double) * 2 * F * nu * nu));
    HIPCHK(hipMalloc(&h->costbuf, sizeof(double) * 2 * F * COST_STRIDE));
    HIPCHK(hipMalloc(&h->mu, sizeof(double) * (F * (size_t)(h->hm.nb > 0 ? h->hm.nb : 1) * 2)));
    HIPCHK(hipMalloc(&h->Lbuf, sizeof(double) * F * (h->pb + 1) * nu * nu));
    HIPCHK(hipMalloc(&h->zbuf, sizeof(double) * F * nu));
    HIPCHK(hipMalloc(&h->gtbuf, sizeof(double) * F * nu));
    HIPCHK(hipMalloc(&h->dgbuf, sizeof(double) * F * nu));
    HIPCHK(hipMalloc(&h->cmax, sizeof(double) * F));
    HIPCHK(hipMalloc(&h->st, sizeof(SeqState) * B));
    HIPCHK(hipMalloc(&h->act, sizeof(int) * B));
    if (h->lr_window > 0) HIPCHK(hipMalloc(&h->Hlr, sizeof(double) * 2 * F * h->pb * nu * nu));
    h->ws_frames = F; h->ws_B = B;
    return CPE_OK;
}

static cpe_status ensure_ws(cpe_handle* h, int B, int N);

cpe_status cpe_eval_normal(cpe_handle* h, int32_t B, int32_t N, const double* q, const double* meas, const double* weight,
                           double* g, double* Bm, double* cost, double* gam, double* q_out) {
    if (!h || !q || !meas || !weight || !g || !Bm || !cost) return fail(CPE_BAD_ARG, "null argument");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    HIPCHK(hipSetDevice(h->device));
    cpe_status s = ensure_ws(h, B, N);
    if (s != CPE_OK) return s;
    const DevModel& m = h->hm;
    hipLaunchKernelGGL(k_state_init, dim3((unsigned)F), dim3(WAVE), 0, h->stream, h->dm, q, h->qbuf);
    HIPCHK(hipMemsetAsync(h->st, 0, sizeof(SeqState) * B, h->stream));
    HIPCHK(hipMemsetAsync(h->mu, 0, sizeof(double) * (F * (size_t)(m.nb > 0 ? m.nb : 1) * 2), h->stream));
    hipLaunchKernelGGL(FRAME_NORMAL(h->gmm_k == 0), dim3((unsigned)F), dim3(WAVE), lds_normal(m, h->gmm_k, h->gmm_dim), h->stream, h->dm, h->st, N, 1, F, h->qbuf, meas, weight,
                       h->gbuf, h->Bbuf, h->costbuf, h->mu, h->gambuf, h->pri, nullptr, nullptr, ShutterArgs{nullptr, nullptr, nullptr});
    HIPCHK(hipMemcpyAsync(g, h->gbuf, sizeof(double) * F * m.nu, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(Bm, h->Bbuf, sizeof(double) * F * m.nu * m.nu, hipMemcpyDeviceToDevice, h->stream));
    hipLaunchKernelGGL(k_gather_normal, dim3((unsigned)F), dim3(128), 0, h->stream, h->dm, F, h->qbuf, h->costbuf, h->gambuf, cost, gam, q_out);
    HIPCHK(hipGetLastError());
    return CPE_OK;
}

// restart of an LM run from the current iterate (shutter-delay outer loop): every sequence runs again, its buffers stay
__global__ void k_reset_status(SeqState* __restrict__ st, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { st[b].status = 0; st[b].al_pending = 0; }
}

// The LM loop of cpe_solve on the handle's workspace: cold start from q_init (device pointer) or, with q_init == nullptr, a restart from
// the current iterate of every sequence.  sh: shutter-delay buffers (all null = off).
static cpe_status lm_run(cpe_handle* h, int B, int N, const double* q_init, const double* meas, const double* weight, ShutterArgs sh) {
    const size_t F = (size_t)B * N;
    const DevModel& m = h->hm;
    const size_t Fw = F;   // buffers are laid out for exactly this call's F (strides use F)
    if (q_init) {
        hipLaunchKernelGGL(k_state_init, dim3((unsigned)F), dim3(WAVE), 0, h->stream, h->dm, q_init, h->qbuf);   // Euler q -> (q, alpha)
        HIPCHK(hipMemsetAsync(h->st, 0, sizeof(SeqState) * B, h->stream));
        HIPCHK(hipMemsetAsync(h->mu, 0, sizeof(double) * (F * (size_t)(m.nb > 0 ? m.nb : 1) * 2), h->stream));
    } else hipLaunchKernelGGL(k_reset_status, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->st, B);
    LmParams prm;
    prm.tol_step = h->opts.tol_step; prm.tol_cost = h->opts.tol_cost; prm.lambda0 = h->opts.lambda0; prm.B = B; prm.N = N;
    prm.bound_tol = h->opts.bound_tol; prm.max_outer = h->opts.max_outer; prm.max_iter = h->opts.max_iter;
    const size_t ldsn = lds_normal(m, h->gmm_k, h->gmm_dim, sh.tau != nullptr);
    const bool lr = h->lr_window > 0;
    // One LM iteration = k_frame_normal (+ k_lr_band) on the evaluated buffer, then k_lm_step, for the first *n_act sequences listed
    // in `act` (device arrays; nullptr = all B).  The grids are sized for `slots` sequences; workgroups past *n_act leave at once.
    auto iterate = [&](int first, const int* act, const int* n_act, int slots) {
        const unsigned gf = (unsigned)((size_t)slots * N);
        const double* hiu = lr ? reinterpret_cast<const double*>(reinterpret_cast<const char*>(h->pri) + offsetof(DevPriors, lr_HIu)) + CPE_NX * CPE_NX : nullptr;
        prof_begin(h, 0);
        hipLaunchKernelGGL(FRAME_NORMAL(h->gmm_k == 0 && sh.tau == nullptr), dim3(gf), dim3(WAVE), ldsn, h->stream, h->dm, h->st, N, first, Fw, h->qbuf, meas, weight, h->gbuf, h->Bbuf,
                           h->costbuf, h->mu, h->gambuf, h->pri, act, n_act, sh);
        prof_end(h);
        if (lr) {
            prof_begin(h, 1);
            hipLaunchKernelGGL(k_lr_band, dim3(gf), dim3(WAVE), 0, h->stream, h->dm, h->st, N, first, Fw, h->qbuf, h->gambuf, h->pri, h->pb, h->gbuf, h->Bbuf,
                               h->Hlr, h->costbuf, act, n_act);
            prof_end(h);
        }
        prof_begin(h, 2);
        if (h->pb == 3) hipLaunchKernelGGL((k_lm_step<3, 0>), dim3(slots), dim3(LM_THREADS), 0, h->stream, h->dm, h->st, prm, first, h->qbuf, h->gbuf, h->Bbuf, h->costbuf,
                                           h->Lbuf, h->zbuf, h->gtbuf, h->gambuf, h->Hlr, act, n_act, 2, sh.gx, h->dgbuf, hiu, h->lr_window);
        else hipLaunchKernelGGL((k_lm_step<4, 0>), dim3(slots), dim3(LM_THREADS), 0, h->stream, h->dm, h->st, prm, first, h->qbuf, h->gbuf, h->Bbuf, h->costbuf,
                                h->Lbuf, h->zbuf, h->gtbuf, h->gambuf, h->Hlr, act, n_act, 2, sh.gx, h->dgbuf, hiu, h->lr_window);
        prof_end(h);
        prof_begin(h, 7);
        if (h->pb == 3) hipLaunchKernelGGL((k_lm_back<3>), dim3(slots), dim3(2 * WAVE), 0, h->stream, h->dm, h->st, prm, h->qbuf, h->Lbuf, h->zbuf, h->gtbuf, h->dgbuf, act, n_act);
        else hipLaunchKernelGGL((k_lm_back<4>), dim3(slots), dim3(2 * WAVE), 0, h->stream, h->dm, h->st, prm, h->qbuf, h->Lbuf, h->zbuf, h->gtbuf, h->dgbuf, act, n_act);
        prof_end(h);
    };
    iterate(1, nullptr, nullptr, B);        // first evaluation and first step of every sequence
    HIPCHK(hipGetLastError());
    // Active window: k_lm_step runs one workgroup per sequence and its duration is the sequential depth of ONE sequence,
    // whatever the grid (3.3 ms for <= 256 workgroups, 4.4 ms for 512 = 2 per CU), so the launches are kept exactly full: before
    // every iteration k_build_act lists the first `window` unfinished sequences ON THE DEVICE, so a sequence that converges hands
    // its slot to the next waiting one at once and the host never waits for an iteration: it queues POLL iterations, then reads
    // the PREVIOUS batch's snapshot of the list length (pinned memory + event) -- the stream stays at least one batch ahead and
    // is never drained inside the loop.  The price is at most 2 POLL iterations of empty launches after the last sequence ends.
    const int window = std::min(B, h->n_cu * (h->pb == 3 ? 2 : 1));
    constexpr int POLL = 4;
    const long per_seq = (long)h->opts.max_iter + 2L * (h->opts.max_outer > 0 ? h->opts.max_outer : 0) + POLL;
    const long max_rounds = ((long)(B + window - 1) / window + 1) * per_seq;
    bool pending[2] = {false, false};
    int slot = 0;
    for (long it = 0; it < max_rounds; it += POLL) {
        for (int k = 0; k < POLL; k++) {
            prof_begin(h, 3);
            hipLaunchKernelGGL(k_build_act, dim3(1), dim3(256), 0, h->stream, h->st, B, window, h->act, h->n_act);
            prof_end(h);
            iterate(0, h->act, h->n_act, window);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h->poll_host + slot, h->n_act, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipEventRecord(h->poll_ev[slot], h->stream));
        pending[slot] = true;
        const int prev = 1 - slot;
        if (pending[prev]) {
            HIPCHK(hipEventSynchronize(h->poll_ev[prev]));
            pending[prev] = false;
            if (h->poll_host[prev] == 0) break;          // no sequence was running when that list was built
        }
        slot = prev;
    }
    return CPE_OK;
}

// outputs of a finished LM run (k_finalize) and the per-sequence statistics; `iters_extra[b]` iterations of earlier runs are added
static cpe_status lm_finish(cpe_handle* h, int B, int N, const double* meas, double* q, double* dq, double* ddq, double* positions, double* meas_err,
                            const double* tau, cpe_stats* stats, const std::vector<int>* iters_extra) {
    const size_t F = (size_t)B * N;
    const DevModel& m = h->hm;
    HIPCHK(hipMemsetAsync(h->cmax, 0, sizeof(double) * B, h->stream));
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)F), dim3(WAVE), lds_fk(m), h->stream, h->dm, h->st, N, F, h->qbuf, meas, q, dq, ddq, positions, meas_err,
                       reinterpret_cast<unsigned long long*>(h->cmax), tau);
    HIPCHK(hipGetLastError());
    std::vector<double> hc(B);
    std::vector<SeqState> hs(B);
    HIPCHK(hipMemcpyAsync(hs.data(), h->st, sizeof(SeqState) * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(hc.data(), h->cmax, sizeof(double) * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    prof_collect(h);
    cpe_status worst = CPE_OK;
    for (int b = 0; b < B; b++) {
        const SeqState& S = hs[b];
        const cpe_status sb = S.status == 1 ? CPE_OK : ((S.status == 0 || S.status == 3) ? CPE_MAX_ITER : CPE_NUMERICAL);
        if (sb > worst) worst = sb;
        if (stats) {
            cpe_stats& o = stats[b];
            o.status = sb; o.iterations = S.iters + (iters_extra ? (*iters_extra)[b] : 0); o.lambda = S.lambda; o.max_constraint = hc[b];
            o.max_bound_violation = S.maxviol; o.outer = S.outer; o._pad = 0;
            o.cost_meas = S.terms[0]; o.cost_model = S.terms[1]; o.cost_pose = S.terms[3]; o.cost_motion = S.terms[4];
            o.cost = h->opts.cost_scale * (S.terms[0] + S.terms[1] + S.terms[3] + S.terms[4]);
        }
    }
    return worst;
}

cpe_status cpe_solve(cpe_handle* h, int32_t B, int32_t N, const double* q_init, const double* meas, const double* weight,
                     double* q, double* dq, double* ddq, double* positions, double* meas_err, cpe_stats* stats) {
    if (!h || !q_init || !meas || !weight || !q) return fail(CPE_BAD_ARG, "null argument");
    if ((dq == nullptr) != (ddq == nullptr)) return fail(CPE_BAD_ARG, "dq and ddq must be given together");
    if (B < 0 || N < 0) return fail(CPE_BAD_ARG, "negative size");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    if (F > 0x7fffffffULL) return fail(CPE_BAD_ARG, "too many frames for one launch");
    HIPCHK(hipSetDevice(h->device));
    cpe_status s = ensure_ws(h, B, N);
    if (s != CPE_OK) return s;
    s = lm_run(h, B, N, q_init, meas, weight, ShutterArgs{nullptr, nullptr, nullptr});
    if (s != CPE_OK) return s;
    return lm_finish(h, B, N, meas, q, dq, ddq, positions, meas_err, nullptr, stats, nullptr);
}

// ---- shutter-delay estimation (include/cpe.h, cpe_solve_shutter) ------------------------------------------------------------------
// Anderson mixing (memory AA_MEM) of the delay iteration tau <- tau + step(tau): the plain iteration contracts slowly where all delays move
// together and the trajectory shifts in time to make up for it.  Host side, C - 1 unknowns per sequence.
namespace {
constexpr int AA_MEM = 4;
struct Anderson {
    int k = 0;
    double X[AA_MEM + 1][CPE_MAX_CAMS], F[AA_MEM + 1][CPE_MAX_CAMS], fn_prev = 0;
    void next(int n, const double* x, const double* f, double* xn) {
        double fn = 0;
        for (int i = 0; i < n; i++) fn += f[i] * f[i];
        if (k > 0 && fn > fn_prev) k = 0;                             // residual grew: drop the history
        fn_prev = fn;
        const int slot = k % (AA_MEM + 1);
        int mk = std::min(std::min(k, AA_MEM), n);
        for (int i = 0; i < n; i++) { X[slot][i] = x[i]; F[slot][i] = f[i]; xn[i] = x[i] + f[i]; }
        if (mk > 0) {
            double dX[AA_MEM][CPE_MAX_CAMS], dF[AA_MEM][CPE_MAX_CAMS], Mn[AA_MEM][AA_MEM], r[AA_MEM], tr = 0;
            for (int j = 0; j < mk; j++) {
                const int s1 = (k - j) % (AA_MEM + 1), s0 = (k - j - 1) % (AA_MEM + 1);
                for (int i = 0; i < n; i++) { dX[j][i] = X[s1][i] - X[s0][i]; dF[j][i] = F[s1][i] - F[s0][i]; }
            }
            for (int a = 0; a < mk; a++) {
                for (int b = 0; b < mk; b++) { double v = 0; for (int i = 0; i < n; i++) v += dF[a][i] * dF[b][i]; Mn[a][b] = v; }
                double v = 0; for (int i = 0; i < n; i++) v += dF[a][i] * f[i];
                r[a] = v; tr += Mn[a][a];
            }
            for (int a = 0; a < mk; a++) Mn[a][a] += 1e-10 * tr / mk + 1e-300;
            bool ok = true;
            for (int j = 0; j < mk && ok; j++) {
                double d = Mn[j][j];
                for (int t = 0; t < j; t++) d -= Mn[j][t] * Mn[j][t];
                if (!(d > 0)) { ok = false; break; }
                d = sqrt(d); Mn[j][j] = d;
                for (int i = j + 1; i < mk; i++) { double v = Mn[i][j]; for (int t = 0; t < j; t++) v -= Mn[i][t] * Mn[j][t]; Mn[i][j] = v / d; }
            }
            if (ok) {
                for (int i = 0; i < mk; i++) { double v = r[i]; for (int t = 0; t < i; t++) v -= Mn[i][t] * r[t]; r[i] = v / Mn[i][i]; }
                for (int i = mk - 1; i >= 0; i--) { double v = r[i]; for (int t = i + 1; t < mk; t++) v -= Mn[t][i] * r[t]; r[i] = v / Mn[i][i]; }
                for (int j = 0; j < mk; j++) for (int i = 0; i < n; i++) xn[i] -= r[j] * (dX[j][i] + dF[j][i]);
            }
        }
        k++;
    }
};
}  // namespace

cpe_status cpe_solve_shutter(cpe_handle* h, int32_t B, int32_t N, const double* q_init, const double* meas, const double* weight, double tau_bound,
                             int32_t max_rounds, double tol_tau, double* q, double* dq, double* ddq, double* positions, double* meas_err,
                             double* tau_out, cpe_stats* stats, int32_t* rounds_out) {
    if (!h || !q_init || !meas || !weight || !q || !tau_out) return fail(CPE_BAD_ARG, "null argument");
    if ((dq == nullptr) != (ddq == nullptr)) return fail(CPE_BAD_ARG, "dq and ddq must be given together");
    if (B < 0 || N < 0 || max_rounds < 1 || !(tau_bound > 0) || !(tol_tau > 0)) return fail(CPE_BAD_ARG, "bad size or tolerance");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    if (F > 0x7fffffffULL) return fail(CPE_BAD_ARG, "too many frames for one launch");
    HIPCHK(hipSetDevice(h->device));
    cpe_status s = ensure_ws(h, B, N);
    if (s != CPE_OK) return s;
    const int C = h->hm.C;
    DevBuf gx, rcb, step;
    HIPCHK(gx.alloc(2 * F * 6)); HIPCHK(rcb.alloc(2 * F * C * 9)); HIPCHK(step.alloc((size_t)B * C));
    HIPCHK(hipMemsetAsync(tau_out, 0, sizeof(double) * B * C, h->stream));      // the first camera is the reference (tau = 0, acinoset_misc.py:274-275); the others start at 0
    ShutterArgs sh{tau_out, gx.p, rcb.p};
    std::vector<int> iters(B, 0);
    std::vector<SeqState> hs(B);
    std::vector<double> hstep((size_t)B * C), htau((size_t)B * C, 0.0);
    std::vector<Anderson> mix(B);
    std::vector<char> settled(B, 0);
    int round = 0;
    for (; round < max_rounds; round++) {
        s = lm_run(h, B, N, round == 0 ? q_init : nullptr, meas, weight, sh);
        if (s != CPE_OK) return s;
        hipLaunchKernelGGL(k_shutter_step, dim3(B), dim3(WAVE), 0, h->stream, h->dm, h->st, N, F, h->qbuf, rcb.p, tau_out, step.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(hs.data(), h->st, sizeof(SeqState) * B, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(hstep.data(), step.p, sizeof(double) * B * C, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        bool all = true, moved = false;
        for (int b = 0; b < B; b++) {
            iters[b] += hs[b].iters;
            if (settled[b]) continue;                                 // its delays are a fixed point already (the trajectory solve above was a no-op restart)
            double worst = 0.0, tn[CPE_MAX_CAMS];
            for (int c = 1; c < C; c++) worst = std::max(worst, fabs(hstep[(size_t)b * C + c]));
            if (worst == 0.0 || hs[b].status == 2) { settled[b] = 1; continue; }   // nothing to move (N < 3), or a numerical failure: leave its delays alone
            mix[b].next(C, &htau[(size_t)b * C], &hstep[(size_t)b * C], tn);
            // the plain step underestimates the distance to the fixed point by the contraction factor (the trajectory has not followed
            // yet): the test is on the mixed update
            worst = 0.0;
            for (int c = 1; c < C; c++) {
                const double v = std::min(tau_bound, std::max(-tau_bound, tn[c]));
                worst = std::max(worst, fabs(v - htau[(size_t)b * C + c]));
                htau[(size_t)b * C + c] = v;
            }
            moved = true;
            if (worst < tol_tau) settled[b] = 1; else all = false;
        }
        if (moved) {
            HIPCHK(hipMemcpyAsync(tau_out, htau.data(), sizeof(double) * B * C, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));                  // htau is pageable: the copy has left the host buffer only now
        }
        if (all) { if (moved) round++; break; }
    }
    // one more solve at the final delays (they moved after the last one if max_rounds ran out; otherwise a restart that stops at once)
    s = lm_run(h, B, N, nullptr, meas, weight, sh);
    if (s != CPE_OK) return s;
    if (rounds_out) *rounds_out = round;
    return lm_finish(h, B, N, meas, q, dq, ddq, positions, meas_err, tau_out, stats, &iters);
}

// ---- physics-based trajectory model (include/cpe.h, cpe_solve_kinetic) -------------------------------------------------------------
void cpe_default_kinetic_options(cpe_kinetic_options* o, double fps, int32_t kinetic_dataset) {
    // o->dyn (inertias, feet, motors) is the caller's
    o->w_slack = 10e3; o->w_torque = 1.0; o->w_smooth = 0.1 / (fps * fps); o->friction = 0.8; o->force_max = 5.0; o->grfz_min = 0.01;
    o->foot_height_tol = kinetic_dataset ? 0.03 : 0.1; o->foot_height_min = 0.0; o->ground_height = 0.0; o->slip_max = 1.0; o->zvel_max = kinetic_dataset ? 1.0 : 0.0;
    o->slack_lo = -2.0; o->slack_hi = 2.0; o->kappa_slack = 1e6;
    o->reg_force = 1e-4; o->kappa_force = 1e5; o->kappa_height = 1e6; o->kappa_slip = 1e2; o->lm_force_damping = 10.0; o->lm_wall_damping = 10.0;
    o->inner_iterations = 30; o->_pad = 0;
}

static cpe_status build_kin(cpe_handle* h, const cpe_kinetic_options* opt, const double* grf_fix = nullptr, const double* tau_box = nullptr, const double* grf_box = nullptr) {
    const DevModel& m = h->hm;
    DevKin& K = h->hk;
    memset(&K, 0, sizeof(K));
    K.o = *opt;
    const cpe_dyn_options& d = opt->dyn;
    if (d.n_feet < 1 || d.n_feet > 4 || d.n_motors < 0 || d.n_motors > CPE_MAX_MOTORS) return fail(CPE_BAD_ARG, "kinetic options: feet / motors out of range");
    for (int f = 0; f < d.n_feet; f++) if (d.foot_marker[f] < 0 || d.foot_marker[f] >= m.L) return fail(CPE_BAD_ARG, "foot marker index out of range");
    for (int k = 0; k < d.n_motors; k++)
        if (d.motor_first[k] < 0 || d.motor_first[k] >= m.nl || d.motor_second[k] < 0 || d.motor_second[k] >= m.nl || d.motor_axis[k] < 0 || d.motor_axis[k] > 2)
            return fail(CPE_BAD_ARG, "motor definition out of range");
    if (!(opt->w_slack > 0) || !(opt->kappa_force > 0) || !(opt->kappa_height > 0) || !(opt->kappa_slip > 0) || !(opt->reg_force > 0) ||
        !(d.eom.gravity > 0) || opt->inner_iterations < 1)
        return fail(CPE_BAD_ARG, "kinetic options: weights, penalties and gravity must be positive");
    for (int k = 0; k < m.nu; k++) if (m.motion_w_u[k] != 0.0) return fail(CPE_BAD_ARG, "the physics-based model replaces the constant-acceleration cost: motion_w must be zero");
    if (h->lr_window > 0) return fail(CPE_BAD_ARG, "the physics-based model does not use the autoregressive motion prior (acinoset_opt.py:905-921)");
    K.nm = d.n_motors; K.nf = d.n_feet; K.nc = 0;
    for (int j = 0; j < m.nj; j++)
        for (int t = (m.joint_kind[j] == CPE_JOINT_REVOLUTE_Y ? 0 : 1); t < 2; t++) { K.con_joint[K.nc] = j; K.con_axis[K.nc] = t == 0 ? 0 : 2; K.nc++; }
    K.nlat = K.nm + K.nc + 3 * K.nf;
    if (K.nlat > KIN_NA_MAX || K.nm + K.nc > 52) return fail(CPE_BAD_ARG, "more than 60 node forces");
    K.nrow = m.nq + 4 * K.nf + 3 * m.L;          // slack | foot heights | foot velocities (x, y, z) | second differences of the markers
    // the evaluation slots of k_dyn_eval are laid out for the reference's skeleton (KS_* in cpe_kinetic.hip.inc): anything larger is refused here,
    // not truncated there
    if (m.nq > KS_NQ || m.nl > KS_NL || m.L > KS_L || m.ns > KS_NS || m.nrev > 32 || K.nrow > KS_NROW || K.nrow > KIN_ROWS_MAX)
        return fail(CPE_BAD_ARG, "skeleton too large for the kinetic kernels (at most 54 coordinates, 17 links, 24 markers)");
    double mt = 0; for (int i = 0; i < m.nl; i++) mt += m.mass[i];
    K.Mg = mt * d.eom.gravity; K.h = h->opts.h; K.ih = 1.0 / h->opts.h;
    for (int i = 0; i < m.nl; i++) { uint32_t mask = 0; for (int j = 0; j < m.nl; j++) { int a = j; while (a >= 0 && a != i) a = m.parent[a]; if (a == i) mask |= 1u << j; } K.sub_mask[i] = mask; }
    K.grf_fix = grf_fix;
    K.grf_box = grf_box;
    K.tau_box = tau_box; K.mu_tau = h->kmut;            // (the workspace is sized before this is called)
    // ---- tables of the analytic Jacobian (k_dyn_jac): ancestry of the link pairs, moments of every subtree about its link's origin
    {
        const int nl = m.nl;
        auto is_anc = [&](int a, int i) { int j = m.parent[i]; while (j >= 0) { if (j == a) return true; j = m.parent[j]; } return false; };      // a strict ancestor of i
        auto toward = [&](int up, int down) { int j = down; while (m.parent[j] != up) j = m.parent[j]; return j; };                                 // child of `up` on the path to `down`
        K.nblk = 0;
        for (int i = 0; i < nl; i++) for (int k = 0; k < nl; k++) {
            int rel = 0, tw = 0;
            if (i == k) rel = 3;
            else if (is_anc(k, i)) { rel = 1; tw = toward(k, i); }
            else if (is_anc(i, k)) { rel = 2; tw = toward(i, k); }
            K.rel[i][k] = (int8_t)rel; K.toward[i][k] = (int8_t)tw; K.blk[i][k] = -1;
            if (rel) {
                if (K.nblk >= 96) return fail(CPE_BAD_ARG, "skeleton too deep for the kinetic kernels (more than 96 related link pairs)");
                K.blk[i][k] = (int16_t)K.nblk; K.blk_i[K.nblk] = (int8_t)i; K.blk_k[K.nblk] = (int8_t)k; K.nblk++;
            }
        }
        for (int bl = 0; bl < K.nblk; bl++) {
            const int i = K.blk_i[bl], k = K.blk_k[bl];
            K.blk_word[bl] = (uint32_t)i | ((uint32_t)k << 8) | ((uint32_t)(uint8_t)K.rel[i][k] << 16) | ((uint32_t)(uint8_t)K.toward[i][k] << 24);
            int nmo = 0, nco = 0;
            for (int mo = 0; mo < K.nm; mo++) {
                const int a1 = d.motor_first[mo], a2 = d.motor_second[mo];
                if (a1 == a2 || (i != a1 && i != a2) || (k != i && k != a1)) continue;
                if (nmo >= KJ_LMAX) return fail(CPE_BAD_ARG, "more motors on one link pair than the kinetic kernels are sized for");
                K.bm[bl][nmo++] = (uint8_t)mo;
            }
            for (int r = 0; r < K.nc; r++) {
                const int p = m.joint_parent[K.con_joint[r]], ch = m.joint_child[K.con_joint[r]];
                if (!((i == p || i == ch) && (k == p || k == ch))) continue;
                if (nco >= KJ_LMAX) return fail(CPE_BAD_ARG, "more joint equalities on one link pair than the kinetic kernels are sized for");
                K.bc[bl][nco++] = (uint8_t)r;
            }
            K.bm_n[bl] = (uint8_t)nmo; K.bc_n[bl] = (uint8_t)nco;
        }
        std::vector<double> msub(nl, 0.0);
        for (int i = 0; i < nl; i++) for (int j = 0; j < nl; j++) if (j == i || is_anc(i, j)) msub[i] += m.mass[j];
        K.mtot = mt;
        for (int i = 0; i < nl; i++) {
            for (int d = 0; d < 3; d++) K.s1[i][d] = m.mass[i] * m.com[i][d];
            for (int a = 0; a < 3; a++) for (int b2 = 0; b2 < 3; b2++) K.S2[i][3 * a + b2] = m.mass[i] * m.com[i][a] * m.com[i][b2];
            for (int c = 0; c < nl; c++) if (m.parent[c] == i) {
                for (int d = 0; d < 3; d++) K.s1[i][d] += msub[c] * m.attach[c][d];
                for (int a = 0; a < 3; a++) for (int b2 = 0; b2 < 3; b2++) K.S2[i][3 * a + b2] += msub[c] * m.attach[c][a] * m.attach[c][b2];
            }
            K.leg_of_link[i] = -1; K.hooke_of_link[i] = -1;
        }
        for (int r = 0; r < m.nrev; r++) K.leg_of_link[m.rev_child[r]] = (int8_t)r;
        K.nh = 0;
        for (int j = 0; j < m.nj; j++) if (m.joint_kind[j] == CPE_JOINT_HOOKE_YZ) {          // joints are stored parents first (cpe_model.h)
            if (K.nh >= 4) return fail(CPE_BAD_ARG, "more than four Hooke joints");
            K.hk_parent[K.nh] = (int8_t)m.joint_parent[j]; K.hk_child[K.nh] = (int8_t)m.joint_child[j]; K.hooke_of_link[m.joint_child[j]] = (int8_t)K.nh; K.nh++;
        }
    }
    for (int l = 0; l < m.L; l++) if (m.chain_len[l] > KJ_CHAIN) return fail(CPE_BAD_ARG, "skeleton too deep for the kinetic kernels (a marker more than 5 links from the root)");
    K.mu_slack = h->kmus; K.sbox = (opt->slack_hi < 1e9 || opt->slack_lo > -1e9) ? 1 : 0;
    if (K.sbox && (!(opt->kappa_slack > 0) || !(opt->slack_lo < opt->slack_hi))) return fail(CPE_BAD_ARG, "kinetic options: slack box needs lo < hi and a positive penalty");
    if (!h->dk) HIPCHK(hipMalloc(&h->dk, sizeof(DevKin)));
    HIPCHK(hipMemcpyAsync(h->dk, &K, sizeof(DevKin), hipMemcpyHostToDevice, h->stream));
    return CPE_OK;
}

static_assert(KE_R1 >= KS_NQ * KIN_LS && KE_R1 >= KIN_SLOT && KE_R2 >= 3 * KIN_SLOT, "regions of k_dyn_eval hold A, an evaluation slot / the three base slots");
static size_t lds_kin_eval() { return sizeof(double) * ((sizeof(KinShared) + 7) / 8 + (size_t)KE_R1 + KE_R2); }
static size_t lds_kin_assemble() { return sizeof(double) * ((size_t)KA_RC * KA_RS + (size_t)KS_NQ * KIN_LS + 2 * KS_NROW); }
static size_t lds_kin_schur() { return sizeof(double) * ((size_t)(KIN_NA_MAX + KIN_NC3) * KIN_MS + KIN_LS); }

static cpe_status ensure_kws(cpe_handle* h, int B, int N) {
    const size_t F = (size_t)B * N;
    if (F <= h->kws_frames) return CPE_OK;
    free_kws(h);
    const int BB = CPE_NX * CPE_NX;
    HIPCHK(hipMalloc(&h->fbuf, sizeof(double) * 2 * F * KIN_LS));
    HIPCHK(hipMalloc(&h->kmu, sizeof(double) * F * 4 * KIN_MU));
    HIPCHK(hipMalloc(&h->kmut, sizeof(double) * F * 2 * CPE_MAX_MOTORS));
    HIPCHK(hipMalloc(&h->kmus, sizeof(double) * F * 2 * CPE_MAX_NQ));
    HIPCHK(hipMalloc(&h->Jbuf, sizeof(double) * F * KIN_JSTRIDE));
    HIPCHK(hipMalloc(&h->Abuf, sizeof(double) * F * CPE_MAX_NQ * KIN_LS));
    HIPCHK(hipMalloc(&h->pieces, sizeof(double) * 2 * F * KIN_PIECE));
    HIPCHK(hipMalloc(&h->pmeta, sizeof(int) * 2 * F * (KIN_LS + 1)));
    HIPCHK(hipMalloc(&h->gTb, sizeof(double) * 2 * (F + 2) * KIN_NC3));
    HIPCHK(hipMalloc(&h->dstat, sizeof(double) * 2 * F * KIN_STAT));
    HIPCHK(hipMalloc(&h->slackb, sizeof(double) * 2 * F * CPE_MAX_NQ));
    HIPCHK(hipMalloc(&h->Tbuf, sizeof(double) * (F + 2) * 6 * BB));
    HIPCHK(hipMalloc(&h->gk, sizeof(double) * F * CPE_NX));
    HIPCHK(hipMalloc(&h->Bk, sizeof(double) * F * BB));
    HIPCHK(hipMalloc(&h->Hk, sizeof(double) * F * 3 * BB));
    const void* ks[] = {(const void*)&k_dyn_eval, (const void*)&k_dyn_assemble, (const void*)&k_dyn_schur, (const void*)&k_dyn_jac};
    for (const void* k : ks) HIPCHK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    h->kws_frames = F;
    return CPE_OK;
}

// one evaluation pass of the physics terms on the evaluated buffer (after k_frame_normal): rows, node forces, cost
static void launch_dyn_eval(cpe_handle* h, int N, int first, size_t Fw, const int32_t* stance, const int* act, const int* n_act, int slots) {
    const unsigned gf = (unsigned)((size_t)slots * N);
    prof_begin(h, 5);
    hipLaunchKernelGGL(k_dyn_eval, dim3(gf), dim3(KIN_THREADS), lds_kin_eval(), h->stream, h->dm, h->dk, h->st, N, first, Fw, h->qbuf, stance, h->fbuf, h->kmu, h->costbuf,
                       h->Jbuf, h->Abuf, h->pieces, h->pmeta, h->dstat, h->slackb, act, n_act);
    prof_end(h);
}
// Jacobian and second-order pieces of the CURRENT iterate -- after the accept step, and only for sequences whose iterate is new: a rejected trial
// costs its evaluation only (a quarter to a third of the iterations of a physics-based solve are rejections)
static void launch_dyn_pieces(cpe_handle* h, int N, int first, size_t Fw, const int* act, const int* n_act, int slots) {
    const unsigned gf = (unsigned)((size_t)slots * N);
    prof_begin(h, 10);
    hipLaunchKernelGGL(k_dyn_jac, dim3(gf), dim3(KJ_THREADS), sizeof(double) * KJ_DOUBLES, h->stream, h->dm, h->dk, h->st, N, first, Fw, h->qbuf, h->fbuf, h->Jbuf, act, n_act);
    prof_end(h);
    prof_begin(h, 8);
    hipLaunchKernelGGL(k_dyn_assemble, dim3(gf), dim3(KIN_THREADS), lds_kin_assemble(), h->stream, h->dm, h->dk, h->st, N, first, Fw, h->Jbuf, h->Abuf, h->pieces, h->pmeta,
                       h->gTb, act, n_act);
    prof_end(h);
}

cpe_status cpe_solve_kinetic(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q_init, const double* meas,
                             const double* weight, const int32_t* stance, double* q, double* dq, double* ddq, double* positions,
                             double* meas_err, double* tau, double* lambda, double* grf, double* slack, cpe_stats* stats, cpe_kinetic_stats* kstats) {
    return cpe_solve_kinetic_fixed(h, opt, B, N, q_init, meas, weight, stance, nullptr, q, dq, ddq, positions, meas_err, tau, lambda, grf, slack, stats, kstats);
}

static cpe_status solve_kinetic_impl(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q_init, const double* meas,
                                     const double* weight, const int32_t* stance, const double* grf_fixed, const double* tau_box, const double* grf_box, double* q, double* dq, double* ddq,
                                     double* positions, double* meas_err, double* tau, double* lambda, double* grf, double* slack, cpe_stats* stats,
                                     cpe_kinetic_stats* kstats);
cpe_status cpe_solve_kinetic_fixed(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q_init, const double* meas,
                                   const double* weight, const int32_t* stance, const double* grf_fixed, double* q, double* dq, double* ddq,
                                   double* positions, double* meas_err, double* tau, double* lambda, double* grf, double* slack, cpe_stats* stats,
                                   cpe_kinetic_stats* kstats) {
    return solve_kinetic_impl(h, opt, B, N, q_init, meas, weight, stance, grf_fixed, nullptr, nullptr, q, dq, ddq, positions, meas_err, tau, lambda, grf, slack, stats, kstats);
}
cpe_status cpe_solve_kinetic_force_box(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q_init, const double* meas,
                                       const double* weight, const int32_t* stance, const double* grf_box, double* q, double* dq, double* ddq,
                                       double* positions, double* meas_err, double* tau, double* lambda, double* grf, double* slack, cpe_stats* stats,
                                       cpe_kinetic_stats* kstats) {
    if (!grf_box) return fail(CPE_BAD_ARG, "null argument");
    return solve_kinetic_impl(h, opt, B, N, q_init, meas, weight, stance, nullptr, nullptr, grf_box, q, dq, ddq, positions, meas_err, tau, lambda, grf, slack, stats, kstats);
}
cpe_status cpe_solve_kinetic_bounded(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q_init, const double* meas,
                                     const double* weight, const int32_t* stance, const double* tau_box, double* q, double* dq, double* ddq,
                                     double* positions, double* meas_err, double* tau, double* lambda, double* grf, double* slack, cpe_stats* stats,
                                     cpe_kinetic_stats* kstats) {
    if (!tau_box) return fail(CPE_BAD_ARG, "null argument");
    return solve_kinetic_impl(h, opt, B, N, q_init, meas, weight, stance, nullptr, tau_box, nullptr, q, dq, ddq, positions, meas_err, tau, lambda, grf, slack, stats, kstats);
}
static cpe_status solve_kinetic_impl(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q_init, const double* meas,
                                     const double* weight, const int32_t* stance, const double* grf_fixed, const double* tau_box, const double* grf_box, double* q, double* dq, double* ddq,
                                     double* positions, double* meas_err, double* tau, double* lambda, double* grf, double* slack, cpe_stats* stats,
                                     cpe_kinetic_stats* kstats) {
    if (!h || !opt || !q_init || !meas || !weight || !stance || !q) return fail(CPE_BAD_ARG, "null argument");
    if ((dq == nullptr) != (ddq == nullptr)) return fail(CPE_BAD_ARG, "dq and ddq must be given together");
    if (B < 0 || N < 0) return fail(CPE_BAD_ARG, "negative size");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    if (F > 0x7fffffffULL) return fail(CPE_BAD_ARG, "too many frames for one launch");
    if (h->pb != 3) return fail(CPE_BAD_ARG, "the physics-based model runs on the half-bandwidth-3 solver");
    HIPCHK(hipSetDevice(h->device));
    cpe_status s = ensure_ws(h, B, N);
    if (s != CPE_OK) return s;
    s = ensure_kws(h, B, N);
    if (s != CPE_OK) return s;
    s = build_kin(h, opt, grf_fixed, tau_box, grf_box);
    if (s != CPE_OK) return s;
    const DevModel& m = h->hm;
    const size_t Fw = F;
    hipLaunchKernelGGL(k_state_init, dim3((unsigned)F), dim3(WAVE), 0, h->stream, h->dm, q_init, h->qbuf);
    HIPCHK(hipMemsetAsync(h->st, 0, sizeof(SeqState) * B, h->stream));
    HIPCHK(hipMemsetAsync(h->mu, 0, sizeof(double) * (F * (size_t)(m.nb > 0 ? m.nb : 1) * 2), h->stream));
    HIPCHK(hipMemsetAsync(h->fbuf, 0, sizeof(double) * 2 * F * KIN_LS, h->stream));
    HIPCHK(hipMemsetAsync(h->kmu, 0, sizeof(double) * F * 4 * KIN_MU, h->stream));
    HIPCHK(hipMemsetAsync(h->kmus, 0, sizeof(double) * F * 2 * CPE_MAX_NQ, h->stream));
    if (tau_box) HIPCHK(hipMemsetAsync(h->kmut, 0, sizeof(double) * F * 2 * CPE_MAX_MOTORS, h->stream));
    LmParams prm;
    prm.tol_step = h->opts.tol_step; prm.tol_cost = h->opts.tol_cost; prm.lambda0 = h->opts.lambda0; prm.B = B; prm.N = N;
    prm.bound_tol = h->opts.bound_tol; prm.max_outer = h->opts.max_outer; prm.max_iter = h->opts.max_iter;
    const size_t ldsn = lds_normal(m, h->gmm_k, h->gmm_dim);
    // One iteration: per-frame terms and physics terms of the evaluated buffer, accept / reject (new damping), elimination of the node
    // forces at that damping for the CURRENT iterate, band system, factor + solve + next trial.
    auto iterate = [&](int first, const int* act, const int* n_act, int slots) {
        const unsigned gf = (unsigned)((size_t)slots * N);
        prof_begin(h, 0);
        hipLaunchKernelGGL(FRAME_NORMAL(h->gmm_k == 0), dim3(gf), dim3(WAVE), ldsn, h->stream, h->dm, h->st, N, first, Fw, h->qbuf, meas, weight, h->gbuf, h->Bbuf,
                           h->costbuf, h->mu, h->gambuf, h->pri, act, n_act, ShutterArgs{nullptr, nullptr, nullptr});
        prof_end(h);
        launch_dyn_eval(h, N, first, Fw, stance, act, n_act, slots);
        prof_begin(h, 2);
        hipLaunchKernelGGL((k_lm_step<3, 1>), dim3(slots), dim3(LM_THREADS), 0, h->stream, h->dm, h->st, prm, first, h->qbuf, h->gbuf, h->Bbuf, h->costbuf,
                           h->Lbuf, h->zbuf, h->gtbuf, h->gambuf, nullptr, act, n_act, 2, nullptr, h->dgbuf);
        prof_end(h);
        launch_dyn_pieces(h, N, 0, Fw, act, n_act, slots);        // (`which` = 0 also in the first pass: the accept step has just marked every sequence new)
        prof_begin(h, 9);
        hipLaunchKernelGGL(k_dyn_schur, dim3(gf), dim3(KIN_THREADS), lds_kin_schur(), h->stream, h->dk, h->st, N, Fw, h->pieces, h->pmeta, h->fbuf, h->kmu, stance, h->Tbuf, act, n_act);
        prof_end(h);
        prof_begin(h, 6);
        hipLaunchKernelGGL(k_dyn_gather, dim3(gf), dim3(KIN_THREADS), 0, h->stream, h->st, N, Fw, h->gbuf, h->Bbuf, h->Tbuf, h->gTb, h->gk, h->Bk, h->Hk, act, n_act);
        prof_end(h);
        prof_begin(h, 2);
        hipLaunchKernelGGL((k_lm_step<3, 2>), dim3(slots), dim3(LM_THREADS), 0, h->stream, h->dm, h->st, prm, first, h->qbuf, h->gk, h->Bk, h->costbuf,
                           h->Lbuf, h->zbuf, h->gtbuf, h->gambuf, h->Hk, act, n_act, 1, nullptr, h->dgbuf);
        prof_end(h);
        prof_begin(h, 7);
        hipLaunchKernelGGL((k_lm_back<3>), dim3(slots), dim3(2 * WAVE), 0, h->stream, h->dm, h->st, prm, h->qbuf, h->Lbuf, h->zbuf, h->gtbuf, h->dgbuf, act, n_act);
        prof_end(h);
    };
    iterate(1, nullptr, nullptr, B);
    HIPCHK(hipGetLastError());
    const int window = std::min(B, h->n_cu * 2);
    constexpr int POLL = 4;
    const long per_seq = (long)h->opts.max_iter + 2L * (h->opts.max_outer > 0 ? h->opts.max_outer : 0) + POLL;
    const long max_rounds = ((long)(B + window - 1) / window + 1) * per_seq;
    bool pending[2] = {false, false};
    int slot = 0;
    for (long it = 0; it < max_rounds; it += POLL) {
        for (int k = 0; k < POLL; k++) {
            hipLaunchKernelGGL(k_build_act, dim3(1), dim3(256), 0, h->stream, h->st, B, window, h->act, h->n_act);
            iterate(0, h->act, h->n_act, window);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h->poll_host + slot, h->n_act, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipEventRecord(h->poll_ev[slot], h->stream));
        pending[slot] = true;
        const int prev = 1 - slot;
        if (pending[prev]) {
            HIPCHK(hipEventSynchronize(h->poll_ev[prev]));
            pending[prev] = false;
            if (h->poll_host[prev] == 0) break;
        }
        slot = prev;
    }
    HIPCHK(hipMemsetAsync(h->cmax, 0, sizeof(double) * B, h->stream));
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)F), dim3(WAVE), lds_fk(m), h->stream, h->dm, h->st, N, Fw, h->qbuf, meas, q, dq, ddq, positions, meas_err,
                       reinterpret_cast<unsigned long long*>(h->cmax));
    hipLaunchKernelGGL(k_dyn_outputs, dim3((unsigned)F), dim3(64), 0, h->stream, h->dk, h->st, N, Fw, h->fbuf, h->Abuf, nullptr, tau, lambda, grf);
    HIPCHK(hipGetLastError());
    std::vector<double> hc(B);
    std::vector<SeqState> hs(B);
    HIPCHK(hipMemcpyAsync(hs.data(), h->st, sizeof(SeqState) * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(hc.data(), h->cmax, sizeof(double) * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    prof_collect(h);
    // slack and the per-node statistics live in the buffer of each sequence's final iterate
    std::vector<double> hd;
    if (kstats) hd.resize(2 * F * KIN_STAT);
    if (kstats) HIPCHK(hipMemcpyAsync(hd.data(), h->dstat, sizeof(double) * 2 * F * KIN_STAT, hipMemcpyDeviceToHost, h->stream));
    std::vector<int> hit;                                  // last word of every node's meta record: Newton iterations of its force solve
    if (kstats) {
        hit.resize(2 * F);
        HIPCHK(hipMemcpy2DAsync(hit.data(), sizeof(int), h->pmeta + KIN_LS, sizeof(int) * (KIN_LS + 1), sizeof(int), 2 * F, hipMemcpyDeviceToHost, h->stream));
    }
    if (slack)
        for (int b = 0; b < B; b++)
            HIPCHK(hipMemcpy2DAsync(slack + (size_t)b * N * m.nq, sizeof(double) * m.nq, h->slackb + ((size_t)hs[b].cur * F + (size_t)b * N) * CPE_MAX_NQ,
                                    sizeof(double) * CPE_MAX_NQ, sizeof(double) * m.nq, N, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    cpe_status worst = CPE_OK;
    for (int b = 0; b < B; b++) {
        const SeqState& S = hs[b];
        const cpe_status sb = S.status == 1 ? CPE_OK : ((S.status == 0 || S.status == 3) ? CPE_MAX_ITER : CPE_NUMERICAL);
        if (sb > worst) worst = sb;
        if (stats) {
            cpe_stats& o = stats[b];
            o.status = sb; o.iterations = S.iters; o.lambda = S.lambda; o.max_constraint = hc[b];
            o.max_bound_violation = S.maxviol; o.outer = S.outer; o._pad = 0;
            o.cost_meas = S.terms[0]; o.cost_model = S.terms[4]; o.cost_pose = S.terms[3]; o.cost_motion = 0.0;
            o.cost = h->opts.cost_scale * (S.terms[0] + S.terms[3] + S.terms[4]);
        }
        if (kstats) {
            cpe_kinetic_stats& k = kstats[b];
            memset(&k, 0, sizeof(k));
            for (int n = 0; n < N; n++) {
                const double* d = hd.data() + ((size_t)S.cur * F + (size_t)b * N + n) * KIN_STAT;
                k.cost_eom += d[0]; k.cost_torque += d[1]; k.cost_energy += d[3];
                k.max_slack = std::max(k.max_slack, d[5]); k.max_base_rows = std::max(k.max_base_rows, d[6]); k.max_violation = std::max(k.max_violation, d[7]);
                k.inner_max = std::max(k.inner_max, hit[(size_t)S.cur * F + (size_t)b * N + n]);
            }
        }
    }
    return worst;
}

// diagnostic building block (as cpe_eval_normal): one evaluation of the physics terms at Euler q, multipliers zero, forces from a cold start.
// Device pointers: f [B][N][64] node forces, stat [B][N][8], g [B][N][84], Huu [B][N][84][84], Hfu [B][N][64][84], Hff [B][N][64][64];
// meta int32 [B][N][65] = (number of free node forces, their indices).  (What ASL would hand IPOPT for the physics constraints of one node.)
cpe_status cpe_eval_kinetic_nodes(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q, const double* meas, const double* weight,
                                  const int32_t* stance, double* f, double* stat, double* g, double* Huu, double* Hfu, double* Hff, int32_t* meta) {
    if (!h || !opt || !q || !meas || !weight || !stance) return fail(CPE_BAD_ARG, "null argument");
    if (B < 0 || N < 0) return fail(CPE_BAD_ARG, "negative size");
    const size_t F = (size_t)B * N;
    if (F == 0) return CPE_OK;
    if (F > 0x7fffffffULL) return fail(CPE_BAD_ARG, "too many frames for one launch");
    if (h->pb != 3) return fail(CPE_BAD_ARG, "the physics-based model runs on the half-bandwidth-3 solver");
    HIPCHK(hipSetDevice(h->device));
    cpe_status s = ensure_ws(h, B, N);                     // the workspaces first: build_kin records pointers into them
    if (s != CPE_OK) return s;
    if ((s = ensure_kws(h, B, N)) != CPE_OK) return s;
    if ((s = build_kin(h, opt)) != CPE_OK) return s;
    const DevModel& m = h->hm;
    hipLaunchKernelGGL(k_state_init, dim3((unsigned)F), dim3(WAVE), 0, h->stream, h->dm, q, h->qbuf);
    HIPCHK(hipMemsetAsync(h->st, 0, sizeof(SeqState) * B, h->stream));
    HIPCHK(hipMemsetAsync(h->mu, 0, sizeof(double) * (F * (size_t)(m.nb > 0 ? m.nb : 1) * 2), h->stream));
    HIPCHK(hipMemsetAsync(h->fbuf, 0, sizeof(double) * 2 * F * KIN_LS, h->stream));
    HIPCHK(hipMemsetAsync(h->kmu, 0, sizeof(double) * F * 4 * KIN_MU, h->stream));
    HIPCHK(hipMemsetAsync(h->kmus, 0, sizeof(double) * F * 2 * CPE_MAX_NQ, h->stream));
    hipLaunchKernelGGL(FRAME_NORMAL(h->gmm_k == 0), dim3((unsigned)F), dim3(WAVE), lds_normal(m, h->gmm_k, h->gmm_dim), h->stream, h->dm, h->st, N, 1, F, h->qbuf, meas, weight,
                       h->gbuf, h->Bbuf, h->costbuf, h->mu, h->gambuf, h->pri, nullptr, nullptr, ShutterArgs{nullptr, nullptr, nullptr});
    launch_dyn_eval(h, N, 1, F, stance, nullptr, nullptr, B);
    launch_dyn_pieces(h, N, 1, F, nullptr, nullptr, B);
    HIPCHK(hipGetLastError());
    if (f) HIPCHK(hipMemcpyAsync(f, h->fbuf, sizeof(double) * F * KIN_LS, hipMemcpyDeviceToDevice, h->stream));
    if (stat) HIPCHK(hipMemcpyAsync(stat, h->dstat, sizeof(double) * F * KIN_STAT, hipMemcpyDeviceToDevice, h->stream));
    if (g) HIPCHK(hipMemcpyAsync(g, h->gTb, sizeof(double) * F * KIN_NC3, hipMemcpyDeviceToDevice, h->stream));
    if (meta) HIPCHK(hipMemcpyAsync(meta, h->pmeta, sizeof(int) * F * (KIN_LS + 1), hipMemcpyDeviceToDevice, h->stream));
    const size_t w = sizeof(double);
    if (Huu) HIPCHK(hipMemcpy2DAsync(Huu, w * KIN_NC3 * KIN_NC3, h->pieces, w * KIN_PIECE, w * KIN_NC3 * KIN_NC3, F, hipMemcpyDeviceToDevice, h->stream));
    if (Hfu) HIPCHK(hipMemcpy2DAsync(Hfu, w * KIN_LS * KIN_NC3, h->pieces + KIN_NC3 * KIN_NC3, w * KIN_PIECE, w * KIN_LS * KIN_NC3, F, hipMemcpyDeviceToDevice, h->stream));
    if (Hff) HIPCHK(hipMemcpy2DAsync(Hff, w * KIN_LS * KIN_LS, h->pieces + KIN_NC3 * KIN_NC3 + KIN_LS * KIN_NC3, w * KIN_PIECE, w * KIN_LS * KIN_LS, F, hipMemcpyDeviceToDevice, h->stream));
    return CPE_OK;
}

#ifdef CPE_LM_STAMPS
// diagnostic build only: table sizes and LDS bytes of the per-frame kernels for a model (needs no GPU)
cpe_status cpe_debug_footprint(const cpe_skeleton* skel, const cpe_camera* cams, int32_t n_cams, const cpe_options* opts, int64_t* out16) {
    std::vector<DevModel> mv(1);
    cpe_status s = build_model(skel, cams, n_cams, opts, mv[0]);
    if (s != CPE_OK) return s;
    const DevModel& m = mv[0];
    const int64_t v[16] = {m.nq, m.ns, m.nu, m.ndep, m.nrev, m.S, m.ss_n, m.sv_n, m.mc_total, (int64_t)lds_normal(m), (int64_t)sizeof(DevModel), m.L, m.C, m.nl, m.nb, 0};
    for (int i = 0; i < 16; i++) out16[i] = v[i];
    return CPE_OK;
}
// diagnostic build only: the Jacobian of the rows of frame `frame` of the last cpe_eval_kinetic_nodes / solve, column-major [84][KIN_ROWS_MAX] + row gradient + row weight
cpe_status cpe_debug_kinetic_jacobian(cpe_handle* h, int64_t frame, double* out) {
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->Jbuf + (size_t)frame * KIN_JSTRIDE, sizeof(double) * KIN_JSTRIDE, hipMemcpyDeviceToHost));
    return CPE_OK;
}
cpe_status cpe_debug_fn_stamps(unsigned long long* out16) {
    HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fn_stamps), sizeof(unsigned long long) * 16));
    unsigned long long z[16] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_fn_stamps), z, sizeof(z)));
    return CPE_OK;
}
// diagnostic build only: per-phase shader-clock totals accumulated by block 0 of k_lm_step since the last call
cpe_status cpe_debug_lm_stamps(unsigned long long* out32) {
    HIPCHK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_lm_stamps), sizeof(unsigned long long) * 32));
    unsigned long long z[32] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_lm_stamps), z, sizeof(z)));
    return CPE_OK;
}
#endif

// ---- host-pointer wrappers: stage through HBM (PCIe-inclusive; never the benchmarked path) -------------

cpe_status cpe_eval_resjac_host(cpe_handle* h, int32_t B, int32_t N, const double* q, const double* meas, const double* weight,
                                double* r, double* J, double* eps, double* cost) {
    if (!h || !q || !meas || !r || !J || !eps) return fail(CPE_BAD_ARG, "null argument");
    const size_t F = (size_t)B * N; const DevModel& m = h->hm;
    if (F == 0) return CPE_OK;
    HIPCHK(hipSetDevice(h->device));
    DevBuf dq_, dm_, dw_, dr_, dJ_, de_, dc_;
    const size_t nm = F * m.C * m.L;
    HIPCHK(dq_.alloc(F * m.nq)); HIPCHK(dm_.alloc(nm * 2)); HIPCHK(dw_.alloc(nm)); HIPCHK(dr_.alloc(nm * 2));
    HIPCHK(dJ_.alloc(F * m.C * m.S * 2)); HIPCHK(de_.alloc(F * m.nq)); HIPCHK(dc_.alloc(F));
    HIPCHK(hipMemcpyAsync(dq_.p, q, sizeof(double) * F * m.nq, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dm_.p, meas, sizeof(double) * nm * 2, hipMemcpyHostToDevice, h->stream));
    if (weight) HIPCHK(hipMemcpyAsync(dw_.p, weight, sizeof(double) * nm, hipMemcpyHostToDevice, h->stream));
    cpe_status s = cpe_eval_resjac(h, B, N, dq_.p, dm_.p, weight ? dw_.p : nullptr, dr_.p, dJ_.p, de_.p, (cost && weight) ? dc_.p : nullptr);
    if (s != CPE_OK) return s;
    HIPCHK(hipMemcpyAsync(r, dr_.p, sizeof(double) * nm * 2, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(J, dJ_.p, sizeof(double) * F * m.C * m.S * 2, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(eps, de_.p, sizeof(double) * F * m.nq, hipMemcpyDeviceToHost, h->stream));
    if (cost && weight) HIPCHK(hipMemcpyAsync(cost, dc_.p, sizeof(double) * F, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return CPE_OK;
}

cpe_status cpe_solve_host(cpe_handle* h, int32_t B, int32_t N, const double* q_init, const double* meas, const double* weight,
                          double* q, double* dq, double* ddq, double* positions, double* meas_err, cpe_stats* stats) {
    if (!h || !q_init || !meas || !weight || !q) return fail(CPE_BAD_ARG, "null argument");
    const size_t F = (size_t)B * N; const DevModel& m = h->hm;
    if (F == 0) return CPE_OK;
    HIPCHK(hipSetDevice(h->device));
    const size_t nm = F * m.C * m.L;
    DevBuf di, dm_, dw_, oq, odq, oddq, op, ome;
    HIPCHK(di.alloc(F * m.nq)); HIPCHK(dm_.alloc(nm * 2)); HIPCHK(dw_.alloc(nm)); HIPCHK(oq.alloc(F * m.nq));
    HIPCHK(odq.alloc(F * m.nq)); HIPCHK(oddq.alloc(F * m.nq)); HIPCHK(op.alloc(F * m.L * 3)); HIPCHK(ome.alloc(nm * 2));
    HIPCHK(hipMemcpyAsync(di.p, q_init, sizeof(double) * F * m.nq, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dm_.p, meas, sizeof(double) * nm * 2, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dw_.p, weight, sizeof(double) * nm, hipMemcpyHostToDevice, h->stream));
    cpe_status s = cpe_solve(h, B, N, di.p, dm_.p, dw_.p, oq.p, odq.p, oddq.p, op.p, ome.p, stats);
    if (s < 0) return s;
    HIPCHK(hipMemcpyAsync(q, oq.p, sizeof(double) * F * m.nq, hipMemcpyDeviceToHost, h->stream));
    if (dq) HIPCHK(hipMemcpyAsync(dq, odq.p, sizeof(double) * F * m.nq, hipMemcpyDeviceToHost, h->stream));
    if (ddq) HIPCHK(hipMemcpyAsync(ddq, oddq.p, sizeof(double) * F * m.nq, hipMemcpyDeviceToHost, h->stream));
    if (positions) HIPCHK(hipMemcpyAsync(positions, op.p, sizeof(double) * F * m.L * 3, hipMemcpyDeviceToHost, h->stream));
    if (meas_err) HIPCHK(hipMemcpyAsync(meas_err, ome.p, sizeof(double) * nm * 2, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return s;
}

}  // extern "C"
