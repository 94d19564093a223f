// cpe_device.h -- fp64 device math shared by the kernels (gfx950 / CDNA4, wave64).
// Reference formulas: SURVEY.md Appendix A; file:line citations at each function.
#pragma once
#include <hip/hip_runtime.h>

#include "cpe_model.h"

#define WAVE 64

// LDS written by some lanes of a wave and read by other lanes of the SAME wave: DS operations of one
// wave execute in order, so only the compiler has to be kept from reordering across this point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and store
// of the wave (s_waitcnt vmcnt(0)), which defeats register prefetching and makes each barrier pay the HBM write latency;
// use this one where waves exchange data through LDS alone.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, WAVE));
    return v;
}

// R = Rz(psi) Ry(theta) Rx(phi) (kind 0) or its derivative w.r.t. phi/theta/psi (kind 1/2/3), row-major.
// sc = {sin phi, cos phi, sin theta, cos theta, sin psi, cos psi}.   SURVEY A.2 (Link3D.Rb_I)
__device__ __forceinline__ void rot_kind(const double* sc, int kind, double* M) {
    const double sf = sc[0], cf = sc[1], st = sc[2], ct = sc[3], sp = sc[4], cp = sc[5];
    if (kind == 0) {
        M[0] = cp * ct; M[1] = sf * st * cp - sp * cf; M[2] = sf * sp + st * cf * cp;
        M[3] = sp * ct; M[4] = sf * sp * st + cf * cp; M[5] = sp * st * cf - sf * cp;
        M[6] = -st;     M[7] = sf * ct;                M[8] = cf * ct;
    } else if (kind == 1) {
        M[0] = 0.0; M[1] = cf * st * cp + sp * sf; M[2] = cf * sp - st * sf * cp;
        M[3] = 0.0; M[4] = cf * sp * st - sf * cp; M[5] = -sp * st * sf - cf * cp;
        M[6] = 0.0; M[7] = cf * ct;                M[8] = -sf * ct;
    } else if (kind == 2) {
        M[0] = -cp * st; M[1] = sf * ct * cp; M[2] = ct * cf * cp;
        M[3] = -sp * st; M[4] = sf * sp * ct; M[5] = sp * ct * cf;
        M[6] = -ct;      M[7] = -sf * st;     M[8] = -cf * st;
    } else {
        M[0] = -sp * ct; M[1] = -sf * st * sp - cp * cf; M[2] = sf * cp - st * cf * sp;
        M[3] = cp * ct;  M[4] = sf * cp * st - cf * sp;  M[5] = sf * sp + cp * st * cf;
        M[6] = 0.0;      M[7] = 0.0;                     M[8] = 0.0;
    }
}

// y axis (second column) of R from sin/cos
__device__ __forceinline__ void rot_ycol(const double* sc, double* a) {
    const double sf = sc[0], cf = sc[1], st = sc[2], ct = sc[3], sp = sc[4], cp = sc[5];
    a[0] = sf * st * cp - sp * cf; a[1] = sf * sp * st + cf * cp; a[2] = sf * ct;
}

// camera projection, residual-ready: uv and G = d(u,v)/dp (2x3 row-major).
// fisheye acinoset_misc.py:1663-1679, pinhole-radial :1682-1696.
__device__ __forceinline__ void project_point(const cpe_camera& c, double px, double py, double pz,
                                              double& u, double& v, double* G) {
    const double X0 = c.R[0] * px + c.R[1] * py + c.R[2] * pz + c.t[0];
    const double X1 = c.R[3] * px + c.R[4] * py + c.R[5] * pz + c.t[1];
    const double X2 = c.R[6] * px + c.R[7] * py + c.R[8] * pz + c.t[2];
    const double iz = 1.0 / X2;
    const double a = X0 * iz, b = X1 * iz;
    const double r2 = a * a + b * b;
    const double r = sqrt(r2);
    double g, dg;
    if (c.model == CPE_CAM_FISHEYE) {
        const double th = atan(r), t2 = th * th;
        const double poly = 1.0 + t2 * (c.D[0] + t2 * (c.D[1] + t2 * (c.D[2] + t2 * c.D[3])));
        const double dpoly = 1.0 + t2 * (3.0 * c.D[0] + t2 * (5.0 * c.D[1] + t2 * (7.0 * c.D[2] + t2 * 9.0 * c.D[3])));
        const double den = r + 1e-12, iden = 1.0 / den;
        g = th * poly * iden;
        dg = (dpoly / (1.0 + r2) - g) * iden;
    } else {
        g = 1.0 + r2 * (c.D[0] + r2 * (c.D[1] + r2 * c.D[2]));
        dg = r * (2.0 * c.D[0] + r2 * (4.0 * c.D[1] + r2 * 6.0 * c.D[2]));
    }
    u = c.fx * a * g + c.cx;
    v = c.fy * b * g + c.cy;
    if (G) {
        const double k = r > 0.0 ? dg / r : 0.0;
        const double xa = g + a * a * k, xb = a * b * k, yb = g + b * b * k;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double da = (c.R[j] - a * c.R[6 + j]) * iz;
            const double db = (c.R[3 + j] - b * c.R[6 + j]) * iz;
            G[j] = c.fx * (xa * da + xb * db);
            G[3 + j] = c.fy * (xb * da + yb * db);
        }
    }
}

// project_point for a camera staged in LDS as CAMW = 22 doubles in cpe_camera's layout (model word | fx fy cx cy | D[4] | R[9] | t[3] | mult).
// Same arithmetic in the same order; the camera is read in three instalments -- R, t for the camera-frame point; D for the distortion; R, fx, fy
// again for the 2x3 derivative -- each through a camera index that an empty `asm` ties to the previous instalment's result, so that the 22 parameters are never live together with the
// atan / sqrt / division temporaries (read up front from global memory they cost 46 VGPRs and the kernel its third wave per SIMD).
#define CAMW 22
static_assert(sizeof(cpe_camera) == CAMW * sizeof(double), "cpe_camera layout");
__device__ __forceinline__ void project_point_staged(const double* scam, int c, double px, double py, double pz, double& u, double& v, double* G) {
    const double* cam = scam + CAMW * c;
    const double* cR = cam + 9; const double* ct = cam + 18;
    const double X0 = cR[0] * px + cR[1] * py + cR[2] * pz + ct[0];
    const double X1 = cR[3] * px + cR[4] * py + cR[5] * pz + ct[1];
    double X2 = cR[6] * px + cR[7] * py + cR[8] * pz + ct[2];
    int c2 = c;
    asm volatile("" : "+v"(c2), "+v"(X2));                  // the camera index of the second instalment exists only once X2 does
    const double* cD = scam + CAMW * c2 + 5;
    const double iz = 1.0 / X2;
    const double a = X0 * iz, b = X1 * iz;
    const double r2 = a * a + b * b;
    const double r = sqrt(r2);
    double g, dg;
    if (reinterpret_cast<const int*>(cD - 5)[0] == CPE_CAM_FISHEYE) {
        const double th = atan(r), t2 = th * th;
        const double poly = 1.0 + t2 * (cD[0] + t2 * (cD[1] + t2 * (cD[2] + t2 * cD[3])));
        const double dpoly = 1.0 + t2 * (3.0 * cD[0] + t2 * (5.0 * cD[1] + t2 * (7.0 * cD[2] + t2 * 9.0 * cD[3])));
        const double den = r + 1e-12, iden = 1.0 / den;
        g = th * poly * iden;
        dg = (dpoly / (1.0 + r2) - g) * iden;
    } else {
        g = 1.0 + r2 * (cD[0] + r2 * (cD[1] + r2 * cD[2]));
        dg = r * (2.0 * cD[0] + r2 * (4.0 * cD[1] + r2 * 6.0 * cD[2]));
    }
    int c3 = c2;
    asm volatile("" : "+v"(c3), "+v"(dg));                  // ... and that of the third once the distortion is done
    const double* cam3 = scam + CAMW * c3; const double* cR3 = cam3 + 9;
    const double fx = cam3[1], fy = cam3[2];
    u = fx * a * g + cam3[3];
    v = fy * b * g + cam3[4];
    const double k = r > 0.0 ? dg / r : 0.0;
    const double xa = g + a * a * k, xb = a * b * k, yb = g + b * b * k;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const double da = (cR3[j] - a * cR3[6 + j]) * iz;
        const double db = (cR3[3 + j] - b * cR3[6 + j]) * iz;
        G[j] = fx * (xa * da + xb * db);
        G[3 + j] = fy * (xb * da + yb * db);
    }
}

// redescending loss rho(s) (acinoset_misc.py:2001-2015): value, d rho/ds, PSD curvature weight.
// curvature mode 0: max(rho'', rho'(|s|)/|s|, 0); mode 1: max(rho'', 0)   (DESIGN.md "Solver")
struct LossOut { double rho, d1, cw; };
__device__ __forceinline__ LossOut robust_loss(double s, double a, double b, double c, int mode, bool want_deriv) {
    const double e = fabs(s);
    const double sa = 1.0 / (1.0 + exp(a - e)), sb = 1.0 / (1.0 + exp(b - e)), sc = 1.0 / (1.0 + exp(c - e));
    const double lin = a * e - 0.5 * a * a;
    const double cb = c - b, w = (c - e) / cb;
    const double kq = a * b - 0.5 * a * a + 0.5 * a * cb * (1.0 - w * w);
    const double K = a * b - 0.5 * a * a + 0.5 * a * cb;
    LossOut o;
    o.rho = 0.5 * (1.0 - sa) * e * e + (sa - sb) * lin + (sb - sc) * kq + sc * K;
    o.d1 = 0.0; o.cw = 0.0;
    if (want_deriv) {
        const double sa1 = sa * (1.0 - sa), sb1 = sb * (1.0 - sb), sc1 = sc * (1.0 - sc);
        const double sa2 = sa1 * (1.0 - 2.0 * sa), sb2 = sb1 * (1.0 - 2.0 * sb), sc2 = sc1 * (1.0 - 2.0 * sc);
        const double k1 = a * (c - e) / cb, k2 = -a / cb;
        const double d1 = -0.5 * sa1 * e * e + (1.0 - sa) * e + (sa1 - sb1) * lin + (sa - sb) * a
                        + (sb1 - sc1) * kq + (sb - sc) * k1 + sc1 * K;
        const double d2 = -0.5 * sa2 * e * e - 2.0 * sa1 * e + (1.0 - sa) + (sa2 - sb2) * lin + 2.0 * (sa1 - sb1) * a
                        + (sb2 - sc2) * kq + 2.0 * (sb1 - sc1) * k1 + (sb - sc) * k2 + sc2 * K;
        o.d1 = s > 0.0 ? d1 : (s < 0.0 ? -d1 : 0.0);
        double cw = d2 > 0.0 ? d2 : 0.0;
        if (mode == 0 && e > 1e-12 && d1 > 0.0) cw = fmax(cw, d1 / e);
        o.cw = cw;
    }
    return o;
}
