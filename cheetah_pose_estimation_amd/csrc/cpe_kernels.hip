// cpe_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X): one 64-lane wavefront per frame.
//
//   k_resjac        metric 1: reprojection residual + sparse Jacobian + acceleration slack per frame
//   k_fk            marker positions + centre of mass
//   k_project       closed-form solve of the joint equalities for the dependent angles
//   k_frame_normal  solver: per-frame cost, reduced gradient g[28] and PSD block B[28x28]
//   k_lm_step       solver: one Levenberg-Marquardt step per sequence (block-banded Cholesky over time)
//   k_finalize      solver outputs (q, dq, ddq, positions, meas_err, cost terms)
//
// Data layout in HBM (all fp64, frame-major, so a wave's loads/stores of one frame are contiguous):
//   q[B][N][nq]  meas[B][N][C][L][2]  weight[B][N][C][L]  r[B][N][C][L][2]  J[B][N][C][S][2]  eps[B][N][nq]
// No MFMA: nothing here is a dense contraction larger than 28x28.
#include <hip/hip_runtime.h>

#include "cpe_device.h"

// --------------------------------------------------------------------------------------------------
// shared wave-level building blocks.  LDS arrays are private to the wave (one wave per workgroup).

// sincos of all link angles of the frame held in sq -> ssc[6*link + 2*ang + {0,1}]
__device__ __forceinline__ void wave_sincos(const DevModel* __restrict__ M, const double* sq, double* ssc, int lane) {
    if (lane < 3 * M->nl) {
        double s, c;
        sincos(sq[3 + lane], &s, &c);
        ssc[2 * lane] = s; ssc[2 * lane + 1] = c;
    }
}

// R and dR/d(phi,theta,psi) of every link -> sR[9*(4*link + kind)]
__device__ __forceinline__ void wave_rotations(const DevModel* __restrict__ M, const double* ssc, double* sR, int lane) {
    for (int t = lane; t < 4 * M->nl; t += WAVE) rot_kind(ssc + 6 * (t >> 2), t & 3, sR + 9 * t);
}

// marker positions p_l = x_base + sum_k R_k v_lk  (acinoset_misc.py:1581-1659)
__device__ __forceinline__ void wave_markers(const DevModel* __restrict__ M, const double* sq, const double* sR,
                                             double* spos, int lane) {
    if (lane < M->L) {
        double p0 = sq[0], p1 = sq[1], p2 = sq[2];
        const int n = M->chain_len[lane];
        for (int k = 0; k < n; k++) {
            const double* R = sR + 36 * M->chain_link[lane][k];
            const double v0 = M->chain_vec[lane][k][0], v1 = M->chain_vec[lane][k][1], v2 = M->chain_vec[lane][k][2];
            p0 += R[0] * v0 + R[1] * v1 + R[2] * v2;
            p1 += R[3] * v0 + R[4] * v1 + R[5] * v2;
            p2 += R[6] * v0 + R[7] * v1 + R[8] * v2;
        }
        spos[3 * lane] = p0; spos[3 * lane + 1] = p1; spos[3 * lane + 2] = p2;
    }
}

// d p_marker / d q_dof for Jacobian slot s
__device__ __forceinline__ void slot_dp(const DevModel* __restrict__ M, const double* sR, int s, double& d0, double& d1, double& d2) {
    const int cpos = M->slot_cpos[s];
    if (cpos < 0) {
        const int ax = M->slot_dof[s];
        d0 = ax == 0 ? 1.0 : 0.0; d1 = ax == 1 ? 1.0 : 0.0; d2 = ax == 2 ? 1.0 : 0.0;
    } else {
        const int l = M->slot_marker[s];
        const double* D = sR + 9 * (4 * M->chain_link[l][cpos] + 1 + M->slot_ang[s]);
        const double v0 = M->chain_vec[l][cpos][0], v1 = M->chain_vec[l][cpos][1], v2 = M->chain_vec[l][cpos][2];
        d0 = D[0] * v0 + D[1] * v1 + D[2] * v2;
        d1 = D[3] * v0 + D[4] * v1 + D[5] * v2;
        d2 = D[6] * v0 + D[7] * v1 + D[8] * v2;
    }
}

// closed-form solution of the joint equalities for the dependent angles (branch
// child.y = +parent.y reached from the reference's initial guess, acinoset_opt.py:574-583).
// Updates sq and ssc in place; returns non-zero in every lane if some revolute child sits in the gimbal
// band |cos(theta)| < |a_z| where the equalities have no solution.
__device__ __forceinline__ int wave_project_joints(const DevModel* __restrict__ M, double* sq, double* ssc, int lane, bool hooke_only = false) {
    int clamped = 0;
    for (int level = 0; level < 2; level++) {
        if (lane < M->nj && !(hooke_only && M->joint_kind[lane] == CPE_JOINT_REVOLUTE_Y)) {
            const int kind = M->joint_kind[lane], p = M->joint_parent[lane], c = M->joint_child[lane];
            const bool lvl1 = kind == CPE_JOINT_HOOKE_YZ && M->dep_of_q[3 + 3 * p] >= 0;   // parent phi is itself dependent
            if ((level == 1) == lvl1) {
                double a[3];
                const double st = ssc[6 * c + 2], ct = ssc[6 * c + 3];
                if (kind == CPE_JOINT_REVOLUTE_Y) {
                    rot_ycol(ssc + 6 * M->joint_body[lane], a);
                    double sphi = a[2] / ct;
                    if (sphi > 1.0) { sphi = 1.0; clamped = 1; }
                    if (sphi < -1.0) { sphi = -1.0; clamped = 1; }
                    const double cphi = sqrt(fmax(0.0, 1.0 - sphi * sphi));
                    double psi = atan2(a[1], a[0]) - atan2(cphi, sphi * st);
                    const double ref = sq[3 + 3 * p + 2];
                    psi += 6.283185307179586476925286766559 * rint((ref - psi) / 6.283185307179586476925286766559);
                    sq[3 + 3 * c] = asin(sphi); sq[3 + 3 * c + 2] = psi;
                    ssc[6 * c] = sphi; ssc[6 * c + 1] = cphi;
                    double s, co; sincos(psi, &s, &co);
                    ssc[6 * c + 4] = s; ssc[6 * c + 5] = co;
                } else {
                    rot_ycol(ssc + 6 * p, a);
                    const double sp = ssc[6 * c + 4], cp = ssc[6 * c + 5];
                    const double num = a[0] * st * cp + a[1] * st * sp + a[2] * ct;
                    const double den = a[1] * cp - a[0] * sp;
                    const double hyp = sqrt(num * num + den * den);
                    sq[3 + 3 * c] = atan2(num, den);
                    ssc[6 * c] = num / hyp; ssc[6 * c + 1] = den / hyp;
                }
            }
        }
        wave_lds_sync();
    }
    return __any(clamped);
}

// --------------------------------------------------------------------------------------------------
// metric 1 kernel: residual + sparse Jacobian + acceleration slack.
//
// Persistent workgroups of NW waves; every wave owns one frame at a time and walks the frame list with a
// grid stride.  Read-only model tables (cameras, Jacobian-slot table, marker-chain table) are staged ONCE
// per workgroup in LDS, so the per-frame code has no dependent global loads; the next frame's q / meas /
// neighbour-q are prefetched into registers while the current frame computes and stores.
// LDS (doubles): shared  cam[23C] | ident[10] | slot[4S] | chain[4*L*MAXCHAIN] | clen[(L+1)/2]
//                per wave A[max(6CL, nq+6nl+36nl)] | pos[3L(+1)]   with A = {q, sin/cos, R & dR} later overlaid by G
// Tables are structure-of-arrays so that consecutive lanes touch consecutive LDS words (no bank conflicts):
//   slot:  v0[S] v1[S] v2[S] (double) | doff[S] marker[S] (int32; doff = offset of the 3x3 dR in the wave's R block, <0: identity)
//   chain: v0[K][L] v1[K][L] v2[K][L] (double) | roff[K][L] (int32), K = CPE_MAX_CHAIN, marker index fastest
#define RJ_MAXPASS 4

__host__ __device__ inline int rj_shared_doubles(int C, int L, int S) {
    return 23 * C + 10 + 3 * S + (2 * S + 1) / 2 + 3 * L * CPE_MAX_CHAIN + (L * CPE_MAX_CHAIN + 1) / 2 + (L + 1) / 2;
}
__host__ __device__ inline int rj_wave_doubles(int C, int L, int nq, int nl) {
    const int a = 6 * C * L, b = nq + 6 * nl + 36 * nl;
    return ((a > b ? a : b) + 3 * L + 3) & ~1;
}

template <bool WANT_COST, int NW, int NPASS, int OCC>
__global__ __launch_bounds__(WAVE * NW, (OCC * NW) / 4) void k_resjac(const DevModel* __restrict__ M, int N, long F,
                                                      const double* __restrict__ q, const double* __restrict__ meas,
                                                      const double* __restrict__ weight, double* __restrict__ r,
                                                      double* __restrict__ J, double* __restrict__ eps,
                                                      double* __restrict__ cost) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = M->nq, nl = M->nl, L = M->L, C = M->C, S = M->S, CL = C * L;
    double* scam = smem;
    double* sident = scam + 23 * C;
    double* slv = sident + 10;                                              // slot vectors [3][S]
    int* sli = reinterpret_cast<int*>(slv + 3 * S);                         // doff[S], marker[S]
    double* chv = slv + 3 * S + (2 * S + 1) / 2;                            // chain vectors [3][K][L]
    int* chr = reinterpret_cast<int*>(chv + 3 * L * CPE_MAX_CHAIN);         // roff[K][L]
    int* sclen = reinterpret_cast<int*>(chv + 3 * L * CPE_MAX_CHAIN + (L * CPE_MAX_CHAIN + 1) / 2);
    double* wbase = smem + ((rj_shared_doubles(C, L, S) + 1) & ~1);
    const int nA = rj_wave_doubles(C, L, nq, nl) - ((3 * L + 3) & ~1) + 0;
    double* wv = wbase + wave * rj_wave_doubles(C, L, nq, nl);
    double* sq = wv;
    double* ssc = sq + nq;
    double* sR = ssc + 6 * nl;
    double* sG = wv;                       // overlays q / sincos / R once they are dead
    double* spos = wv + nA;

    // ---- stage the model tables (once per workgroup)
    for (int t = tid; t < 23 * C; t += WAVE * NW) scam[t] = reinterpret_cast<const double*>(M->cam)[t];
    if (tid < 9) sident[tid] = (tid == 0 || tid == 4 || tid == 8) ? 1.0 : 0.0;
    for (int s = tid; s < S; s += WAVE * NW) {
        const int l = M->slot_marker[s], cpos = M->slot_cpos[s];
        sli[S + s] = l;
        if (cpos < 0) {
            const int ax = cpos == -2 ? 3 : M->slot_dof[s];            // -2: alignment slot, structurally zero
            slv[s] = ax == 0 ? 1.0 : 0.0; slv[S + s] = ax == 1 ? 1.0 : 0.0; slv[2 * S + s] = ax == 2 ? 1.0 : 0.0; sli[s] = -1;
        } else {
            slv[s] = M->chain_vec[l][cpos][0]; slv[S + s] = M->chain_vec[l][cpos][1]; slv[2 * S + s] = M->chain_vec[l][cpos][2];
            sli[s] = 9 * (4 * M->chain_link[l][cpos] + 1 + M->slot_ang[s]);
        }
    }
    for (int t = tid; t < L * CPE_MAX_CHAIN; t += WAVE * NW) {
        const int k = t / L, l = t - k * L;
        const bool on = k < M->chain_len[l];
        chv[t] = on ? M->chain_vec[l][k][0] : 0.0;
        chv[L * CPE_MAX_CHAIN + t] = on ? M->chain_vec[l][k][1] : 0.0;
        chv[2 * L * CPE_MAX_CHAIN + t] = on ? M->chain_vec[l][k][2] : 0.0;
        chr[t] = on ? 36 * M->chain_link[l][k] : 0;
    }
    for (int t = tid; t < L; t += WAVE * NW) sclen[t] = M->chain_len[t];
    const double ih2 = M->ih2, la = M->loss_a, lb = M->loss_b, lc = M->loss_c;
    __syncthreads();

    const cpe_camera* cams = reinterpret_cast<const cpe_camera*>(scam);
    const long wstride = (long)gridDim.x * NW;
    long f = (long)blockIdx.x * NW + wave;

    // One drain point per frame.  vmcnt counts loads and stores together and the two kinds may retire out of order with
    // respect to each other, so with stores in flight the only safe wait for a load is vmcnt(0) -- which drains every
    // outstanding store of the wave.  The first version waited for its measurements inside each projection pass and for
    // the next q at the frame end: 4 drains per frame, compute and the 26 KB of J stores never overlapped (compute-only
    // 1.86 ms + store-only 2.0 ms ~ 3.1 ms measured).  Now a frame issues all its loads at the top, computes everything
    // (projections and G included) without touching them, waits ONCE, and only then issues its 34 stores in one burst;
    // the burst drains while the next frame computes.
    const double2* meas2 = reinterpret_cast<const double2*>(meas);
    double qreg = 0.0;
    if (f < F && lane < nq) qreg = q[f * nq + lane];
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0) before the loop: its header then carries no pending load on either edge

    while (f < F) {
        // Two waves share a SIMD: the one in its dependent chain (rotations, chain walk, projections) goes first at issue, the one in its store
        // burst (independent G reads, products and stores) fills the gaps -- measured -3.0 % against equal priorities, the reverse +0.7 % (60 launches each, interleaved in one process)
        __builtin_amdgcn_s_setprio(2);
        if (lane < nq) sq[lane] = qreg;
        const long fn = f + wstride;
        const bool has_prev = (int)(f % N) >= 3;
        double qn = 0.0, qp1 = 0.0, qp2 = 0.0, qp3 = 0.0;
        double2 mz[NPASS];
        double wz[NPASS];
        if (lane < nq) {
            const double* qf = q + f * nq + lane;
            if (fn < F) qn = q[fn * nq + lane];
            if (has_prev) { qp1 = qf[-nq]; qp2 = qf[-2 * nq]; qp3 = qf[-3 * nq]; }
        }
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            const int t = lane + WAVE * p;
            mz[p] = make_double2(0.0, 0.0); wz[p] = 0.0;
            if (t < CL) { mz[p] = meas2[f * CL + t]; if (WANT_COST) wz[p] = weight[f * CL + t]; }
        }
        wave_lds_sync();
        wave_sincos(M, sq, ssc, lane);
        wave_lds_sync();
        for (int t = lane; t < 4 * nl; t += WAVE) rot_kind(ssc + 6 * (t >> 2), t & 3, sR + 9 * t);
        wave_lds_sync();
        // marker positions from the LDS chain table.  Two lanes per marker (2 L <= 64): lane l takes the even chain elements
        // (and the base position), lane L + l the odd ones; the halves meet through one cross-lane read.  Halves the depth of
        // the dependent table-index -> rotation reads.
        {
            const int half = lane >= L ? 1 : 0, l = lane - half * L;
            double p0 = 0.0, p1 = 0.0, p2 = 0.0;
            if (lane < 2 * L) {
                if (!half) { p0 = sq[0]; p1 = sq[1]; p2 = sq[2]; }
                const int n = sclen[l];
                const int KL = L * CPE_MAX_CHAIN;
                for (int k = half; k < n; k += 2) {
                    const int t = k * L + l;
                    const double v0 = chv[t], v1 = chv[KL + t], v2 = chv[2 * KL + t];
                    const double* R = sR + chr[t];
                    p0 += R[0] * v0 + R[1] * v1 + R[2] * v2;
                    p1 += R[3] * v0 + R[4] * v1 + R[5] * v2;
                    p2 += R[6] * v0 + R[7] * v1 + R[8] * v2;
                }
            }
            const int partner = (lane + L) & (WAVE - 1);
            const double o0 = __shfl(p0, partner, WAVE), o1 = __shfl(p1, partner, WAVE), o2 = __shfl(p2, partner, WAVE);
            if (lane < L) { spos[3 * lane] = p0 + o0; spos[3 * lane + 1] = p1 + o1; spos[3 * lane + 2] = p2 + o2; }
        }
        // d p / d q of my Jacobian slots (slot s = lane + 64 i) stay in registers
        double dp0[5], dp1[5], dp2[5];
        int mk[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int s = lane + WAVE * i;
            dp0[i] = dp1[i] = dp2[i] = 0.0; mk[i] = 0;
            if (s < S) {
                const int doff = sli[s];
                const double v0 = slv[s], v1 = slv[S + s], v2 = slv[2 * S + s];
                const double* D = doff < 0 ? sident : sR + doff;
                mk[i] = sli[S + s];
                dp0[i] = D[0] * v0 + D[1] * v1 + D[2] * v2;
                dp1[i] = D[3] * v0 + D[4] * v1 + D[5] * v2;
                dp2[i] = D[6] * v0 + D[7] * v1 + D[8] * v2;
            }
        }
        wave_lds_sync();        // R, sincos, q are dead from here: G may overlay them

        // project every (camera, marker) pair: (u, v) stay in registers, d(u,v)/dp goes to LDS; nothing is stored yet
        const long pair0 = f * CL;
        double2* Jf = reinterpret_cast<double2*>(J) + f * (long)(C * S);
        double fc = 0.0;
        double2 uv[NPASS];
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            __builtin_amdgcn_sched_barrier(0);      // keep the passes sequential: interleaving them costs ~40 VGPRs
            const int t = lane + WAVE * p;
            uv[p] = make_double2(0.0, 0.0);
            if (t < CL) {
                const int c = t / L, l = t - c * L;
                double u, v, G[6];
                project_point(cams[c], spos[3 * l], spos[3 * l + 1], spos[3 * l + 2], u, v, G);      // (the staged form, 210 instead of 225 VGPRs, measured 0.8 % slower here: no occupancy step in reach)
                uv[p] = make_double2(u, v);
                double2* Gd = reinterpret_cast<double2*>(sG + 6 * t);
                Gd[0] = make_double2(G[0], G[1]); Gd[1] = make_double2(G[2], G[3]); Gd[2] = make_double2(G[4], G[5]);
            }
        }
        wave_lds_sync();
        // the frame's only wait on vector memory: this frame's loads have arrived, the previous frame's stores have drained
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        // the loaded values become usable HERE: without this the compiler hoists 3*qp1, 3*qp2 to the top of the frame and
        // waits for them there (which drains the previous frame's stores at once)
        asm volatile("" : "+v"(qp1), "+v"(qp2), "+v"(qp3), "+v"(qn));
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            const int t = lane + WAVE * p;
            if (t < CL) {
                const double e0 = uv[p].x - mz[p].x, e1 = uv[p].y - mz[p].y;
                reinterpret_cast<double2*>(r)[pair0 + t] = make_double2(e0, e1);
                if (WANT_COST) {
                    const double w = cams[t / L].mult * wz[p];
                    fc += robust_loss(w * e0, la, lb, lc, 0, false).rho + robust_loss(w * e1, la, lb, lc, 0, false).rho;
                }
            }
        }
        // J[c][s][0..1] = G_{c,marker(s)} . dp_s : 16-byte streaming stores, consecutive lanes -> consecutive slots
        for (int c = 0; c < C; c++) {
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int s = lane + WAVE * i;
                if (s < S) {
                    const double2* G = reinterpret_cast<const double2*>(sG + 6 * (c * L + mk[i]));
                    const double2 g0 = G[0], g1 = G[1], g2 = G[2];
                    typedef double v2d __attribute__((ext_vector_type(2)));
                    v2d val;
                    val.x = g0.x * dp0[i] + g0.y * dp1[i] + g1.x * dp2[i];
                    val.y = g1.y * dp0[i] + g2.x * dp1[i] + g2.y * dp2[i];
#ifdef CPE_RJ_NOSTORE       // diagnostic build only (profiles/r01_resjac_ablation.md): J computed, not stored -- the compute floor of the box
                    if (val.x == 1.2345e300) Jf[c * S + s] = make_double2(val.x, val.y);
#else
                    __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(&Jf[c * S + s]));
#endif
                }
            }
        }
        if (WANT_COST) {
            fc = wave_sum(fc);
            if (lane == 0) cost[f] = fc;
        }
        // acceleration slack of the constant-acceleration model (free dq0/ddq0 => 0 for n < 3)
        if (lane < nq) {
            double e = 0.0;
            if (has_prev) e = (qreg - 3.0 * qp1 + 3.0 * qp2 - qp3) * ih2;
            eps[f * nq + lane] = e;
        }
        qreg = qn;
        f = fn;
        wave_lds_sync();        // G region is rewritten as q / sincos / R by the next frame
    }
}

// --------------------------------------------------------------------------------------------------
// forward kinematics: positions[F][L][3], com[F][3] (acinoset_misc.py:1581-1659, :722-742)
// dynamic LDS: q[nq] | sc[6 nl] | R[36 nl] | pos[3 L]
__global__ __launch_bounds__(WAVE) void k_fk(const DevModel* __restrict__ M, const double* __restrict__ q,
                                             double* __restrict__ positions, double* __restrict__ com) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const int nq = M->nq, nl = M->nl, L = M->L;
    double* sq = smem;
    double* ssc = sq + nq;
    double* sR = ssc + 6 * nl;
    double* spos = sR + 36 * nl;
    const size_t f = blockIdx.x;
    if (lane < nq) sq[lane] = q[f * nq + lane];
    wave_lds_sync();
    wave_sincos(M, sq, ssc, lane);
    wave_lds_sync();
    for (int t = lane; t < nl; t += WAVE) rot_kind(ssc + 6 * t, 0, sR + 36 * t);
    wave_lds_sync();
    wave_markers(M, sq, sR, spos, lane);
    wave_lds_sync();
    for (int t = lane; t < 3 * L; t += WAVE) positions[f * (size_t)(3 * L) + t] = spos[t];
    if (com) {
        // link origins by walking up the chain (depth <= CPE_MAX_CHAIN), one lane per link
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
        if (lane < nl) {
            const double* R = sR + 36 * lane;
            double p0 = R[0] * M->com[lane][0] + R[1] * M->com[lane][1] + R[2] * M->com[lane][2];
            double p1 = R[3] * M->com[lane][0] + R[4] * M->com[lane][1] + R[5] * M->com[lane][2];
            double p2 = R[6] * M->com[lane][0] + R[7] * M->com[lane][1] + R[8] * M->com[lane][2];
            int k = lane;
            while (M->parent[k] >= 0) {
                const int p = M->parent[k];
                const double* Rp = sR + 36 * p;
                p0 += Rp[0] * M->attach[k][0] + Rp[1] * M->attach[k][1] + Rp[2] * M->attach[k][2];
                p1 += Rp[3] * M->attach[k][0] + Rp[4] * M->attach[k][1] + Rp[5] * M->attach[k][2];
                p2 += Rp[6] * M->attach[k][0] + Rp[7] * M->attach[k][1] + Rp[8] * M->attach[k][2];
                k = p;
            }
            const double m = M->mass[lane];
            c0 = m * (p0 + sq[0]); c1 = m * (p1 + sq[1]); c2 = m * (p2 + sq[2]);
        }
        c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
        if (lane == 0) {
            com[3 * f] = c0 * M->inv_total_mass; com[3 * f + 1] = c1 * M->inv_total_mass; com[3 * f + 2] = c2 * M->inv_total_mass;
        }
    }
}

// --------------------------------------------------------------------------------------------------
// marker velocities v_l = (d p_l / d q) dq, the analytic foot velocity of the contact heuristic
// (`foot.Pb_I_vel` lambdified in acinoset_misc.py:347-360).  One lane per marker walks the marker's Jacobian slots.
// dynamic LDS: q[nq] | dq[nq] | sc[6 nl] | R[36 nl]
__global__ __launch_bounds__(WAVE) void k_marker_vel(const DevModel* __restrict__ M, const double* __restrict__ q,
                                                     const double* __restrict__ dq, double* __restrict__ vel) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const int nq = M->nq, nl = M->nl, L = M->L;
    double* sq = smem;
    double* sdq = sq + nq;
    double* ssc = sdq + nq;
    double* sR = ssc + 6 * nl;
    const size_t f = blockIdx.x;
    if (lane < nq) { sq[lane] = q[f * nq + lane]; sdq[lane] = dq[f * nq + lane]; }
    wave_lds_sync();
    wave_sincos(M, sq, ssc, lane);
    wave_lds_sync();
    wave_rotations(M, ssc, sR, lane);
    wave_lds_sync();
    if (lane < L) {
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
        for (int s = M->slot_off[lane]; s < M->slot_off[lane + 1]; s++) {
            double d0, d1, d2;
            slot_dp(M, sR, s, d0, d1, d2);
            const double w = sdq[M->slot_dof[s]];
            v0 += d0 * w; v1 += d1 * w; v2 += d2 * w;
        }
        double* o = vel + (f * (size_t)L + lane) * 3;
        o[0] = v0; o[1] = v1; o[2] = v2;
    }
}

// --------------------------------------------------------------------------------------------------
// reprojection of stored 3D marker positions into every camera, for the cam*_fte writers
// (acinoset_misc.py:1339-1407).  One lane per (camera, marker) pair; no LDS.
__global__ __launch_bounds__(WAVE) void k_reproject(const DevModel* __restrict__ M, const double* __restrict__ positions,
                                                    double* __restrict__ uv) {
    const int L = M->L, CL = M->C * L;
    const size_t f = blockIdx.x;
    const double* P = positions + f * (size_t)(3 * L);
    for (int t = threadIdx.x; t < CL; t += WAVE) {
        const int c = t / L, l = t - c * L;
        double u, v, G[6];
        project_point(M->cam[c], P[3 * l], P[3 * l + 1], P[3 * l + 2], u, v, G);
        reinterpret_cast<double2*>(uv)[f * (size_t)CL + t] = make_double2(u, v);
    }
}

// --------------------------------------------------------------------------------------------------
// initial-guess ingestion: undistort two detections and triangulate them linearly (triangulate_points[_fisheye],
// acinoset_misc.py:1432-1453: cv[.fisheye].undistortPoints + cv.triangulatePoints), one lane per point pair.
// cam_b < 0: back-projection of the first detection to `depth` along its ray (the monocular rule of the initial
// trajectory estimate).  Pure register code.
__device__ __forceinline__ void undistort_pixel(const cpe_camera& c, double u, double v, double& x, double& y) {
    const double x0 = (u - c.cx) / c.fx, y0 = (v - c.cy) / c.fy;
    if (c.model == CPE_CAM_FISHEYE) {
        // theta_d = theta (1 + D0 theta^2 + .. + D3 theta^8) solved for theta by Newton, then (x, y) *= tan(theta) / theta_d
        const double rd = sqrt(x0 * x0 + y0 * y0);
        double th = rd;
        for (int it = 0; it < 20; it++) {
            const double t2 = th * th;
            const double f = th * (1.0 + t2 * (c.D[0] + t2 * (c.D[1] + t2 * (c.D[2] + t2 * c.D[3])))) - rd;
            const double df = 1.0 + t2 * (3.0 * c.D[0] + t2 * (5.0 * c.D[1] + t2 * (7.0 * c.D[2] + t2 * 9.0 * c.D[3])));
            th -= f / df;
        }
        const double sc = rd > 1e-12 ? tan(th) / rd : 1.0;
        x = x0 * sc; y = y0 * sc;
    } else {
        // radial polynomial inverted by the fixed-point iteration of cv.undistortPoints
        x = x0; y = y0;
        for (int it = 0; it < 20; it++) {
            const double r2 = x * x + y * y;
            const double g = 1.0 + r2 * (c.D[0] + r2 * (c.D[1] + r2 * c.D[2]));
            x = x0 / g; y = y0 / g;
        }
    }
}

__global__ __launch_bounds__(256) void k_triangulate(const DevModel* __restrict__ M, int n, const int* __restrict__ cam_a,
                                                     const int* __restrict__ cam_b, const double* __restrict__ uv_a,
                                                     const double* __restrict__ uv_b, double depth, double* __restrict__ xyz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cpe_camera& ca = M->cam[cam_a[i]];
    double xa, ya;
    undistort_pixel(ca, uv_a[2 * i], uv_a[2 * i + 1], xa, ya);
    double X[3];
    if (cam_b[i] < 0) {
        // X = R^T (depth (x, y, 1) - t)
        const double w[3] = {depth * xa - ca.t[0], depth * ya - ca.t[1], depth - ca.t[2]};
        for (int k = 0; k < 3; k++) X[k] = ca.R[k] * w[0] + ca.R[3 + k] * w[1] + ca.R[6 + k] * w[2];
    } else {
        const cpe_camera& cb = M->cam[cam_b[i]];
        double xb, yb;
        undistort_pixel(cb, uv_b[2 * i], uv_b[2 * i + 1], xb, yb);
        // rows of A: x P[2] - P[0], y P[2] - P[1] for both views, P = [R | t]; the solution is the right singular vector of
        // the smallest singular value = eigenvector of the smallest eigenvalue of A^T A (cyclic Jacobi, 4 x 4)
        double A[4][4];
        for (int k = 0; k < 4; k++) {
            const double a0 = k < 3 ? ca.R[k] : ca.t[0], a1 = k < 3 ? ca.R[3 + k] : ca.t[1], a2 = k < 3 ? ca.R[6 + k] : ca.t[2];
            const double b0 = k < 3 ? cb.R[k] : cb.t[0], b1 = k < 3 ? cb.R[3 + k] : cb.t[1], b2 = k < 3 ? cb.R[6 + k] : cb.t[2];
            A[0][k] = xa * a2 - a0; A[1][k] = ya * a2 - a1; A[2][k] = xb * b2 - b0; A[3][k] = yb * b2 - b1;
        }
        double S[4][4], V[4][4];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                S[r][c] = A[0][r] * A[0][c] + A[1][r] * A[1][c] + A[2][r] * A[2][c] + A[3][r] * A[3][c];
                V[r][c] = r == c ? 1.0 : 0.0;
            }
        for (int sweep = 0; sweep < 12; sweep++)
#pragma unroll
            for (int p = 0; p < 3; p++)
#pragma unroll
                for (int q = p + 1; q < 4; q++) {
                    const double apq = S[p][q];
                    if (fabs(apq) < 1e-300) continue;
                    const double tau = (S[q][q] - S[p][p]) / (2.0 * apq);
                    const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
#pragma unroll
                    for (int k = 0; k < 4; k++) { const double skp = S[k][p], skq = S[k][q]; S[k][p] = cs * skp - sn * skq; S[k][q] = sn * skp + cs * skq; }
#pragma unroll
                    for (int k = 0; k < 4; k++) { const double spk = S[p][k], sqk = S[q][k]; S[p][k] = cs * spk - sn * sqk; S[q][k] = sn * spk + cs * sqk; }
#pragma unroll
                    for (int k = 0; k < 4; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq; }
                }
        int m = 0;
#pragma unroll
        for (int k = 1; k < 4; k++) if (S[k][k] < S[m][m]) m = k;
        double h[4];
#pragma unroll
        for (int k = 0; k < 4; k++) h[k] = m == 0 ? V[k][0] : m == 1 ? V[k][1] : m == 2 ? V[k][2] : V[k][3];
        for (int k = 0; k < 3; k++) X[k] = h[k] / h[3];
    }
    xyz[3 * i] = X[0]; xyz[3 * i + 1] = X[1]; xyz[3 * i + 2] = X[2];
}

// --------------------------------------------------------------------------------------------------
// DeepLabCut table of one camera -> that camera's slice of meas[N][C][L][2] and weight[N][C][L]
// (init_measurements / init_meas_weights, acinoset_misc.py:211-256).  One lane per (frame, marker): a strided gather.
__global__ __launch_bounds__(256) void k_tensorise_dlc(int N, int L, int n_slots, int slot, const double* __restrict__ table, int rows,
                                                       int parts, int first_row, const int* __restrict__ part_of_marker,
                                                       const double* __restrict__ inv_sigma, double thresh,
                                                       double* __restrict__ meas, double* __restrict__ weight) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * L) return;
    const int n = (int)(i / L), l = (int)(i - (long)n * L);
    const long row = (long)first_row + n;
    double x = 0.0, y = 0.0, w = 0.0;
    const int part = part_of_marker[l];
    if (row >= 0 && row < rows && part >= 0 && part < parts) {
        const double* v = table + row * (long)(3 * parts) + 3 * part;
        const double vx = v[0], vy = v[1], lik = v[2];
        const bool finite = vx == vx && vy == vy && fabs(vx) != INFINITY && fabs(vy) != INFINITY;
        if (finite) { x = vx; y = vy; if (lik > thresh) w = inv_sigma[l]; }
    }
    const long o = ((long)n * n_slots + slot) * L + l;
    meas[2 * o] = x; meas[2 * o + 1] = y; weight[o] = w;
}

// --------------------------------------------------------------------------------------------------
// dependent-angle projection in place.  dynamic LDS: q[nq] | sc[6 nl]
__global__ __launch_bounds__(WAVE) void k_project(const DevModel* __restrict__ M, double* __restrict__ q,
                                                  int* __restrict__ clamped_flag) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const int nq = M->nq;
    double* sq = smem;
    double* ssc = sq + nq;
    const size_t f = blockIdx.x;
    if (lane < nq) sq[lane] = q[f * nq + lane];
    wave_lds_sync();
    wave_sincos(M, sq, ssc, lane);
    wave_lds_sync();
    const int cl = wave_project_joints(M, sq, ssc, lane);
    if (lane < nq) q[f * nq + lane] = sq[lane];
    if (cl && lane == 0 && clamped_flag) atomicOr(clamped_flag, 1);
}

#include "cpe_solver.hip.inc"
#include "cpe_kinetic.hip.inc"
#include "cpe_kinetic_jac.hip.inc"
