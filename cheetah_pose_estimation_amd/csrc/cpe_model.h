// cpe_model.h -- device-side model tables (built on the host in cpe_create, read-only on the GPU).
// gfx950 only.  All tables are derived from include/cpe.h's cpe_skeleton / cpe_camera / cpe_options.
#pragma once
#include <stdint.h>

#include "../../include/cpe.h"

#define CPE_MAX_CHAIN 6     // links on the path root -> marker link (paw: base,bodyF,thigh,calf,hock)
#define CPE_MAX_TL 512      // terms of all reduced marker columns (cheetah: ~300)
#define CPE_MAX_SLOTS 320   // structurally non-zero (marker,dof) pairs; 276 for the 25-marker cheetah
#define CPE_MAX_MCOL 12     // reduced (independent) dofs a marker depends on
#define CPE_MAX_TERMS 8     // terms of one reduced marker-Jacobian column (1 direct + dependent angles)
#define CPE_MAX_HG 56        // contributions Dp_i^T M Dp_j one lane gathers into its entries of H (cheetah: ~21)
#define CPE_MAX_GG 32        // markers one reduced dof can move
#define CPE_MAX_DEP 32      // dependent angles (26)
#define CPE_MAX_SCOL 8      // columns of one row of S = d(dependent)/d(independent)
#define CPE_MAX_SDYN 48     // dynamic (alpha-dependent) body-frame vectors of the leg markers

struct DevModel {
    int32_t nl, L, C, nq, nu, S, nj, nb, ndep, curvature;
    double h, ih2, loss_a, loss_b, loss_c, bound_penalty, rho0;

    int32_t parent[CPE_MAX_LINKS];
    double attach[CPE_MAX_LINKS][3];
    double com[CPE_MAX_LINKS][3];
    double mass[CPE_MAX_LINKS];
    double inv_total_mass;

    // marker chains: p_l = x_base + sum_k R_{chain_link[l][k]} * chain_vec[l][k]
    int32_t chain_len[CPE_MAX_MARKERS];
    int32_t chain_link[CPE_MAX_MARKERS][CPE_MAX_CHAIN];
    double chain_vec[CPE_MAX_MARKERS][CPE_MAX_CHAIN][3];

    // Jacobian slots (marker-major): slot -> marker, q index, chain position (-1: translation), angle
    int32_t slot_off[CPE_MAX_MARKERS + 1];
    int32_t slot_marker[CPE_MAX_SLOTS];
    int32_t slot_dof[CPE_MAX_SLOTS];
    int32_t slot_cpos[CPE_MAX_SLOTS];
    int32_t slot_ang[CPE_MAX_SLOTS];

    cpe_camera cam[CPE_MAX_CAMS];

    // joints, processed parents first
    int32_t joint_parent[CPE_MAX_JOINTS], joint_child[CPE_MAX_JOINTS], joint_kind[CPE_MAX_JOINTS];
    int32_t joint_body[CPE_MAX_JOINTS];     // revolute: link whose y axis the whole chain shares
    int32_t joint_dep0[CPE_MAX_JOINTS];     // first dependent row of this joint (phi; psi = +1 if revolute)

    int32_t indep[CPE_NX];
    // COST PITCH of the leg links: the terms of the objective that act on the Euler pitch of a leg link -- constant-acceleration cost, joint
    // ranges, learned priors -- take theta_B + alpha_c (pitch of the body the leg hangs from + the leg angle about the body's y axis): the reference's
    // variable for an unrolled trunk, smooth through +-90 degrees.  ucost[k] = s0 | (s1 + 1) << 8: value of weighted coordinate k = state[s0] (+ state[s1]).
    int32_t ucost[CPE_NX];
    int32_t u_of_q[CPE_MAX_NQ];             // -1 for dependent dofs
    int32_t dep_of_q[CPE_MAX_NQ];           // -1 for independent dofs
    double motion_w_u[CPE_NX];
    double rel_sign_u[CPE_NX];              // x_k = rel_sign_u[k] * (u_k - u_{rel_ref_u[k]})  (rel_ref_u < 0: x_k = u_k)
    int32_t rel_ref_u[CPE_NX];

    int32_t bound_ua[CPE_MAX_BOUNDS], bound_ub[CPE_MAX_BOUNDS];
    double bound_lo[CPE_MAX_BOUNDS], bound_up[CPE_MAX_BOUNDS];

    // S rows: d(dependent angle r)/d(u_{scol[r][j]}), j < scol_n[r]
    int32_t scol_n[CPE_MAX_DEP];
    int32_t scol[CPE_MAX_DEP][CPE_MAX_SCOL];

    // reduced marker Jacobian Dp_l[:, j] (j < mcol_n[l], reduced column mcol[l][j]) =
    //   sum_{t < term_n} dp[term_slot] * (term_s < 0 ? 1 : Sval[term_s])     (Sval flat index = srow[r]*CPE_MAX_SCOL + j)
    int32_t mcol_n[CPE_MAX_MARKERS];
    int32_t mcol[CPE_MAX_MARKERS][CPE_MAX_MCOL];
    int32_t mcol_off[CPE_MAX_MARKERS + 1];                 // prefix sum of mcol_n
    int32_t term_n[CPE_MAX_MARKERS][CPE_MAX_MCOL];
    alignas(16) int16_t term_slot[CPE_MAX_MARKERS][CPE_MAX_MCOL][CPE_MAX_TERMS];     // one 16-byte load per column
    alignas(16) int16_t term_s[CPE_MAX_MARKERS][CPE_MAX_MCOL][CPE_MAX_TERMS];
    // flat task list over all (marker, reduced column) pairs
    int32_t mc_total;
    // gather form of H = sum_l Dp_l^T M_l Dp_l (lower triangle, mirrored on store) and g = sum_l Dp_l^T v_l: every lane
    // owns a balanced set of entries and sums their contributions in a register -- no read-modify-write of H in LDS,
    // no per-marker synchronisation.  hg_code = ci | cj << 8 | marker << 16 | first-of-entry << 21 | (a << 5 | b) << 22 with
    // ci, cj = reduced-column indices into Dp and (a, b) the entry of H the contribution belongs to; lane-interleaved.
    int32_t hg_cnt[64], hg_max;
    uint32_t hg_code[CPE_MAX_HG][64];
    int32_t gg_cnt[CPE_NX];
    uint16_t gg_code[CPE_MAX_GG][CPE_NX];             // ci | marker << 8
    // flat list of the terms of all reduced marker columns (k_frame_normal builds Dp from it, one term per lane per round, no staging of the
    // slot vectors): item = (marker, column) index into Dp; w0 = item | (S index + 1) << 10 | (dynamic vector + 1) << 20; moff / vec as ss_*
    int32_t tl_n;
    struct alignas(16) { int32_t w0, moff; double v[3]; } tl[CPE_MAX_TL];
    uint32_t h_covered[(CPE_NX * CPE_NX + 31) / 32];   // entries of H (row-major) that the gather list writes; the others are structural zeros
    int16_t mc_marker[CPE_MAX_MARKERS * CPE_MAX_MCOL];
    int16_t mc_j[CPE_MAX_MARKERS * CPE_MAX_MCOL];

    // Hooke rows of S: entry j of row r = -(direct)/g_phi + t0 * Sval[hk_chain]
    //   hk_kind 0: child angle hk_ang (y_p . dR_c[ang] e_z) ; 1: parent angle hk_ang (dR_p[ang] e_y . z_c) ; 2: none
    //   hk_chain: flat Sval index of the parent's dependent-phi row entry for the same column, or -1
    int8_t hk_kind[CPE_MAX_DEP][CPE_MAX_SCOL];
    int8_t hk_ang[CPE_MAX_DEP][CPE_MAX_SCOL];
    int16_t hk_chain[CPE_MAX_DEP][CPE_MAX_SCOL];
    int32_t dep_joint[CPE_MAX_DEP];      // joint that defines dependent row r
    int32_t dep_level[CPE_MAX_DEP];      // 0: parent fully independent (or revolute); 1: parent's phi dependent
    int32_t srow[CPE_MAX_DEP], n_srow;   // hooke rows only get S values computed: compact row index (Sval flat index = srow * CPE_MAX_SCOL + j), -1 otherwise

    // ---- solver parametrisation: leg link c = body B rotated about B's y axis by alpha_c.
    // State of one frame: ns = nq + nrev doubles = Euler q followed by the leg angles alpha.
    int32_t nrev, ns, n_trunk, ss_n, sv_n;
    int32_t rev_child[CPE_MAX_JOINTS], rev_body[CPE_MAX_JOINTS], rev_u[CPE_MAX_JOINTS];
    int32_t rev_body_u[CPE_MAX_JOINTS][3];          // reduced index of the body's phi, theta, psi
    int32_t rev_of_u[CPE_NX];                       // -1 or revolute index
    int32_t ucoord_src[CPE_NX];                     // position of reduced coordinate k in the state vector
    int32_t euler_rev[CPE_MAX_NQ];                  // revolute index if Euler dof p is a leg pitch theta_c, else -1
    int32_t bodyang_j[CPE_NX];                      // k is angle j (0..2) of a body that carries legs, else -1
    int32_t bodyang_legs_n[CPE_NX];
    int32_t bodyang_legs[CPE_NX][CPE_MAX_JOINTS];
    int32_t trunk_link[CPE_MAX_LINKS];              // links whose R / dR are needed (not leg links)
    int32_t trunk_slot[CPE_MAX_LINKS];              // position of a link in trunk_link (R block = 36 doubles per slot), -1 for leg links
    // dynamic body-frame vectors: kind 0: sum_i Ry(alpha_{rev[i]}) vec[i] ; kind 1: dRy/dalpha(alpha_{rev[0]}) vec[0]
    int32_t sv_kind[CPE_MAX_SDYN], sv_cnt[CPE_MAX_SDYN], sv_rev[CPE_MAX_SDYN][3];
    double sv_vec[CPE_MAX_SDYN][3][3];
    // marker position: x + sum_k R_{pc_link} pc_vec + (pw_id >= 0 ? R_{pw_body} dyn[pw_id] : 0)
    int32_t pc_len[CPE_MAX_MARKERS], pc_link[CPE_MAX_MARKERS][CPE_MAX_CHAIN], pw_id[CPE_MAX_MARKERS], pw_body[CPE_MAX_MARKERS];
    double pc_vec[CPE_MAX_MARKERS][CPE_MAX_CHAIN][3];
    // k_frame_normal's per-lane tables with every index resolved on the host (no table read depends on another one):
    //   hooke S items, one per lane: w0 = parent slot | child slot << 6 | hk_kind << 12 | hk_ang << 14 | level << 16, w1 = (hk_chain + 1) | Sval index << 16;
    //   marker chains as offsets into the trunk rotation table (36 * trunk_slot; entries past pc_len: offset 0 and a zero vector);
    //   bounds as indices into the Euler state: q index of a | (q index of b + 1) << 8
    //   first-phase words, one int4 per lane: x = hooke joint `lane` (parent | child << 8 | parent's phi dependent << 16; hj_n of them),
    //   y = trunk link of rotation block `lane` (lane >> 2 indexes trunk_link), z = leg link `lane` (body | child << 8 | body's trunk slot << 16),
    //   w = dynamic vector `lane` (sv_kind | sv_cnt << 2 | sv_rev[0..2] << 4, 10, 16)
    int32_t hj_n;
    alignas(16) int32_t fn_lane[64][4];
    int32_t hk_n, hk_w[64][2];
    int32_t pc_off[CPE_MAX_MARKERS][CPE_MAX_CHAIN], pw_off[CPE_MAX_MARKERS];
    int32_t bnd_q[CPE_MAX_BOUNDS];
    // solver Jacobian slots: dp = Mat(ss_moff) * (ss_vdyn < 0 ? ss_vec : dyn[ss_vdyn]); ss_moff < 0: identity
    int32_t ss_moff[CPE_MAX_SLOTS], ss_vdyn[CPE_MAX_SLOTS];
    double ss_vec[CPE_MAX_SLOTS][3];
};

// learned priors (config 3) in device memory: the ABI struct plus host-precomputed x-space Gauss-Newton blocks of the
// autoregressive prior, slack_n = sum_{t=0..W} K_t x_{n-W+t} - b with K_W = I, K_t = -coef[:, t] (acinoset_misc.py:291-336):
// lr_PK[ta][tb] = 2 K_ta^T diag(w) K_tb for ta >= tb (frame distance ta - tb);
// lr_HI[k] = sum_ta lr_PK[ta][ta-k]: the block at frame distance k for a frame away from the sequence ends.
struct DevPriors {
    cpe_priors p;
    double lr_PK[CPE_MAX_WINDOW + 1][CPE_MAX_WINDOW + 1][CPE_NX * CPE_NX];
    double lr_HI[CPE_MAX_WINDOW + 1][CPE_NX * CPE_NX];
    // the input features (lag, coordinate) that ANY output of the motion model uses -- the multi-task lasso of the reference keeps 36 of 112 -- in
    // ascending order, and the coefficients transposed (output index fastest: consecutive lanes read consecutive words)
    int32_t lr_nf, _pad_nf;
    uint8_t lr_feat[CPE_MAX_WINDOW * CPE_NX];
    double lr_coefT[CPE_MAX_WINDOW * CPE_NX][CPE_NX];
    // Since the cost pitch the relative angles are LINEAR in the reduced coordinates, x = X' u with a constant X' (entries 0, +-1): every
    // second-order piece of the two priors is a constant of the model, built once by the host (cpe_api.hip):
    double Xc[CPE_NX][CPE_NX];                                                         // X' (row = relative angle, column = reduced coordinate)
    double lr_PKu[CPE_MAX_WINDOW + 1][CPE_MAX_WINDOW + 1][CPE_NX * CPE_NX];            // X'^T lr_PK[ta][tb] X'
    double lr_HIu[CPE_MAX_WINDOW + 1][CPE_NX * CPE_NX];                                // X'^T lr_HI[k] X'
    double gmm_PT[CPE_MAX_GMM][CPE_NX][CPE_NX];                                        // gmm_P[k] transposed (consecutive lanes read consecutive words)
    double gmm_Q[CPE_MAX_GMM][CPE_NX * CPE_NX];                                        // X'_g^T P_k X'_g, X'_g = the rows of X' the pose prior sees
};

// per-sequence Levenberg-Marquardt state (device global memory)
struct SeqState {
    int32_t cur;        // which of the two buffers holds the current iterate
    int32_t status;     // 0 running, 1 converged, 2 numerical failure, 3 iteration limit
    int32_t iters;
    int32_t rejects;
    int32_t outer;      // augmented-Lagrangian multiplier updates done
    int32_t al_pending; // 1: the next k_frame_normal updates the multipliers at the current iterate
    int32_t back_pending; // 1: k_lm_step has factored and left z; k_lm_back still has to solve and write the trial iterate
    int32_t fresh;      // 1: the current iterate is new since its second-order pieces were last built (accepted step, first evaluation, multiplier update)
    double lambda, nu, cost_cur, pred, maxstep, maxviol;
    double terms[5];    // meas, model, bound, pose, motion at the current iterate
};
