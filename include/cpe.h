/*
 * cpe.h -- C ABI of the MI355X trajectory-optimisation back end for cheetah 3D pose.
 *
 * This is the drop-in boundary.  The reference (zicodasilva/cheetah_pose_estimation) has no FFI: its
 * seam is the hand-off of the Pyomo model to the IPOPT executable,
 *     pe.utils.default_solver(...).solve(robot.m)          acinoset_opt.py:611-617
 * plus the model-building calls around it (acinoset_opt.py:413-536, :539-635).  The entry points below
 * are what a ctypes binding placed at that seam binds.  Plain pointers and sizes,
 * caller-owned buffers, integer status codes, no torch types.
 *
 * Conventions:
 *   - q[54] = [x,y,z,phi,theta,psi]_base then [phi,theta,psi] of links 1..16, absolute ZYX Euler angles;
 *     link order base,bodyF,neck,tail0,tail1,UFL,LFL,HFL,UFR,LFR,HFR,UBL,LBL,UBR,LBR,HBL,HBR
 *     (cheetah.py:197-198).  DOF index of angle j of link i is 3 + 3*i + j.
 *   - all arrays are C-contiguous fp64, frames are the slowest index inside a sequence:
 *     q[B][N][nq], meas[B][N][C][L][2], weight[B][N][C][L].
 *   - "device" entry points take pointers into HBM (e.g. torch-ROCm tensor.data_ptr()); the *_host
 *     variants take host pointers and stage through HBM (PCIe-inclusive).
 */
#ifndef CPE_H
#define CPE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPE_MAX_LINKS 20
#define CPE_MAX_MARKERS 32
#define CPE_MAX_CAMS 18      /* 6 cameras x 3: with pairwise pseudo-measurements (m.W = RangeSet(3), acinoset_misc.py:179) every camera
                              * appears three times with the same parameters and its own meas / weight slice */
#define CPE_MAX_JOINTS 16
#define CPE_MAX_BOUNDS 32
#define CPE_MAX_NQ (3 + 3 * CPE_MAX_LINKS)
#define CPE_MAX_GMM 8        /* mixture components of the pose prior */
#define CPE_NX 28            /* size of the reduced / relative-angle vector x (acinoset_misc.py:1699-1757) */
#define CPE_MAX_WINDOW 4     /* window of the linear motion prior (acinoset_opt.py:545) */

typedef int32_t cpe_status;
#define CPE_OK 0              /* converged                                             */
#define CPE_MAX_ITER 1        /* iteration limit reached (solution still written)      */
#define CPE_NUMERICAL 2       /* numerical failure (non-finite cost / factorisation)   */
#define CPE_BAD_ARG (-1)
#define CPE_NO_DEVICE (-2)    /* HIP runtime / GPU unavailable: there is NO CPU fallback */
#define CPE_HIP_ERROR (-3)

/* joint kinds (cheetah.py:71-72,101,160-161): the equalities live in the .robot `angle_constraints` */
#define CPE_JOINT_REVOLUTE_Y 0   /* (R_p e_y).(R_c e_x) = 0 and (R_p e_y).(R_c e_z) = 0 ; child phi,psi dependent */
#define CPE_JOINT_HOOKE_YZ 1     /* (R_p e_y).(R_c e_z) = 0                            ; child phi dependent     */

/* camera models (acinoset_misc.py:1663-1696) */
#define CPE_CAM_FISHEYE 0
#define CPE_CAM_PINHOLE 1

/*
 * Skeleton = absolute-orientation articulated chain.  Every point of the body is
 *     x_base + sum_{k on the path root->link} R_k(q) * v_k
 * origin_i = origin_parent(i) + R_parent(i) * attach[i]   (origin_base = q[0:3])
 * com_i    = origin_i + R_i * com[i]                      (Link3D.Pb_I)
 * marker_l = origin_link(l) + R_link(l) * marker_off[l]   (acinoset_misc.py:1586-1659)
 */
typedef struct cpe_skeleton {
    int32_t n_links;
    int32_t n_markers;
    int32_t n_joints;
    int32_t n_bounds;
    int32_t parent[CPE_MAX_LINKS];
    double attach[CPE_MAX_LINKS][3];
    double com[CPE_MAX_LINKS][3];
    double mass[CPE_MAX_LINKS];
    int32_t marker_link[CPE_MAX_MARKERS];
    double marker_off[CPE_MAX_MARKERS][3];
    int32_t joint_parent[CPE_MAX_JOINTS];
    int32_t joint_child[CPE_MAX_JOINTS];
    int32_t joint_kind[CPE_MAX_JOINTS];
    /* lo <= q[a] - q[b] <= up ; b < 0 means a plain bound on q[a]   (cheetah.py:306-352) */
    int32_t bound_a[CPE_MAX_BOUNDS];
    int32_t bound_b[CPE_MAX_BOUNDS];
    double bound_lo[CPE_MAX_BOUNDS];
    double bound_up[CPE_MAX_BOUNDS];
    /* constant-acceleration model weight 1/Q_p^2, 0 where Q_p = 0 (acinoset_misc.py:234,1852-1908) */
    double motion_w[CPE_MAX_NQ];
    /* relative angles (acinoset_misc.py:508-528): rel[p] = rel_sign[p] * (q[p] - q[rel_ref[p]]),
     * rel_ref[p] < 0 means rel[p] = q[p].  x (28) = rel restricted to the independent dofs. */
    int32_t rel_ref[CPE_MAX_NQ];
    double rel_sign[CPE_MAX_NQ];
} cpe_skeleton;

typedef struct cpe_camera {
    int32_t model;      /* CPE_CAM_FISHEYE | CPE_CAM_PINHOLE */
    int32_t _pad;
    double fx, fy, cx, cy;
    double D[4];        /* fisheye k1..k4 ; pinhole uses D[0..2] */
    double R[9];        /* row-major world->camera rotation */
    double t[3];
    double mult;        /* cam_uncertainty_multiplier (acinoset_misc.py:462-464) */
} cpe_camera;

/* learned priors of the monocular "data-driven" model (config 3) */
typedef struct cpe_priors {
    int32_t gmm_k;                                   /* 0 = pose prior off (acinoset_misc.py:680-714) */
    int32_t gmm_dim;                                 /* 22 = x[6:]                                    */
    double gmm_logw[CPE_MAX_GMM];                    /* log(w_k) - 0.5*logdet(2 pi Sigma_k)           */
    double gmm_mu[CPE_MAX_GMM][CPE_NX];
    double gmm_P[CPE_MAX_GMM][CPE_NX][CPE_NX];       /* precision matrices Sigma_k^-1                 */
    int32_t lr_window;                               /* 0 = motion prior off (acinoset_misc.py:291-336) */
    int32_t _pad;
    double lr_coef[CPE_NX][CPE_MAX_WINDOW * CPE_NX]; /* y = coef . [x_{n-w};...;x_{n-1}] + b, oldest first */
    double lr_b[CPE_NX];
    double lr_w[CPE_NX];                             /* 1/error_variance (0 if variance is 0)         */
} cpe_priors;

typedef struct cpe_options {
    double h;            /* 1/fps; implicit-Euler step of make_pyomo_model (acinoset_opt.py:508) */
    double loss_a, loss_b, loss_c;  /* redescending loss knots 3,10,20 (acinoset_misc.py:479-481) */
    double cost_scale;   /* 1e-3 (acinoset_opt.py:602); only scales the reported objective      */
    double bound_penalty;/* kappa of the augmented Lagrangian that enforces the 23 angle bounds   */
    double bound_tol;    /* bounds are met when the largest violation is below this (rad)         */
    double lambda0;      /* initial Levenberg-Marquardt damping (default 1e-4)                  */
    double tol_step;     /* converged when max |du| < tol_step                                  */
    double tol_cost;     /* ... or relative cost decrease < tol_cost (default 1e-9: within 0.01 mm RMSE of the 1e-12 solution;
                          * the reference runs IPOPT with Tol = 1e-3, acinoset_opt.py:611-617)   */
    int32_t max_iter;
    int32_t curvature;   /* 0: max(rho'', rho'(s)/s, 0) ; 1: max(rho''(s), 0)                     */
    int32_t max_outer;   /* multiplier updates of the augmented Lagrangian (0 = pure penalty)     */
    int32_t _pad;
} cpe_options;

typedef struct cpe_stats {
    int32_t status;      /* cpe_status of this sequence */
    int32_t iterations;  /* LM iterations (accepted + rejected) */
    double cost;         /* final objective, already multiplied by cost_scale */
    double cost_meas, cost_model, cost_pose, cost_motion;   /* estimator.costs (acinoset_opt.py:603-608) */
    double lambda;       /* final damping */
    double max_constraint; /* max |joint equality| at the solution */
    double max_bound_violation; /* largest violation of an angle bound at the solution (rad) */
    int32_t outer;       /* multiplier updates performed */
    int32_t _pad;
} cpe_stats;

typedef struct cpe_handle cpe_handle;

/* ---- life cycle -------------------------------------------------------------------------------- */
/* replaces init_trajectory()'s model construction (acinoset_opt.py:459-525). `priors` may be NULL. */
cpe_status cpe_create(const cpe_skeleton* skel, const cpe_camera* cams, int32_t n_cams,
                      const cpe_options* opts, const cpe_priors* priors, int32_t device,
                      cpe_handle** out);
/* frees everything cpe_create allocated (the reference drops its Pyomo model and calls gc.collect() between sequences,
 * run_dataset.py:1145-1231) */
void cpe_destroy(cpe_handle* h);
/* text of the last failure on this thread's most recent call (the reference raises Python exceptions, acinoset_opt.py:400-406) */
const char* cpe_last_error(void);
/* the values the reference hard-codes: loss knots (3, 10, 20) acinoset_misc.py:2001-2015, fps-derived h acinoset_opt.py:483-487 */
void cpe_default_options(cpe_options* o);
/* stream the handle launches on (hipStream_t as void*), for event timing by the caller */
void* cpe_stream(cpe_handle* h);
cpe_status cpe_synchronize(cpe_handle* h);
/* Synchronisation contract of the device-pointer entry points: they only ENQUEUE work on the handle's own (non-blocking) stream
 * and order nothing against any other stream.  A caller that produced the inputs on another stream (`other`, a hipStream_t as
 * void*; NULL = the legacy default stream; e.g. torch's current stream) calls cpe_stream_wait before the entry point, and
 * cpe_stream_signal after it when that stream will consume the outputs; or it calls cpe_synchronize.  (The reference is
 * synchronous: Pyomo blocks on the IPOPT subprocess, acinoset_opt.py:611-617.) */
cpe_status cpe_stream_wait(cpe_handle* h, void* other);     /* later launches of the handle wait for the work queued on `other` so far */
cpe_status cpe_stream_signal(cpe_handle* h, void* other);   /* `other` waits for the work the handle has queued so far */
/* per-kernel device time of cpe_solve / cpe_solve_kinetic, accumulated by HIP events on the handle's stream while enabled
 * (what the reference stores as processing_time_s is one wall time around .solve(), acinoset_opt.py:610-618).
 * slots: 0 k_frame_normal, 1 k_lr_band, 2 k_lm_step, 3 k_build_act, 4 k_finalize, 5 k_dyn_eval, 6 k_dyn_gather, 7 k_lm_back,
 *        8 k_dyn_assemble, 9 k_dyn_schur, 10 k_dyn_jac, 11 free */
#define CPE_PROFILE_SLOTS 12
cpe_status cpe_profile_enable(cpe_handle* h, int32_t on);                              /* also clears the totals */
cpe_status cpe_profile_get(cpe_handle* h, double* ms /*[12]*/, int64_t* launches /*[12]*/);

/* number of Jacobian slots: the structurally non-zero (marker, dof) pairs, sum_l (3 + 3*chain_len(l)), followed by 0-3
 * structurally ZERO pairs (marker 0, a dof outside its chain; the stored value is 0) that round the count up to a multiple of 4,
 * so that every camera row of J starts on a 64-byte line */
int32_t cpe_jacobian_slots(const cpe_handle* h);       /* (the non-zeros of d measurement_constraints / dq, acinoset_misc.py:278-288) */
/* slot -> (marker, dof) tables; caller provides int32[cpe_jacobian_slots] each */
cpe_status cpe_jacobian_layout(const cpe_handle* h, int32_t* slot_marker, int32_t* slot_dof);
/* the reduced coordinates x of get_relative_angles + get_relative_angle_mask (acinoset_misc.py:487-528, :1699-1757) */
int32_t cpe_num_independent(const cpe_handle* h);                      /* 28 */
cpe_status cpe_independent_dofs(const cpe_handle* h, int32_t* dofs);   /* q index of each reduced coordinate */

/* ---- metric 1: residual + Jacobian evaluation ------------------------------------------------------
 * One pass of acinoset_misc.py:269-288 (pose + measurement constraints) and :639-677 (acceleration
 * slack) over B*N frames; all pointers are DEVICE pointers.
 *   r     [B][N][C][L][2]      reprojection residual  proj(q) - meas   (= slack_meas / fte.pickle meas_err)
 *   J     [B][N][C][S][2]      d r / d q on the S structurally non-zero (marker,dof) slots
 *   eps   [B][N][nq]           acceleration slack ddq_n - ddq_{n-1} (0 for n < 3, free initial states)
 *   cost  [B][N]   (optional)  per-frame sum_c,l,d rho(mult*w*r)  (acinoset_misc.py:459-484), unscaled
 */
cpe_status cpe_eval_resjac(cpe_handle* h, int32_t B, int32_t N, const double* q, const double* meas,
                           const double* weight, double* r, double* J, double* eps, double* cost);

/* ---- dependent-angle projection: closed-form solve of the 26 joint equalities for the 26 dependent
 * angles given the 28 independent ones (in place on q[B][N][nq], device pointer). */
/* (the `angle_constraints` of the .robot model, created by add_revolute_joint / add_hookes_joint, cheetah.py:71-72,101,160-161) */
cpe_status cpe_project_joints(cpe_handle* h, int32_t B, int32_t N, double* q);

/* ---- building block of the solver: per-frame terms in the reduced coordinates at the Euler
 * iterate q (leg angles are taken as rotations of their body about its y axis, tails re-projected).
 * Device pointers.  g [B][N][28]; Bm [B][N][28][28] (measurement + bound + pose-prior Gauss-Newton block);
 * cost [B][N][3] = {robust measurement cost, bound term, pose-prior term}; gam [B][N][nrev][4] = d (cost pitch of the leg link) / d(alpha, phi_B, theta_B, psi_B) = (1, 0, 1, 0) since round 3;
 * q_out [B][N][nq] = the consistent Euler angles.  gam and q_out may be NULL. */
/* (per frame, what ASL hands IPOPT for measurement_cost acinoset_misc.py:459-484, the angle bounds cheetah.py:306-352 and
 * gmm_pose_cost acinoset_misc.py:680-714: value, gradient and Hessian block) */
cpe_status cpe_eval_normal(cpe_handle* h, int32_t B, int32_t N, const double* q, const double* meas, const double* weight,
                           double* g, double* Bm, double* cost, double* gam, double* q_out);

/* ---- metric 2: full-trajectory solve ---------------------------------------------------------------
 * replaces default_solver(...).solve(robot.m) for the kinematic model (acinoset_opt.py:589-617).
 * Device pointers.  q_init [B][N][nq] (dependent angles are re-projected), outputs as the reference
 * saves them (acinoset_opt.py:289-361):
 *   q,dq,ddq [B][N][nq] ; positions [B][N][L][3] ; meas_err [B][N][C][L][2] ; stats[B] (HOST pointer)
 * Leg links: q holds the principal ZYX triple of every link (|pitch| <= pi / 2; the same rotation has the triple (phi + pi, pi - theta, psi + pi), which
 * is the one the reference's variables are on while a limb is beyond the horizontal).  The three terms of the objective that act on a leg link's pitch
 * itself -- constant-acceleration cost (acinoset_misc.py:639-677), joint ranges (cheetah.py:306-352), learned priors (acinoset_misc.py:291-336, :680-714)
 * -- are evaluated on theta_body + alpha_link (the leg's angle about its body's y axis): the reference's variable for an unrolled trunk, smooth through
 * +-pi / 2 ("cost pitch").
 */
cpe_status cpe_solve(cpe_handle* h, int32_t B, int32_t N, const double* q_init, const double* meas,
                     const double* weight, double* q, double* dq, double* ddq, double* positions,
                     double* meas_err, cpe_stats* stats);

/* ---- shutter-delay estimation (`shutter_delay_estimation=True`, acinoset_misc.py:179-183, :274-288; run_dataset.py:1323) -----------
 * The reference adds one unknown delay tau_c per camera (camera 1 fixed to 0, |tau_c| <= h) and lets camera c see every marker displaced by
 * q'_base tau_c + q''_base tau_c^2 (implicit-Euler velocity and acceleration of the base position).  Solved here by block-coordinate descent on
 * the reference's objective: cpe_solve's LM with the delays fixed (exact gradient, including the parts that reach frames n-1 and n-2 through q',
 * q''; curvature inside the frame exact, cross-frame curvature blocks left out of the Gauss-Newton model), then one Newton step per delay with
 * the trajectory fixed; the sequence of delay vectors is accelerated by Anderson mixing (memory 4, host side, per sequence: the plain alternation
 * contracts slowly where all delays move together and the trajectory shifts in time), clamped to +-tau_bound; repeated until no delay moves by
 * more than tol_tau (<= max_rounds), then a last trajectory solve.
 * Deviation: the displacement acts from node 2 on (q'_0 and q''_0 are free variables of the reference's collocation: its first two nodes can
 * absorb any displacement).  Device pointers; tau [B][C] (output; tau[.][0] = 0); stats HOST; rounds (HOST, optional) = delay updates done. */
cpe_status cpe_solve_shutter(cpe_handle* h, int32_t B, int32_t N, const double* q_init, const double* meas, const double* weight, double tau_bound,
                             int32_t max_rounds, double tol_tau, double* q, double* dq, double* ddq, double* positions, double* meas_err,
                             double* tau, cpe_stats* stats, int32_t* rounds);

/* host-pointer convenience wrappers of cpe_eval_resjac / cpe_solve above (same reference counterparts: acinoset_misc.py:269-288,
 * acinoset_opt.py:611-617); they stage through HBM, so their rates are PCIe-inclusive */
cpe_status cpe_eval_resjac_host(cpe_handle* h, int32_t B, int32_t N, const double* q, const double* meas,
                                const double* weight, double* r, double* J, double* eps, double* cost);
cpe_status cpe_solve_host(cpe_handle* h, int32_t B, int32_t N, const double* q_init, const double* meas,
                          const double* weight, double* q, double* dq, double* ddq, double* positions,
                          double* meas_err, cpe_stats* stats);

/* ---- per-frame ground-reaction-force fit (CheetahEstimator.estimate_grf, acinoset_opt.py:176-270).
 * Rows 0-5 (root x, y, z, phi, theta, psi) of d/dt dL/dq' - dL/dq for L = sum_i (m_i |P_i'|^2 / 2 + w_i^T I_i w_i / 2
 * - m_i g P_i,z) are balanced against B = sum_feet (d foot / dq)^T M g (GRFz e_z + sum_k D_k GRFxy_k),
 * D = [+x, +y, -x, -y], for the feet flagged in contact:  minimise |rows - B| subject to 0 <= GRF <= force_max and
 * friction_ratio * GRFz >= sum_k GRFxy_k per foot (acinoset_opt.py:183-192).  Forces are in body weights (M g).
 * The reference hands each frame to IPOPT; the least-squares problem is convex but not strictly (opposing friction
 * components, three or more feet), so IPOPT's answer is one point of a face of minimisers.  Here the face is resolved
 * by the minimum-norm minimiser (Tikhonov term `regularisation`), computed by FISTA with exact projections, a fixed
 * `iterations` count, identically in oracle and HIP.  Only the root link's inertia enters rows 0-5 (all other
 * orientations are independent coordinates). */
typedef struct cpe_grf_options {
    double root_inertia[3];   /* principal moments of the root link about its body axes (cheetah: cylinder along x)   */
    double friction_ratio;    /* 1.3 (acinoset_opt.py:190)                                                            */
    double force_max;         /* 5.0 (acinoset_opt.py:186-187)                                                        */
    double regularisation;    /* 1e-6                                                                                 */
    double gravity;           /* 9.81                                                                                 */
    int32_t n_feet;           /* <= 4                                                                                 */
    int32_t foot_marker[4];   /* marker sitting at hock.bottom of each foot (the paw markers), order = output order   */
    int32_t iterations;       /* FISTA iterations (2000)                                                              */
} cpe_grf_options;

/* q, dq, ddq [B][N][nq]; contact [B][N][n_feet] (0 / 1); grfz [B][N][n_feet]; grfxy [B][N][n_feet][4];
 * residual [B][N][6] = rows - B at the solution, in units of M g (may be NULL).  Device pointers. */
/* CheetahEstimator.estimate_grf, acinoset_opt.py:176-270 (one IPOPT launch per frame there) */
cpe_status cpe_grf_fit(cpe_handle* h, const cpe_grf_options* opt, int32_t B, int32_t N, const double* q, const double* dq,
                       const double* ddq, const int32_t* contact, double* grfz, double* grfxy, double* residual);

/* ---- all rows of the equations of motion (the residual function of the physics-based model, config 4 / SURVEY row a12:
 * `make_pyomo_model(include_eom_slack=True)`, acinoset_opt.py:510-514): rows [B][N][nq] = d/dt dL/dq' - dL/dq in N and N m for
 * L = sum_i (m_i |P_i'|^2 / 2 + w_i^T I_i w_i / 2 - m_i g P_i,z), evaluated from (q, q', q'') without forming M, C, G:
 * row of angle a of link i = f_i . (dR_i/da c_i) + sum over the children c of i  F_subtree(c) . (dR_i/da attach_c)
 *                            + (I_i alpha_i + w_i x I_i w_i) . dw_i/dq'_a,     f_j = m_j (P_j'' + g e_z).
 * What the reference balances these rows against (motor torques, the 26 joint constraint forces, contact forces, slack) is
 * cpe_dyn_forces / cpe_eom_residual below, and the solve with all of them as unknowns is cpe_solve_kinetic.  link_inertia:
 * principal moments of every link about its body axes.  Device pointers. */
typedef struct cpe_eom_options {
    double gravity;
    double link_inertia[CPE_MAX_LINKS][3];
} cpe_eom_options;
/* the lambdified equations of motion `eom_f` of the .robot model (acinoset_opt.py:120-161, :510-514) evaluated at (q, dq, ddq) */
cpe_status cpe_eom_rows(cpe_handle* h, const cpe_eom_options* opt, int32_t B, int32_t N, const double* q, const double* dq,
                        const double* ddq, double* rows);

/* ---- generalised forces of the physics-based model: what the rows above are balanced against,
 *   Q = B_grf + B_tau + (dc/dq)^T lambda,
 *   B_grf = sum_feet (d foot / dq)^T  M g (GRFz e_z + sum_k D_k GRFxy_k),  D = [+x, +y, -x, -y]   (forces in body weights)
 *   B_tau = sum_motors (J_w,second - J_w,first)^T  M g tau  R_first e_axis,  J_w,i = d w_i(world) / dq' = R_i dw_i(body)/dq'
 *   c     = the joint equalities in the order of cpe_skeleton.joint_* (two rows per revolute, one per hooke joint)
 * cpe_eom_residual returns rows - Q (the reference's slack_eom).  The pairing of the reference's motor names with
 * (first, second, axis) follows cheetah.py:70-165 and is supplied by the caller; against the reference's `.robot` pickles
 * this is "parity unpinned" -- the tests check virtual work instead (Q . q' = sum F . v_foot + sum T . (w_second - w_first)). */
typedef struct cpe_dyn_options {
    cpe_eom_options eom;
    int32_t n_feet, n_motors;
    int32_t foot_marker[4];
    int32_t motor_first[32], motor_second[32], motor_axis[32];      /* axis 0/1/2 = x/y/z of the FIRST link */
} cpe_dyn_options;
/* tau [B][N][n_motors], lambda [B][N][n_constraints], grf [B][N][n_feet][5] = (z, +x, +y, -x, -y) per foot; any may be NULL (= 0).
 * residual [B][N][nq].  Device pointers. */
/* `slack_eom` of make_pyomo_model(include_eom_slack=True) (acinoset_opt.py:510-514; cost misc.eom_slack_cost, acinoset_misc.py:631) */
cpe_status cpe_eom_residual(cpe_handle* h, const cpe_dyn_options* opt, int32_t B, int32_t N, const double* q, const double* dq,
                            const double* ddq, const double* tau, const double* lambda, const double* grf, double* residual);

/* ---- physics-based trajectory model (config 4: estimate_kinetics, acinoset_opt.py:693-963) -------------------
 * The reference adds to the kinematic NLP, per node: 22 joint torques tau, 26 joint constraint forces lambda (`Fr`), four feet x
 * (GRFz + 4 non-negative friction components), 54 slack_eom, and the equality
 *     rows(q, q', q'') = B_grf + B_tau + (dc/dq)^T lambda + slack            (forces in body weights M g, cpe_eom_residual above)
 * with the cost (acinoset_opt.py:905-921)
 *     1e-3 ( measurement + GMM pose + sum tau^2 + 0.1 fps^-2 sum (fps^2 (pose[n+2] - 2 pose[n+1] + pose[n]))^2 + 10e3 sum slack^2 )
 * -- no constant-acceleration term, no autoregressive prior -- and the contact schedule of prescribe_contact_order
 * (acinoset_misc.py:1140-1167) + the no-slip rule (acinoset_opt.py:783-806): outside stance the foot force is fixed to 0; in stance
 * GRFz >= grfz_min, |foot height| <= foot_height_tol, foot xy speed components <= slip_max; always 0 <= GRF <= force_max and
 * friction * GRFz >= sum_k GRFxy_k.
 *
 * Statement solved here: slack is eliminated (it IS the residual e_n of the equality), the per-node forces
 * f_n = (tau, lambda, net foot force (z, x, y) of every stance foot) enter e_n linearly and are minimised out exactly per node
 * (a convex piecewise-quadratic problem, semismooth Newton), and the trajectory is found by the same Levenberg-Marquardt /
 * block-banded Cholesky as cpe_solve on  V_n(u_n, u_n-1, u_n-2) = min_f [ w_slack |e_n|^2 + w_torque |tau|^2 + ... ]  (n >= 2: q'_0
 * and q''_0 are free in the implicit-Euler collocation, so nodes 0 and 1 carry no dynamics), whose Gauss-Newton blocks are the
 * Schur complements of the force unknowns.  The reference's four non-negative friction components enter only through their
 * differences; the cost-neutral face is resolved by x+ x- = 0 (the minimum-norm point), i.e. net forces with |Fx| + |Fy| <= mu Fz.
 * Inequalities: augmented Lagrangian (multipliers updated with the angle bounds'), including the box [slack_lo, slack_hi] on every
 * component of the residual (the reference's bound_eom_error): inside the node solve its active set is iterated to a fixed point around the
 * exact elimination, like the torque boxes'.  motion_w of the handle's skeleton must be all zero.
 * Versus the reference's `.robot` equations of motion this model is "parity unpinned" (SURVEY 8c-8): the pins are the numerical
 * Lagrangian and virtual work (tests/test_grf.py). */
#define CPE_MAX_MOTORS 32
#define CPE_KIN_MAXLAT (CPE_MAX_MOTORS + 2 * CPE_MAX_JOINTS + 12)    /* latent forces per node: tau | lambda | (z, x, y) per foot */
typedef struct cpe_kinetic_options {
    cpe_dyn_options dyn;      /* gravity, link inertias, feet (markers at the hock bottoms), motors                              */
    double w_slack;           /* 1e4 = `10e3 * slack_cost`            (acinoset_opt.py:921)                                      */
    double w_torque;          /* 1                                    (torque_squared_penalty, :915)                             */
    double w_smooth;          /* 0.1 / fps^2                          (:919-920), on |fps^2 * second difference of the markers|^2 */
    double friction;          /* 0.8                                  (:506)                                                     */
    double force_max;         /* 5                                    (:494-497)                                                 */
    double grfz_min;          /* 0.01                                 (acinoset_misc.py:1144)                                    */
    double foot_height_tol;   /* 0.1; 0.03 for the kinetic dataset    (acinoset_opt.py:780)                                      */
    double foot_height_min;   /* 0: lower bound of `foot_height` outside stance (variable bound inside the absent pe.foot: unpinned); <= -1e9 disables */
    double ground_height;     /* Foot3D.ground_plane_height           (:500)                                                     */
    double slip_max;          /* 1: `gamma <= 1` in stance            (:803-806); <= 0 disables                                  */
    double zvel_max;          /* 1 on the kinetic dataset, else off: `foot_z_vel <= 1` in stance (:807-810, :864-866), read as |vertical foot speed| <= zvel_max
                               * (the variable lives in the absent pe.foot: unpinned); <= 0 disables                             */
    double slack_lo, slack_hi;/* bound_eom_error: box on every slack_eom component, ENFORCED as augmented-Lagrangian rows on the residual:
                               * (-2, 2) run_dataset.py:984, :1211; (-0.1, 0.1) in the last stage of run_kinetic, :1136; lo <= -1e9 and hi >= 1e9: no box */
    double reg_force;         /* Tikhonov weight on lambda and the foot forces (1e-4): picks the minimum-norm point of a face the reference leaves open */
    double kappa_force, kappa_height, kappa_slip;     /* augmented-Lagrangian penalties (1e5, 1e6, 1e2; kappa_slip also serves zvel_max)  */
    double kappa_slack;       /* penalty of the slack box (1e6 = 100 w_slack)                                                    */
    double lm_force_damping;  /* the node forces are eliminated from (H_ff + lambda * lm_force_damping * diag(H_ff) + walls): the trust region
                               * also acts in FORCE space, where the walls of this problem (force bounds, friction polyhedron) are */
    double lm_wall_damping;   /* walls: + lambda * lm_wall_damping * 2 w_slack * sum over the INACTIVE force inequalities c c^T / gap^2
                               * (the Hessian of a log barrier, affine-scaling trust region): forces near a bound are held back in the model */
    int32_t inner_iterations; /* cap on the per-node Newton iterations for the forces (30)                                       */
    int32_t _pad;
} cpe_kinetic_options;

typedef struct cpe_kinetic_stats {
    double cost_torque;       /* sum tau^2                      estimator.costs["torque"]    (acinoset_opt.py:922-928)           */
    double cost_energy;       /* sum (fps^2 second difference)^2               ["energy"]                                        */
    double cost_eom;          /* sum slack^2                                   ["eom_error"]                                     */
    double max_slack;         /* largest |slack_eom| component (body weights), to hold against [slack_lo, slack_hi]              */
    double max_base_rows;     /* largest |rows 0-2 - forces| / (M g): the reference's stored solutions have <= 8e-5 */
    double max_violation;     /* largest violated force / height / slip inequality at the solution                               */
    int32_t inner_max;        /* largest Newton iteration count of a node in the last evaluation                                 */
    int32_t _pad;
} cpe_kinetic_stats;

/* the reference's values for a given frame rate and data set (kinetic_dataset: foot_height_tol 0.03, zvel_max 1) */
void cpe_default_kinetic_options(cpe_kinetic_options* o, double fps, int32_t kinetic_dataset);

/* Device pointers.  q_init [B][N][nq] = the kinematic solution (init_prev_kinematic_solution, acinoset_opt.py:739-777);
 * stance int32 [B][N][n_feet] (1 = the foot is inside a contact window of autogen-contact.json, :783-812);
 * outputs as cpe_solve plus tau [B][N][n_motors], lambda [B][N][n_constraints], grf [B][N][n_feet][5] = (z, +x, +y, -x, -y),
 * slack [B][N][nq] (all zero at nodes 0 and 1); stats / kstats [B] are HOST pointers (kstats may be NULL).
 * cpe_stats.cost_model carries w_torque * torque + w_smooth * energy + w_slack * eom (the reference's "motion prior + slack" part),
 * cpe_stats.cost the whole objective times cost_scale. */
cpe_status cpe_solve_kinetic(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q_init, const double* meas,
                             const double* weight, const int32_t* stance, double* q, double* dq, double* ddq, double* positions,
                             double* meas_err, double* tau, double* lambda, double* grf, double* slack, cpe_stats* stats,
                             cpe_kinetic_stats* kstats);

/* the same with PRESCRIBED foot forces (estimate_kinetics(joint_estimation=False, fix_grf=True), acinoset_opt.py:813-838: GRFz / GRFxy fixed to a
 * synthesised or previously fitted profile): grf_fixed [B][N][n_feet][3] = net force (z, x, y) of every foot in body weights, device pointer,
 * read where stance != 0 (elsewhere the force is zero as before).  The foot forces stop being unknowns of the node -- only torques and
 * joint constraint forces are eliminated -- and the friction / positivity rows drop out, as the reference removes its friction constraint for fixed
 * forces; the foot-height and no-slip rules of the stance frames stay.  grf_fixed == NULL is cpe_solve_kinetic. */
cpe_status cpe_solve_kinetic_fixed(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q_init, const double* meas,
                                   const double* weight, const int32_t* stance, const double* grf_fixed, double* q, double* dq, double* ddq,
                                   double* positions, double* meas_err, double* tau, double* lambda, double* grf, double* slack, cpe_stats* stats,
                                   cpe_kinetic_stats* kstats);

/* the same with every TORQUE BOXED: the reference's module-level estimate_grf (acinoset_opt.py:966-1048, called by run_dataset.py:1138 as the last
 * stage of the kinetic-dataset pipeline) re-solves the physics-based model from the previous solve with `Tc.bounds = bound_value(init_tau, 0.1)` (:995-1003),
 * the foot forces free where the measured force plates saw a contact and zero elsewhere.  tau_box [B][N][n_motors][2] = (lower, upper), device pointer.
 * The boxes are augmented-Lagrangian rows of the node (penalty kappa_force, multipliers updated with the others); inside the node solve their active
 * set is iterated to a fixed point around the exact elimination of torques and constraint forces.  `stance` carries the measured contact pattern. */
cpe_status cpe_solve_kinetic_bounded(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q_init, const double* meas,
                                     const double* weight, const int32_t* stance, const double* tau_box, double* q, double* dq, double* ddq,
                                     double* positions, double* meas_err, double* tau, double* lambda, double* grf, double* slack, cpe_stats* stats,
                                     cpe_kinetic_stats* kstats);

/* the same with the net force of every stance foot BOXED (estimate_kinetics(joint_estimation=False, fix_grf=False), acinoset_opt.py:838-850: `GRFz` and
 * the `GRFxy` sides unfixed within `bound_value(profile, 0.2)`): grf_box [B][N][n_feet][3][2] = (lower, upper) of the net (z, x, y) force in body
 * weights, device pointer, read where stance != 0.  The boxes replace the constants of the positivity / force_max rows of the foot; the friction
 * polyhedron stays, as in the reference (it removes that constraint only when the forces are fixed). */
cpe_status cpe_solve_kinetic_force_box(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q_init, const double* meas,
                                       const double* weight, const int32_t* stance, const double* grf_box, double* q, double* dq, double* ddq,
                                       double* positions, double* meas_err, double* tau, double* lambda, double* grf, double* slack, cpe_stats* stats,
                                       cpe_kinetic_stats* kstats);

/* diagnostic building block of cpe_solve_kinetic (as cpe_eval_normal is of cpe_solve): ONE evaluation of the physics terms of every node at
 * Euler q, multipliers zero, forces from a cold start -- what ASL hands IPOPT per node for the constraints of make_pyomo_model(include_eom_slack=True)
 * (acinoset_opt.py:510-514) after the node forces are minimised out.  Device pointers, each may be NULL: f [B][N][64] node forces (tau | lambda |
 * (z, x, y) per foot), stat [B][N][8] = (sum slack^2, sum tau^2, regularised norm, smoothing energy, multiplier terms, max |slack|, max |base rows|,
 * max violation), g [B][N][84] gradient with respect to the coordinates of frames n, n-1, n-2, Huu [B][N][84][84], Hfu [B][N][64][84], Hff [B][N][64][64]
 * (rows = free node forces in meta's order), meta int32 [B][N][65] = (count, indices). */
cpe_status cpe_eval_kinetic_nodes(cpe_handle* h, const cpe_kinetic_options* opt, int32_t B, int32_t N, const double* q, const double* meas,
                                  const double* weight, const int32_t* stance, double* f, double* stat, double* g, double* Huu, double* Hfu,
                                  double* Hff, int32_t* meta);

/* forward kinematics only (get_pose_state / get_com, acinoset_misc.py:1581-1659, :722-742); device ptrs */
cpe_status cpe_forward_kinematics(cpe_handle* h, int32_t B, int32_t N, const double* q,
                                  double* positions /*[B][N][L][3]*/, double* com /*[B][N][3] or NULL*/);

/* marker velocities v_l = (d p_l / d q) dq (the lambdified `foot.Pb_I_vel` of the contact heuristic, acinoset_misc.py:347-360,
 * for every marker); device ptrs */
cpe_status cpe_marker_velocities(cpe_handle* h, int32_t B, int32_t N, const double* q, const double* dq,
                                 double* velocities /*[B][N][L][3]*/);

/* reprojection of 3D marker positions into every camera of the handle, for the cam*_fte.{h5,csv} writers
 * (acinoset_misc.py:1339-1407: cv.fisheye.projectPoints / pt3d_to_2d per camera); device ptrs */
cpe_status cpe_reproject(cpe_handle* h, int32_t B, int32_t N, const double* positions /*[B][N][L][3]*/,
                         double* uv /*[B][N][C][L][2]*/);

/* initial-guess ingestion: two-view linear triangulation of n detection pairs, i.e. triangulate_points[_fisheye]
 * (acinoset_misc.py:1432-1453: cv[.fisheye].undistortPoints of both pixels with the cameras' K and D, cv.triangulatePoints with
 * [R | t]).  cam_a / cam_b index the handle's cameras; cam_b[i] < 0 asks for the monocular rule instead: the first detection
 * back-projected to `depth` metres along its ray.  Device ptrs: cam_a, cam_b int32[n]; uv_a, uv_b [n][2]; xyz [n][3]. */
cpe_status cpe_triangulate(cpe_handle* h, int32_t n, const int32_t* cam_a, const int32_t* cam_b, const double* uv_a, const double* uv_b,
                           double depth, double* xyz);

/* measurement ingestion: the DeepLabCut table of ONE camera -> that camera's slice `slot` of meas [N][n_slots][L][2]
 * and weight [N][n_slots][L], as init_measurements / init_meas_weights fill the Pyomo params (acinoset_misc.py:211-256): frame n
 * reads table row first_row + n (first_row = start_frame - sync_offset of the camera), marker l reads body part part_of_marker[l];
 * weight = inv_sigma[l] if likelihood > thresh else 0.  Rows outside the table and non-finite coordinates give meas 0, weight 0.
 * table [rows][3*parts] = (x, y, likelihood) per body part; L is the handle's marker count.  Device ptrs. */
cpe_status cpe_tensorise_dlc(cpe_handle* h, int32_t N, int32_t n_slots, int32_t slot, const double* table, int32_t rows, int32_t parts,
                             int32_t first_row, const int32_t* part_of_marker, const double* inv_sigma, double thresh, double* meas, double* weight);

#ifdef __cplusplus
}
#endif
#endif /* CPE_H */
