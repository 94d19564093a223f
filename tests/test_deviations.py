"""the three documented deviations of the restated NLP, each bounded by a number on the reference's own stored data or on
the benchmark gallop (DESIGN.md 2 / 2b carry the figures).  CPU only: oracle + numpy."""
import os

import numpy as np

from cheetah_pose_estimation_amd import abi, skeleton, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _principal(q):
    """the principal ZYX triple (|theta| <= pi / 2) of every link of q [N, nq], yaw unwrapped along the sequence: what this build's coordinate map
    reads back for the leg links; the same rotations, so the same markers"""
    q = q.copy()
    nl = (q.shape[1] - 3) // 3
    for i in range(nl):
        a = q[:, 3 + 3 * i:6 + 3 * i]
        R = synth.rot_zyx(a)
        th = np.arcsin(np.clip(-R[:, 2, 0], -1, 1)); ph = np.arctan2(R[:, 2, 1], R[:, 2, 2]); ps = np.arctan2(R[:, 1, 0], R[:, 0, 0])
        ps = ps + 2 * np.pi * np.round((a[:, 2] - ps) / (2 * np.pi))          # same 2 pi sheet as the stored yaw where the triples agree
        q[:, 3 + 3 * i] = ph; q[:, 4 + 3 * i] = th; q[:, 5 + 3 * i] = ps
    return q


def _unwrap_time(q):
    q = q.copy()
    q[:, 3:] = np.unwrap(q[:, 3:], axis=0)
    return q


def _continuous(q, w, h):
    """the path an optimiser with the Euler angles as its variables ends on: per link one of the two equivalent triples per frame -- (phi, theta, psi)
    or (phi + pi, pi - theta, psi + pi), multiples of 2 pi by continuity -- such that the link's own constant-acceleration cost (weights w of its three
    angles) is smallest over ALL 2^N assignments: dynamic programme over the last three choices"""
    q = q.copy()
    N = len(q)
    for i in range((q.shape[1] - 3) // 3):
        a = q[:, 3 + 3 * i:6 + 3 * i]
        wi = w[3 + 3 * i:6 + 3 * i]
        if not wi.any():
            continue
        cand = np.stack([a, np.stack([a[:, 0] + np.pi, np.pi - a[:, 1], a[:, 2] + np.pi], 1)], 1)      # [N, 2, 3]

        def near(x, ref):
            return x + 2 * np.pi * np.round((ref - x) / (2 * np.pi))
        # states: choices of frames (n-2, n-1, n); value: best cost and the unwrapped triples of those three frames
        best = {}
        for c0 in (0, 1):
            for c1 in (0, 1):
                for c2 in (0, 1):
                    t0 = cand[0, c0]; t1 = near(cand[1, c1], t0); t2 = near(cand[2, c2], t1)
                    best[(c0, c1, c2)] = (0.0, [t0, t1, t2], [c0, c1, c2])
        for n in range(3, N):
            new = {}
            for (c0, c1, c2), (cost, tr, hist) in best.items():
                for c3 in (0, 1):
                    t3 = near(cand[n, c3], tr[-1])
                    e = (t3 - 3 * tr[-1] + 3 * tr[-2] - tr[-3]) / h ** 2
                    cc = cost + float((wi * e * e).sum())
                    key = (c1, c2, c3)
                    if key not in new or cc < new[key][0]:
                        new[key] = (cc, tr + [t3], hist + [c3])
            best = new
        _, tr, _ = min(best.values(), key=lambda v: v[0])
        q[:, 3 + 3 * i:6 + 3 * i] = np.array(tr)
    return q


def test_principal_triple_versus_the_reference_path_on_the_stored_run():
    """Deviation (a): one rotation has two ZYX triples and both satisfy the reference's joint equalities; its optimiser's variables followed one
    continuous path (in the stored AcinoSet run 2019_03_07/phantom/run 29 of the 798 leg-link states sit on the cos(phi) < 0 side), this build always
    reads back the principal triple (|theta| <= 90 deg).  The markers are identical.  The terms of the objective that act on the Euler angles
    themselves are not; evaluated here on the stored run along BOTH paths --
      * constant-acceleration cost sum_p ((third difference of q_p) / h^2)^2 / Q_p^2 (acinoset_misc.py:639-677): along the continuous path it is the
        reference's own number (236.80 from the `ddq` of its fte.pickle, SURVEY 8c-3: a pin of the recovered angles AND of the Q table); along the
        principal triple the pitch of a limb turns back at the horizontal instead of passing it;
      * Gaussian-mixture pose prior on the 22 relative angles (acinoset_misc.py:680-714).
    The figures are printed (DESIGN.md 2 quotes them); the assertions bound the effect."""
    from cheetah_pose_estimation_amd import priors as P
    Z = np.load(os.path.join(GOLD, "fk_csv_pin.npz"))
    sk = skeleton.build_skeleton("phantom", 24)
    leg_links = [sk.joint_child[j] for j in range(sk.n_joints) if sk.joint_kind[j] == abi.JOINT_REVOLUTE_Y]
    w = np.array(sk.motion_w[:sk.nq]); h = 1.0 / 120.0
    q_pri = _unwrap_time(_principal(Z["q"]))
    q_ref = _continuous(_unwrap_time(Z["q"]), w, h)
    pos_a, _ = synth.fk_numpy(sk, q_ref); pos_b, _ = synth.fk_numpy(sk, q_pri)
    assert np.abs(pos_a - pos_b).max() < 1e-12                # the same rotations
    beyond = np.array([np.cos(Z["q"][:, 3 + 3 * c]) < 0 for c in leg_links]).T       # states the recovered run holds on the cos(phi) < 0 side
    assert int(beyond.sum()) == 29
    def const_acc(q):
        e = (q[3:] - 3 * q[2:-1] + 3 * q[1:-2] - q[:-3]) / h ** 2
        return float((w * e * e).sum()), (w * e * e).sum(1)
    ca_ref, per_ref = const_acc(q_ref); ca_pri, per_pri = const_acc(q_pri)
    z = np.load(os.path.join(os.path.dirname(P.__file__), "data", "priors_full_pose.npz"))
    ind = skeleton.independent_dofs(sk)
    ref_idx = np.array(sk.rel_ref[:sk.nq]); sgn = np.array(sk.rel_sign[:sk.nq])

    def gmm(q):
        rel = np.where(ref_idx < 0, q, sgn * (q - q[:, np.maximum(ref_idx, 0)]))[:, ind][:, 6:]
        rel = rel - 2 * np.pi * np.round(rel / (2 * np.pi))
        lp = []
        for k in range(len(z["gmm_weights"])):
            d = rel - z["gmm_means"][k]
            Pm = np.linalg.inv(z["gmm_covariances"][k]); _, ld = np.linalg.slogdet(z["gmm_covariances"][k])
            lp.append(np.log(z["gmm_weights"][k]) - 0.5 * (rel.shape[1] * np.log(2 * np.pi) + ld) - 0.5 * np.einsum("ni,ij,nj->n", d, Pm, d))
        lp = np.array(lp); m = lp.max(0)
        return -np.log(np.exp(lp - m).sum(0) * np.exp(m) + 1e-12)
    g_ref, g_pri = gmm(q_ref), gmm(q_pri)
    fr = beyond.any(1)
    stored_objective = 33960.2                                # obj_cost / 1e-3 of the run's fte.pickle (33 723.39 measurement + 236.80 model)
    print(f"\nstored run, 57 frames, {int(beyond.sum())} leg-link states on the cos(phi) < 0 side in {int(fr.sum())} frames:")
    print(f"  constant-acceleration cost of the recovered angles: best assignment of triples {ca_ref:.2f}, principal triple {ca_pri:.2f} "
          f"(+{ca_pri - ca_ref:.2f} = {100 * (ca_pri - ca_ref) / stored_objective:.2f} % of the run's stored objective {stored_objective:.0f})")
    print(f"  GMM pose term, sum over frames: {g_ref.sum():.2f} vs {g_pri.sum():.2f} ({100 * (g_pri.sum() - g_ref.sum()) / g_ref.sum():+.1f} %)")
    # (the reference's own 236.80 for this term cannot be reproduced from the pixels: the recovered world frame is the animal's, not the scene's, and
    # the Euler angles of a tilted frame have other third differences)
    assert ca_pri >= ca_ref - 1e-6                            # the dynamic programme is the minimum over all assignments
    assert ca_pri - ca_ref < 0.05 * stored_objective          # the principal triple costs less than 5 % of the objective on this run (measured: 1.8 %)
    assert abs(g_pri.sum() - g_ref.sum()) < 0.15 * abs(g_ref.sum()) and np.isfinite(g_pri).all()


def test_grf_fit_reports_its_own_convergence(oracle):
    """Deviation (b): the per-frame ground-reaction-force fit (CheetahEstimator.estimate_grf, acinoset_opt.py:176-270) is a fixed number of FISTA
    iterations towards the minimum-norm point of a face of minimisers.  Whether 2 000 are enough is measured, not assumed: the objective of the
    last iterate against 4 000 and 8 000 iterations, on frames with one, two and four feet on the ground."""
    sk = skeleton.build_skeleton("phantom", 24)
    d = synth.make_gallop_batch(skeleton.without_motion_model(sk), synth.make_cameras(2), B=1, N=40, seed=4321)
    q = d["q_true"][0]
    dq, ddq = oracle.derivatives(q, 1.0 / 120.0)
    frames = [n for n in range(3, 40) if d["stance"][0][n].sum() >= 1][:6]
    worst = 0.0
    for n in frames:
        res, ys = [], []
        for it in (2000, 4000, 8000):
            g = skeleton.grf_options("phantom", iterations=it)
            gz, gxy, r = oracle.grf_fit(sk, g, q[n:n + 1], dq[n:n + 1], ddq[n:n + 1], d["stance"][0][n:n + 1])
            res.append(float((r ** 2).sum())); ys.append(np.concatenate([gz.ravel(), gxy.ravel()]))
        worst = max(worst, np.abs(ys[0] - ys[2]).max())
        assert abs(res[0] - res[2]) <= 1e-6 * max(res[2], 1e-12) + 1e-12, (n, res)      # the residual has converged after 2 000 iterations
        assert np.abs(ys[1] - ys[2]).max() <= np.abs(ys[0] - ys[2]).max() + 1e-12       # ... and the forces keep moving towards the minimum-norm point only
    print(f"\\nGRF fit: forces after 2 000 vs 8 000 FISTA iterations differ by at most {worst:.2e} body weights on {len(frames)} stance frames")
    assert worst < 5e-2                                      # measured 2.6e-2: the residual is converged, the split of the force along the open face still drifts


def test_tikhonov_weight_of_the_node_forces_is_inert(oracle):
    """Deviation (c): the physics-based model regularises the joint constraint forces and foot forces of a node with 1e-4 |f|^2 to pick one point of
    the face the reference leaves open.  Its influence on the TRAJECTORY: the same 40-frame gallop solved with 1e-4 and 1e-6."""
    sk = skeleton.without_motion_model(skeleton.build_skeleton("phantom", 24)); skf = skeleton.build_skeleton("phantom", 24)
    cams = synth.make_cameras(6)
    d = synth.make_gallop_batch(sk, cams, B=1, N=40, seed=4321)
    kin = oracle.solve(skf, cams, abi.default_options(120.0), None, d["q_init"][0], d["meas"][0], d["weight"][0])
    opts = abi.default_options(120.0); opts.tol_cost, opts.max_iter = 1e-8, 600
    out = {}
    for reg in (1e-4, 1e-6):
        ko = abi.default_kinetic_options(skeleton.dyn_options("phantom"), 120.0); ko.reg_force = reg
        out[reg] = oracle.solve_kinetic(sk, cams, opts, None, ko, kin["q"], d["meas"][0], d["weight"][0], d["stance"][0])
        assert out[reg]["status"] == abi.OK
    a, b = out[1e-4], out[1e-6]
    rmse = float(np.sqrt(((a["positions"] - b["positions"]) ** 2).sum(-1).mean()))
    dtau = float(np.abs(a["tau"] - b["tau"]).max()); dgrf = float(np.abs(a["grf"] - b["grf"]).max()); dlam = float(np.abs(a["lam"] - b["lam"]).max())
    print(f"\\nTikhonov 1e-4 vs 1e-6: marker RMSE {rmse:.2e} m, torques {dtau:.2e}, foot forces {dgrf:.2e}, constraint forces {dlam:.2e} (body weights); cost {a['stats'].cost:.6f} vs {b['stats'].cost:.6f}")
    assert rmse < 1e-3 and dtau < 0.05 and dgrf < 0.05


def test_cost_pitch_on_a_200_fps_gallop_of_the_kinetic_dataset():
    """Deviation (a) where it WAS large, and what replaced it.  kinetic_dataset/2009_09_07/arabia/trial06 (tests/golden/fk_csv_pin_arabia.npz, in the world
    frame the joint-angle bounds fix): a 200 fps gallop in which four leg links swing beyond the horizontal (30 of 400 leg-link states with cos(phi) < 0).
    The constant-acceleration cost of the stored solution -- third differences / h^2, weights 1 / Q^2 (acinoset_misc.py:639-677) --
      * along the triple whose roll stays next to the body's (pitch runs past +-90 degrees: the path the reference's variables take from their start
        at phi = theta = 0):                                                                 102.5
      * along the principal triple rounds 1-2 let the cost terms see (pitch turns around at +-90 degrees):   15 420.9, 99 % of it in those four links
        -- on such a trial the deviation WAS the model term, and the kinematic solve crept (no stop within 200 iterations from the stored solution);
      * along the COST PITCH the solver uses since round 3, theta_B + alpha_c (synth.cost_view_numpy):   105.4 -- the reference's variable
        up to O(roll^2) away from the pole, smooth through it.  (The body-relative triple itself was built and measured too: it jumps by ~2 x roll at
        the pole and the Levenberg-Marquardt loop parks on the jump, profiles/r03_notes.md.)"""
    Z = np.load(os.path.join(GOLD, "fk_csv_pin_arabia.npz"))
    sk = skeleton.build_skeleton("arabia-02", 24, kinetic_dataset=True)
    q = Z["q"]
    w = np.array(sk.motion_w[:sk.nq]); h = 1.0 / 200.0
    lay = synth.leg_layout(sk)

    def cost(qq):
        u = qq.copy(); u[:, 3:] = np.unwrap(u[:, 3:], axis=0)
        e = (u[3:] - 3 * u[2:-1] + 3 * u[1:-2] - u[:-3]) / h ** 2
        return (w * e * e).sum(0)
    # the same rotations with the body-relative triple for the leg links
    qb = q.copy()
    flips = 0
    for c, B in lay:
        ph, th, ps = q[:, 3 + 3 * c], q[:, 4 + 3 * c], q[:, 5 + 3 * c]
        flip = np.cos(ph - q[:, 3 + 3 * B]) < 0.0
        flips += int(flip.sum())
        ph2 = np.where(flip, ph + np.pi, ph); th2 = np.where(flip, np.where(th >= 0, np.pi, -np.pi) - th, th); ps2 = np.where(flip, ps + np.pi, ps)
        qb[:, 3 + 3 * c] = ph2 + 2 * np.pi * np.round((q[:, 3 + 3 * B] - ph2) / (2 * np.pi))
        qb[:, 4 + 3 * c] = th2
        qb[:, 5 + 3 * c] = ps2 + 2 * np.pi * np.round((q[:, 5 + 3 * B] - ps2) / (2 * np.pi))
    assert np.abs(synth.fk_numpy(sk, q)[0] - synth.fk_numpy(sk, qb)[0]).max() < 1e-12       # the same poses
    qv = synth.cost_view_numpy(sk, q)
    cp, cb, cv = cost(q), cost(qb), cost(qv)
    print(f"\narabia trial06, 200 fps: constant-acceleration cost along the principal triple {cp.sum():.1f}, along the body-relative triple {cb.sum():.1f}, "
          f"along the cost pitch {cv.sum():.1f}; {flips} leg-link states differ; largest pitch {np.abs(qb[:, 4::3]).max():.2f} rad; "
          f"cost pitch - body-relative pitch: max {np.abs(qv[:, 4::3] - qb[:, 4::3]).max():.3f} rad")
    assert flips == 30 and np.abs(qb[:, 4::3]).max() > 1.7
    assert cb.sum() < 150.0 and cp.sum() > 100.0 * cb.sum()
    assert np.sort(cp)[-4:].sum() > 0.99 * cp.sum()
    assert abs(cv.sum() - cb.sum()) < 0.05 * cb.sum()                                     # 105.4 against 102.5
    assert np.abs(qv[:, 4::3] - qb[:, 4::3]).max() < 0.12                                 # within ~roll of the reference's variable, at the pole
