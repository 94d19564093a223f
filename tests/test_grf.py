"""Per-frame ground-reaction-force fit (acinoset_opt.py:176-270).  CPU: the oracle's rows 0-5 of the
equations of motion against an independent numerical Lagrangian (numpy finite differences of T and V along a smooth
trajectory), properties of the fit.  The reference's lambdified rows live in dill pickles (`models/*_grf_eom`) that the
permitted loaders refuse and `pe.foot` is absent, so versus the reference this row is "parity unpinned"; the formula
follows SURVEY A.8, which reports 1e-13 agreement with those functions."""
import numpy as np
import pytest

from cheetah_pose_estimation_amd import abi, skeleton, synth


def _traj(sk, t):
    """smooth analytic trajectory q(t) (every angle of every link moves)"""
    k = np.arange(sk.nq)
    return 0.3 * np.sin(1.7 * t + 0.37 * k) + 0.1 * np.cos(3.1 * t + 0.11 * k * k) + np.where(k < 3, 2.0 * t, 0.0)


def _omega_body(a, da):
    sf, cf, st, ct = np.sin(a[0]), np.cos(a[0]), np.sin(a[1]), np.cos(a[1])
    return np.array([da[0] - st * da[2], cf * da[1] + sf * ct * da[2], -sf * da[1] + cf * ct * da[2]])


def _rot_c(ang):
    """rot_zyx for complex angles (complex-step differentiation)"""
    sf, cf, st, ct, sp, cp = np.sin(ang[0]), np.cos(ang[0]), np.sin(ang[1]), np.cos(ang[1]), np.sin(ang[2]), np.cos(ang[2])
    return np.array([[cp * ct, sf * st * cp - sp * cf, sf * sp + st * cf * cp],
                     [sp * ct, sf * sp * st + cf * cp, -sf * cp + sp * st * cf],
                     [-st, sf * ct, cf * ct]])


def _lagrangian(sk, gopt, q, dq, all_inertia=None):
    """L = sum_i (m_i |P_i'|^2 / 2 - m_i g P_i,z) + w_root^T I w_root / 2; the rotational energy of the other links does not
    depend on the root coordinates and drops out of rows 0-5.  P_i' = d/ds P_i(q + s dq) exactly, by the complex step."""
    def coms(qq):
        R = [_rot_c(qq[3 + 3 * i:6 + 3 * i]) for i in range(sk.n_links)]
        origin, out = [None] * sk.n_links, []
        for i in range(sk.n_links):
            origin[i] = qq[:3] if sk.parent[i] < 0 else origin[sk.parent[i]] + R[sk.parent[i]] @ np.array(sk.attach[i][:])
            out.append(origin[i] + R[i] @ np.array(sk.com[i][:]))
        return np.array(out)
    cs = coms(q + 1e-30j * dq)
    P, V = cs.real, cs.imag / 1e-30
    m = np.array(sk.mass[:sk.n_links])
    if all_inertia is not None:                                 # rotational energy of every link (needed for the rows of all angles)
        rot = sum(0.5 * (np.array(all_inertia[i][:]) * _omega_body(q[3 + 3 * i:6 + 3 * i], dq[3 + 3 * i:6 + 3 * i]) ** 2).sum()
                  for i in range(sk.n_links))
    else:
        w = _omega_body(q[3:6], dq[3:6])
        rot = 0.5 * (np.array(gopt.root_inertia[:]) * w * w).sum()
    return 0.5 * (m * (V ** 2).sum(1)).sum() - gopt.gravity * (m * P[:, 2]).sum() + rot


def test_eom_rows_match_a_numerical_lagrangian(oracle):
    sk = skeleton.build_skeleton("phantom", 24)
    gopt = skeleton.grf_options("phantom")
    M = sum(sk.mass[:sk.n_links])
    ht = 1e-4
    for t0 in (0.3, 1.1):
        qf = lambda t: _traj(sk, t)
        q = qf(t0)
        dq = (qf(t0 + ht) - qf(t0 - ht)) / (2 * ht)
        ddq = (qf(t0 + ht) - 2 * q + qf(t0 - ht)) / ht ** 2
        E, A = oracle.grf_terms(sk, gopt, q, dq, ddq)

        def dL_ddq(t, a):       # dL/dq'_a at time t
            qq = qf(t); dd = (qf(t + ht) - qf(t - ht)) / (2 * ht)
            e = np.zeros(sk.nq); e[a] = 1.0                      # L is quadratic in q': the central difference is exact
            return (_lagrangian(sk, gopt, qq, dd + e) - _lagrangian(sk, gopt, qq, dd - e)) / 2.0

        for a in range(6):
            ddt = (dL_ddq(t0 + 1e-3, a) - dL_ddq(t0 - 1e-3, a)) / 2e-3
            e = np.zeros(sk.nq); e[a] = 1e-6
            dLdq = (_lagrangian(sk, gopt, q + e, dq) - _lagrangian(sk, gopt, q - e, dq)) / 2e-6
            ref = (ddt - dLdq) / (M * gopt.gravity)
            assert abs(E[a] - ref) < 2e-6 * max(1.0, abs(ref)), (a, E[a], ref)
    # the force matrix: d foot / d(root coordinates), by finite differences of the marker positions
    pos0 = oracle.markers(sk, q)
    for a in range(6):
        e = np.zeros(sk.nq); e[a] = 1e-6
        dp = (oracle.markers(sk, q + e) - oracle.markers(sk, q - e)) / 2e-6
        D = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0.0]])
        for f in range(4):
            for k in range(5):
                assert abs(A[a, 5 * f + k] - dp[gopt.foot_marker[f]] @ D[k]) < 1e-8


def test_standing_still_carries_the_body_weight(oracle):
    """at rest with all four feet down the vertical forces sum to one body weight, friction vanishes, moments balance"""
    sk = skeleton.build_skeleton("phantom", 24)
    gopt = skeleton.grf_options("phantom", iterations=4000)
    d = synth.make_batch(sk, synth.make_cameras(1), B=1, N=1, seed=3)
    q = d["q_true"][0]
    z = np.zeros_like(q)
    grfz, grfxy, res = oracle.grf_fit(sk, gopt, q, z, z, np.ones((1, 4), np.int32))
    assert abs(grfz.sum() - 1.0) < 1e-4 and np.abs(res).max() < 1e-4
    assert grfxy.max() < 0.05 and grfz.min() >= 0                     # the minimum-norm tie-break spreads a little friction
    net = grfxy[0].sum(0)
    assert abs(net[0] - net[2]) < 1e-4 and abs(net[1] - net[3]) < 1e-4   # no net horizontal force at rest
    # flight phase: no contact -> no force, the residual is the unbalanced weight
    grfz, grfxy, res = oracle.grf_fit(sk, gopt, q, z, z, np.zeros((1, 4), np.int32))
    assert grfz.max() == 0 and abs(res[0, 2] - 1.0) < 1e-12
    # one foot cannot balance the moments: constraints hold, residual stays
    c = np.zeros((1, 4), np.int32); c[0, 2] = 1
    grfz, grfxy, res = oracle.grf_fit(sk, gopt, q, z, z, c)
    assert (grfxy[0, 2].sum() <= 1.3 * grfz[0, 2] + 1e-9) and grfz[0, [0, 1, 3]].max() == 0 and 0 <= grfz[0, 2] <= 5


def test_free_fall_needs_no_force(oracle):
    """a rigid, non-rotating body in free fall: every row of the equations of motion vanishes (sign and scale of gravity),
    and a body held still needs exactly one body weight along z"""
    sk = skeleton.build_skeleton("phantom", 24)
    gopt = skeleton.grf_options("phantom")
    d = synth.make_batch(sk, synth.make_cameras(1), B=1, N=1, seed=9)
    q = d["q_true"][0, 0]
    dq = np.zeros_like(q); dq[0] = 3.0; dq[2] = -1.5                       # translating, not rotating
    ddq = np.zeros_like(q); ddq[2] = -gopt.gravity
    E, _ = oracle.grf_terms(sk, gopt, q, dq, ddq)
    assert np.abs(E).max() < 1e-12
    E, _ = oracle.grf_terms(sk, gopt, q, np.zeros_like(q), np.zeros_like(q))
    assert abs(E[2] - 1.0) < 1e-12 and np.abs(E[:2]).max() < 1e-12


def test_all_eom_rows_match_a_numerical_lagrangian(oracle):
    """cpo_eom_rows: all 54 rows of d/dt dL/dq' - dL/dq (the residual function of the physics-based model)
    against finite differences of the Lagrangian with every link's rotational energy"""
    sk = skeleton.build_skeleton("phantom", 24)
    eopt = skeleton.eom_options("phantom")
    gopt = skeleton.grf_options("phantom")
    assert np.allclose(eopt.link_inertia[0][:], gopt.root_inertia[:])
    ht, t0 = 1e-4, 0.7
    qf = lambda t: _traj(sk, t)
    q = qf(t0)
    dq = (qf(t0 + ht) - qf(t0 - ht)) / (2 * ht)
    ddq = (qf(t0 + ht) - 2 * q + qf(t0 - ht)) / ht ** 2
    E = oracle.eom_rows(sk, eopt, q, dq, ddq)
    L = lambda qq, dd: _lagrangian(sk, gopt, qq, dd, eopt.link_inertia)

    def dL_ddq(t, a):
        qq = qf(t); dd = (qf(t + ht) - qf(t - ht)) / (2 * ht)
        e = np.zeros(sk.nq); e[a] = 1.0
        return (L(qq, dd + e) - L(qq, dd - e)) / 2.0

    scale = sum(sk.mass[:sk.n_links]) * 9.81
    for a in range(sk.nq):
        ddt = (dL_ddq(t0 + 1e-3, a) - dL_ddq(t0 - 1e-3, a)) / 2e-3
        e = np.zeros(sk.nq); e[a] = 1e-6
        ref = ddt - (L(q + e, dq) - L(q - e, dq)) / 2e-6
        assert abs(E[a] - ref) < 2e-6 * scale, (a, E[a], ref)
    # the first six rows are the ones the GRF fit uses (in units of M g)
    E6, _ = oracle.grf_terms(sk, gopt, q, dq, ddq)
    assert np.abs(E[:6] / scale - E6).max() < 1e-12


def test_generalised_forces_do_the_right_virtual_work(oracle):
    """cpo_dyn_forces: for any q' the power of the generalised forces equals the power of the physical ones --
    foot forces times foot velocities, motor torques times the relative angular velocity of the two links about the motor's
    axis, and constraint forces through d c / dt."""
    sk = skeleton.build_skeleton("phantom", 24)
    dopt = skeleton.dyn_options("phantom")
    assert dopt.n_motors == 22
    rng = np.random.default_rng(11)
    q = _traj(sk, 0.4)
    Mg = sum(sk.mass[:sk.n_links]) * dopt.eom.gravity
    D = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0.0]])
    h = 1e-6
    for _ in range(3):
        dq = rng.normal(0, 1, sk.nq)
        # feet
        grf = rng.uniform(0, 2, (4, 5))
        Q = oracle.dyn_forces(sk, dopt, q, grf=grf)
        vel = (oracle.markers(sk, q + h * dq) - oracle.markers(sk, q - h * dq)) / (2 * h)
        power = sum(Mg * (grf[f] @ D) @ vel[dopt.foot_marker[f]] for f in range(4))
        assert abs(Q @ dq - power) < 1e-6 * max(1.0, abs(power))
        # motors: T . (w_second - w_first) with w from the rotation matrices, w_world = vee(R' R^T)
        tau = rng.normal(0, 1, 22)
        Q = oracle.dyn_forces(sk, dopt, q, tau=tau)
        Rm = lambda qq: synth.rot_zyx(qq[3:].reshape(sk.n_links, 3))
        Rd = (Rm(q + h * dq) - Rm(q - h * dq)) / (2 * h)
        W = np.einsum("lij,lkj->lik", Rd, Rm(q))                                   # R' R^T = [w]_x
        w = np.stack([W[:, 2, 1], W[:, 0, 2], W[:, 1, 0]], axis=1)
        power = 0.0
        for m in range(22):
            a1, a2, ax = dopt.motor_first[m], dopt.motor_second[m], dopt.motor_axis[m]
            power += Mg * tau[m] * Rm(q)[a1][:, ax] @ (w[a2] - w[a1])
        assert abs(Q @ dq - power) < 1e-6 * max(1.0, abs(power))
        # constraint forces: lambda . dc/dt
        lam = rng.normal(0, 1, 26)
        Q = oracle.dyn_forces(sk, dopt, q, lam=lam)
        cdot = (np.array(oracle.constraints(sk, q + h * dq)) - np.array(oracle.constraints(sk, q - h * dq))) / (2 * h)
        assert abs(Q @ dq - lam @ cdot) < 1e-6 * max(1.0, abs(lam @ cdot))
    # residual = rows - Q, linear in the forces
    dq = rng.normal(0, 1, sk.nq); ddq = rng.normal(0, 5, sk.nq)
    r0 = oracle.eom_residual(sk, dopt, q, dq, ddq)
    assert np.abs(r0 - oracle.eom_rows(sk, dopt.eom, q, dq, ddq)).max() == 0
    r1 = oracle.eom_residual(sk, dopt, q, dq, ddq, tau=tau, lam=lam, grf=grf)
    Qs = oracle.dyn_forces(sk, dopt, q, tau=tau) + oracle.dyn_forces(sk, dopt, q, lam=lam) + oracle.dyn_forces(sk, dopt, q, grf=grf)
    assert np.abs(r1 - (r0 - Qs)).max() < 1e-9 * Mg
