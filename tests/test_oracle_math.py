"""Self-consistency of the CPU oracle: analytic derivatives against finite differences, closed-form
joint projection against the joint equalities, reduced gradient against finite differences of the
objective, band Cholesky against dense linear algebra.  Runs without a GPU."""
import numpy as np
import pytest

from cheetah_pose_estimation_amd import abi, skeleton, synth


def test_rotation_convention(oracle):
    # R = Rz(psi) Ry(theta) Rx(phi)  
    a = np.array([0.3, -0.5, 1.1])
    cf, sf, ct, st, cp, sp = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
    Rx = np.array([[1, 0, 0], [0, cf, -sf], [0, sf, cf]]); Ry = np.array([[ct, 0, st], [0, 1, 0], [-st, 0, ct]])
    Rz = np.array([[cp, -sp, 0], [sp, cp, 0], [0, 0, 1]])
    assert np.abs(oracle.rot(a) - Rz @ Ry @ Rx).max() < 1e-15
    assert np.abs(synth.rot_zyx(a) - Rz @ Ry @ Rx).max() < 1e-15
    dR = oracle.drot(a)
    for j in range(3):
        e = np.zeros(3); e[j] = 1e-6
        fd = (oracle.rot(a + e) - oracle.rot(a - e)) / 2e-6
        assert np.abs(fd - dR[j]).max() < 1e-9


def test_marker_jacobian_fd(oracle, sk25):
    rng = np.random.default_rng(0)
    q = rng.normal(0, 0.6, sk25.nq)
    pos, dpos = oracle.markers_jac(sk25, q)
    assert np.abs(pos - oracle.markers(sk25, q)).max() < 1e-15
    for p in range(sk25.nq):
        e = np.zeros(sk25.nq); e[p] = 1e-6
        fd = (oracle.markers(sk25, q + e) - oracle.markers(sk25, q - e)) / 2e-6
        assert np.abs(fd - dpos[:, :, p]).max() < 1e-8
    # three independent FK statements agree (C oracle, numpy host generator)
    pn, cn = synth.fk_numpy(sk25, q)
    assert np.abs(pn - pos).max() < 1e-14 and np.abs(cn - oracle.com(sk25, q)).max() < 1e-14


@pytest.mark.parametrize("model", [abi.CAM_FISHEYE, abi.CAM_PINHOLE])
def test_projection_derivative_fd(oracle, cams6, model):
    cam = abi.Camera.from_buffer_copy(cams6[1])
    cam.model = model
    if model == abi.CAM_PINHOLE:
        cam.D[0], cam.D[1], cam.D[2] = -0.1, 0.03, -0.004
    p = np.array([9.0, 0.3, 0.6])
    uv, Gm = oracle.project(cam, p, want_G=True)
    for k in range(3):
        e = np.zeros(3); e[k] = 1e-6
        fd = (oracle.project(cam, p + e) - oracle.project(cam, p - e)) / 2e-6
        assert np.abs(fd - Gm[:, k]).max() < 1e-5 * max(1, np.abs(Gm).max())


def test_loss_derivatives_fd(oracle):
    for s in [-25.0, -12.0, -4.0, -0.3, 0.2, 1.0, 2.9, 3.1, 7.0, 10.5, 15.0, 19.0, 22.0, 60.0]:
        L = oracle.loss(s)
        h = 1e-5
        d1 = (oracle.loss(s + h)[0] - oracle.loss(s - h)[0]) / (2 * h)
        d2 = (oracle.loss(s + h)[1] - oracle.loss(s - h)[1]) / (2 * h)
        assert abs(d1 - L[1]) < 1e-7 and abs(d2 - L[2]) < 1e-6


def test_joint_constraints_and_projection(oracle, sk25):
    rng = np.random.default_rng(1)
    q = rng.normal(0, 0.3, (20, sk25.nq))
    q[:, 3 + 2::3] += np.pi                                     # psi around pi as in a real run
    qp, clamped = oracle.project_dependents(sk25, q, return_clamped=True)
    assert not clamped.any()
    # a limb pitched to 90 degrees under a rolled body has NO solution (gimbal band): must be flagged
    qbad = q[0].copy(); qbad[3] = 0.4; qbad[3 + 3 * 5 + 1] = np.pi / 2 - 0.05
    assert oracle.project_dependents(sk25, qbad, return_clamped=True)[1]
    indep = skeleton.independent_dofs(sk25)
    assert np.array_equal(qp[:, indep], q[:, indep])            # only the 26 dependent angles move
    for x in qp:
        assert np.abs(oracle.constraints(sk25, x)).max() < 1e-14  # all 26 equalities hold
    assert np.abs(synth.project_dependents_numpy(sk25, q) - qp).max() < 1e-13
    # constraint Jacobian
    c, Cq = oracle.constraints(sk25, qp[0], want_jac=True)
    assert c.shape == (26,)
    for p in range(sk25.nq):
        e = np.zeros(sk25.nq); e[p] = 1e-6
        fd = (oracle.constraints(sk25, qp[0] + e) - oracle.constraints(sk25, qp[0] - e)) / 2e-6
        assert np.abs(fd - Cq[:, p]).max() < 1e-8
    # tangent basis: moving along Z keeps the equalities to second order, and equals d(project)/du
    Z = oracle.tangent_basis(sk25, qp[0])
    assert np.abs(Cq @ Z).max() < 1e-12
    for k in range(abi.NX):
        e = np.zeros(sk25.nq); e[indep[k]] = 1e-6
        fd = (oracle.project_dependents(sk25, qp[0] + e) - oracle.project_dependents(sk25, qp[0] - e)) / 2e-6
        assert np.abs(fd - Z[:, k]).max() < 1e-7


def test_reduced_gradient_fd_and_band_matrix(oracle, sk25, cams6):
    d = synth.make_batch(sk25, cams6, B=1, N=8, seed=9)
    opts = abi.default_options()
    q0 = d["q_true"][0] + np.random.default_rng(3).normal(0, 0.01, d["q_true"][0].shape)
    f, g, H, terms, qproj = oracle.objective(sk25, cams6, opts, None, q0, d["meas"][0], d["weight"][0], want_grad=True, want_H=True)
    indep = skeleton.independent_dofs(sk25)
    rng = np.random.default_rng(4)
    for _ in range(12):
        n, k = rng.integers(0, 8), rng.integers(0, abi.NX)
        fp = oracle.objective(sk25, cams6, opts, None, oracle.move_coordinate(sk25, qproj, n, k, 1e-6), d["meas"][0], d["weight"][0])[0]
        fm = oracle.objective(sk25, cams6, opts, None, oracle.move_coordinate(sk25, qproj, n, k, -1e-6), d["meas"][0], d["weight"][0])[0]
        fd = (fp - fm) / 2e-6
        assert abs(fd - g[n * abi.NX + k]) < 1e-5 * max(1.0, abs(fd))
    # band matrix is symmetric positive semidefinite with half-bandwidth 3 blocks
    ntot, kd = H.shape[0], H.shape[1] - 1
    A = np.zeros((ntot, ntot))
    for i in range(ntot):
        for dd in range(min(kd, i) + 1):
            A[i, i - dd] = A[i - dd, i] = H[i, dd]
    assert np.linalg.eigvalsh(A).min() > -1e-6 * np.abs(A).max()
    assert abs(terms[:4].sum() + terms[4] - f) < 1e-9 * abs(f)


def test_motion_model_gauge(oracle):
    """free dq0/ddq0: ddq0 = ddq1 = ddq2 and the collocation holds for n >= 1"""
    rng = np.random.default_rng(2)
    q = np.cumsum(rng.normal(0, 0.01, (9, 54)), axis=0)
    h = 1 / 120
    dq, ddq = oracle.derivatives(q, h)
    assert np.abs(q[1:] - (q[:-1] + h * dq[1:])).max() < 1e-14
    assert np.abs(dq[1:] - (dq[:-1] + h * ddq[1:])).max() < 1e-10
    assert np.array_equal(ddq[0], ddq[2]) and np.array_equal(ddq[1], ddq[2])


def test_oracle_solver_converges_to_stationary_point(oracle, sk25, cams6):
    d = synth.make_batch(sk25, cams6, B=1, N=24, seed=31)
    opts = abi.default_options()
    res = oracle.solve(sk25, cams6, opts, None, d["q_init"][0], d["meas"][0], d["weight"][0])
    assert res["stats"].status == abi.OK
    f, g, _, _, _ = oracle.objective(sk25, cams6, opts, None, res["q"], d["meas"][0], d["weight"][0], want_grad=True)
    # KKT: reduced gradient vanishes except on dofs held by an ACTIVE angle bound (there it equals the multiplier)
    indep = list(skeleton.independent_dofs(sk25))
    free = np.ones((24, abi.NX), bool)
    for b in range(sk25.n_bounds):
        a, bb = sk25.bound_a[b], sk25.bound_b[b]
        v = res["q"][:, a] - (res["q"][:, bb] if bb >= 0 else 0.0)
        act = (np.abs(v - sk25.bound_up[b]) < 1e-5) | (np.abs(v - sk25.bound_lo[b]) < 1e-5)
        free[act, indep.index(a)] = False
        if bb >= 0:
            free[act, indep.index(bb)] = False
    assert np.abs(g.reshape(24, abi.NX)[free]).max() < 1e-2
    assert res["stats"].max_bound_violation < 1e-5
    truth = oracle.markers(sk25, d["q_true"][0])
    assert np.sqrt(((res["positions"] - truth) ** 2).sum(-1).mean()) < 0.05      # 2 px noise -> centimetres
    assert res["stats"].max_constraint < 1e-13


def test_oracle_solver_through_the_gimbal_region(oracle, sk25, cams6):
    """Limbs pitched beyond 90 degrees under a rolled trunk: the absolute-Euler pitch of the reference's model turns
    back and phi/psi swing by ~180 deg (29 of 798 link states of the stored run 2019_03_07/phantom are like that).
    The leg-angle coordinates alpha carry the solver through; the result satisfies all 26 joint equalities."""
    sk25 = skeleton.build_skeleton("phantom", 25)
    sk25.n_bounds = 0            # the exaggerated swing leaves the reference's joint ranges; bounds have their own test
    d = synth.make_batch(sk25, cams6, B=1, N=24, seed=61, wide_limbs=True)
    qt = d["q_true"][0]
    assert np.abs(np.array([oracle.constraints(sk25, x) for x in qt])).max() < 1e-12
    phis = qt[:, 3::3][:, 5:]
    assert (np.abs(phis) > np.pi / 2).sum() > 10                       # the other cos(phi) branch really occurs
    opts = abi.default_options()
    res = oracle.solve(sk25, cams6, opts, None, d["q_init"][0], d["meas"][0], d["weight"][0])
    assert res["stats"].status == abi.OK and res["stats"].max_constraint < 1e-12
    truth = oracle.markers(sk25, qt)
    assert np.sqrt(((res["positions"] - truth) ** 2).sum(-1).mean()) < 0.05
