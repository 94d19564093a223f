#!/usr/bin/env python3
"""Generate golden vectors from the reference's OWN pure functions (build container only).

`acinoset_misc.py` imports pyomo, cv2 and the two absent git submodules at module level, none of which
the functions below use (except `pyo.atan`).  Following SURVEY.md 8c-4, empty stand-in modules are
pre-seeded in sys.modules (pyomo.environ.atan = math.atan) so that the module imports; then the
reference's functions are CALLED on seeded inputs and (input, output) pairs are written to
tests/golden/misc_golden.npz.  Only numbers are committed -- no reference source.

Functions exercised (acinoset_misc.py): pt3d_to_2d_fisheye :1663, pt3d_to_2d :1682, redescending_loss
:2001, get_uncertainty_models :1760, get_relative_angles :487 (numpy branch), get_relative_angle_mask
:1699, get_markers :1914, get_dlc_marker_indices :1943, get_pairwise_graph :1972, rmse :93-98; into a second file: the contact
helpers (:69-90, :2033-2057) and traj_smoothness / traj_error (:1170-1199).
"""
import json
import math
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def seed_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    pyomo = mod("pyomo")
    env = mod("pyomo.environ", atan=math.atan, ConcreteModel=_Any, Objective=_Any, Constraint=_Any, Var=_Any,
              Param=_Any, RangeSet=_Any, value=lambda v: v)
    pyomo.environ = env
    util = mod("pyomo.util")
    mod("pyomo.util.infeasible", log_infeasible_constraints=lambda *a, **k: None)
    pyomo.util = util
    mod("cv2")
    shared = mod("shared")
    pe = mod("shared.physical_education")
    shared.physical_education = pe
    for sub in ("system", "links", "foot", "motor", "utils", "drag", "spring", "damper"):
        sm = mod(f"shared.physical_education.{sub}", System3D=_Any, Link3D=_Any, Foot3D=_Any, Motor3D=_Any)
        setattr(pe, sub, sm)
    common = mod("common")
    pu = mod("common.py_utils")
    common.py_utils = pu
    pu.data_ops = mod("common.py_utils.data_ops")
    pu.log = mod("common.py_utils.log", logger=lambda name: _Any())


def main():
    sys.dont_write_bytecode = True
    seed_stubs()
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("Agg")
    import acinoset_misc as misc  # the reference module itself

    rng = np.random.default_rng(20241008)
    out = {}
    # ---- projection ---------------------------------------------------------------------------------
    n = 64
    pts = np.c_[rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(2, 12, n)]
    K = np.array([[1241.84, 0, 1346.96], [0, 1239.92, 773.02], [0, 0, 1.0]])
    Df = np.array([0.0366, 0.0480, -0.0347, 0.0074])
    Dp = np.array([-0.12, 0.05, -0.01])
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = 0.4
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + math.sin(ang) * Kx + (1 - math.cos(ang)) * Kx @ Kx
    t = np.array([[0.3], [-0.2], [0.5]])
    uvf = np.array([misc.pt3d_to_2d_fisheye(p[0], p[1], p[2], K, Df, R, t) for p in pts], dtype=float)
    uvp = np.array([misc.pt3d_to_2d(p[0], p[1], p[2], K, Dp, R, t) for p in pts], dtype=float)
    out.update(proj_pts=pts, proj_K=K, proj_Df=Df, proj_Dp=Dp, proj_R=R, proj_t=t.ravel(), proj_uv_fisheye=uvf, proj_uv_pinhole=uvp)
    # ---- robust loss --------------------------------------------------------------------------------
    errs = np.concatenate([np.linspace(-40, 40, 161), rng.uniform(-30, 30, 40), [0.0, 0.5, 5.0, 100.0, -100.0]])
    out.update(loss_err=errs, loss_val=np.array([misc.redescending_loss(e, 3, 10, 20) for e in errs], dtype=float))
    errs2 = rng.uniform(-15, 15, 32)
    out.update(loss2_err=errs2, loss2_val=np.array([misc.redescending_loss(e, 2, 6, 9) for e in errs2], dtype=float))
    # ---- uncertainty tables ---------------------------------------------------------------------------
    R_pw, Q = misc.get_uncertainty_models()
    out.update(R_pw=np.asarray(R_pw, float), Q=np.asarray(Q, float))
    # ---- relative angles ------------------------------------------------------------------------------
    q = rng.normal(0, 0.7, (5, 54))
    mask = misc.get_relative_angle_mask()
    x = np.array([np.array(sum(misc.get_relative_angles(q, fe), []))[mask] for fe in range(5)], dtype=float)
    out.update(rel_q=q, rel_x=x, rel_mask=np.asarray(mask[0], np.int64))
    # ---- parity metric ----------------------------------------------------------------------------------
    a = rng.normal(size=(7, 24, 3)); b = a + rng.normal(0, 0.01, a.shape)
    out.update(metric_a=a, metric_b=b, metric_rmse=float(misc.rmse(a, b)))
    os.makedirs(OUT, exist_ok=True)
    np.savez(os.path.join(OUT, "misc_golden.npz"), **out)
    # ---- helpers of the contact heuristic and the trajectory metrics (second file; the first one stays byte-identical) ----
    # acinoset_misc.py: SimpleLinearModel :69-81, bound_value :84-90, positive_zero_crossings :2033-2046,
    # group_by_consecutive_values :2049-2051, find_minimum_foot_height :2054-2057, traj_smoothness :1170-1176, traj_error :1179-1199
    rng2 = np.random.default_rng(20241009)
    aux, lists = {}, {}
    for name, pts in (("stance", [[9.0, 0.09], [14.0, 0.06]]), ("lfl", [[9.0, 2.0], [15.0, 1.8]]), ("lhl", [[9.0, 2.1], [15.0, 2.6]]),
                      ("nlfl", [[9.5, 2.1], [15.0, 2.0]]), ("nlhl", [[9.0, 1.7], [15.0, 2.5]])):
        mdl = misc.SimpleLinearModel(pts)
        aux[f"line_{name}"] = np.array([mdl.m, mdl.c, mdl.predict(11.3)], dtype=float)
    vals = np.array([2.5, -1.25, 0.0, 1e-3, -7.0])
    aux["bound_in"] = vals
    aux["bound_out"] = np.array([misc.bound_value(float(v), 0.2) for v in vals], dtype=float)
    series = []
    for k in range(4):
        v = np.sin(np.linspace(0, 9 + k, 70) + 0.3 * k) + 0.1 * rng2.normal(size=70)
        v[rng2.integers(0, 70, 6)] = 0.0                      # exact zeros are dropped before the sign test
        cnt, idx = misc.positive_zero_crossings(v)
        series.append(v)
        lists[f"zc_{k}"] = dict(count=int(cnt), idx=[int(i) for i in idx])
    aux["zc_series"] = np.array(series)
    for k, arr in enumerate(([3, 4, 5, 9, 10, 14], [0], [2, 3, 7, 8, 9, 20, 22, 23], [])):
        lists[f"runs_{k}"] = dict(inp=[int(a) for a in arr], out=[[int(a) for a in run] for run in misc.group_by_consecutive_values(np.array(arr, dtype=int))])
    h = rng2.uniform(0, 0.3, 50)
    aux["minh_series"] = h
    regions = [(0, 50), (10, 30), (25, -1), (40, 45)]
    aux["minh_regions"] = np.array(regions)
    aux["minh_out"] = np.array([misc.find_minimum_foot_height(h, r) for r in regions], dtype=np.int64)
    X = rng2.normal(size=(12, 24, 3)).cumsum(axis=0) * 0.05
    Y = X + rng2.normal(0, 0.02, X.shape)
    aux["traj_X"], aux["traj_Y"] = X, Y
    aux["traj_smoothness"] = np.array(misc.traj_smoothness(X, Y))
    for centered in (False, True):
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            res, per_frame, smooth = misc.traj_error(X.copy(), Y.copy(), "golden", centered=centered)   # the function edits its inputs
        tag = "c" if centered else "u"
        aux[f"traj_mpjpe_{tag}"] = res.to_numpy(dtype=float).ravel()
        aux[f"traj_frame_{tag}"] = np.asarray(per_frame, dtype=float)
        aux[f"traj_smooth_{tag}"] = np.array(float(smooth))
    # ---- the assembled contact rule and the force templates: contact_detection :745-862, synth_grf_data :865-943 ---------
    # Both take the Pyomo robot.  Only its accessors are stood in for -- the four feet as plain objects holding a name and the
    # foot-height series, `pe.foot.feet`, `pe.utils.get_vals` (returns the series), `pe.foot.Foot3D.ground_plane_height` = 0.0
    # (the constant lives in the absent submodule), `init_foot_height` (no-op) and `init_foot_velocity` (returns the velocity
    # series), and `DataFrame.to_hdf` (PyTables is absent: the frame is captured instead of written).  The decision logic that
    # runs is the reference's own.
    import tempfile
    import pandas as pd
    names = ["HFL_foot", "HFR_foot", "HBL_foot", "HBR_foot"]
    pe = sys.modules["shared.physical_education"]
    cases = []
    for case, (N, fps, speed, start, touch) in enumerate(((60, 120.0, 12.0, 100, (10, 22, 34, 46)), (90, 120.0, 10.5, 0, (2, 40, 30, 85)),
                                                          (70, 200.0, 9.0, 7, (15, 33, 51, 60)))):
        n = np.arange(N)
        z = 0.2 + 0.02 * np.sin(0.37 * n[:, None] + np.arange(4)[None, :]) + 0.003 * rng2.normal(size=(N, 4))
        for i, c in enumerate(touch):
            w = np.abs(n - c) <= 8
            z[w, i] = 0.01 + 0.19 * ((n[w] - c) / 8.0) ** 2
        if case == 1:                                       # a second contact of the first foot
            w = np.abs(n - 70) <= 8
            z[w, 0] = 0.012 + 0.19 * ((n[w] - 70) / 8.0) ** 2
        vel = np.zeros((N, 4, 3))
        vel[:, :, 2] = np.gradient(z, 1.0 / fps, axis=0)
        feet = [types.SimpleNamespace(name=nm, pyomo_vars={"foot_height": z[:, i:i + 1].copy()}) for i, nm in enumerate(names)]
        robot = types.SimpleNamespace(m=types.SimpleNamespace(fe=range(1, N + 1)))
        pe.foot.feet = lambda r, feet=feet: feet
        pe.foot.Foot3D = types.SimpleNamespace(ground_plane_height=0.0)
        pe.utils.get_vals = lambda var, idx: var
        misc.init_foot_height = lambda r: None
        misc.init_foot_velocity = lambda r, vel=vel: vel
        with tempfile.TemporaryDirectory() as tmp:
            contacts, by_height = misc.contact_detection(robot, start, speed, fps, tmp, plot=False)
            with open(os.path.join(tmp, "grf", "autogen-contact.json")) as f:
                cj = json.load(f)
            captured = {}
            orig = pd.DataFrame.to_hdf
            pd.DataFrame.to_hdf = lambda self, *a, **k: captured.setdefault("df", self.copy())
            try:
                direction = -1.0 if case != 1 else 1.0
                misc.synth_grf_data(robot, speed, direction, os.path.join(tmp, "grf"))
            finally:
                pd.DataFrame.to_hdf = orig
        df = captured["df"]
        plates = {int(k): df.loc[k].to_numpy(dtype=float).tolist() for k in df.index.get_level_values(0).unique()}
        aux[f"cd_height_{case}"], aux[f"cd_velz_{case}"] = z, vel[:, :, 2]
        cases.append(dict(N=N, fps=fps, speed=speed, start_frame=start, direction=direction, contacts=contacts, by_height=by_height,
                          json=cj, plates=plates))
    lists["contact_cases"] = cases
    np.savez(os.path.join(OUT, "contacts_metrics_golden.npz"), **aux)
    with open(os.path.join(OUT, "contacts_metrics_lists.json"), "w") as f:
        json.dump(lists, f, indent=1, sort_keys=True)
    names = dict(markers=misc.get_markers(), dlc_index=misc.get_dlc_marker_indices(), pairwise=misc.get_pairwise_graph())
    with open(os.path.join(OUT, "misc_names.json"), "w") as f:
        json.dump(names, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(OUT, "misc_golden.npz"), {k: np.shape(v) for k, v in out.items()})
    print("spot:", misc.redescending_loss(0.5, 3, 10, 20), misc.redescending_loss(5, 3, 10, 20), misc.redescending_loss(100, 3, 10, 20))


if __name__ == "__main__":
    main()
