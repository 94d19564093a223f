"""A pin of the MASS model on the reference's own stored physics-based results (CPU; the GPU twin is in tests/test_gpu_kinetic.py).

`fte_kinetic_<cam>/cam{1..6}_fte.csv` are the 2D reprojections of the trajectories the reference's physics-based monocular model found
(estimate_kinetics, acinoset_opt.py:693-963, equations of motion from models/*-test_tmp.robot).  The joint angles are recovered from them
with cameras that are already pinned (tools/pin_contacts_from_csv.py <seq> phantom <camera fixture> fte_kinetic_<cam> ...; 2.3e-6 px and
7.6e-6 px worst); tests/golden/kinetic_pin_*.npz hold those angles and the stance table of the stored grf/autogen-contact.json.

In the frames in which, by that table, no paw is on the ground, nothing but gravity acts on the animal: whatever the limbs do, the centre
of mass of a correct mass model falls with g.  The reference's result obeys ITS mass model (masses and centre-of-mass offsets of
cheetah_params.py inside its equations of motion); here the centre of mass is computed from THIS repository's link masses, centre-of-mass
offsets, link lengths and forward kinematics -- the same tables the GRF fit and the physics-based model use -- and its second difference at
the recording's frame rate is compared with 9.81 m/s^2.  A wrong mass table, a wrong frame rate or a wrong chain shows at once (control below)."""
import os

import numpy as np
import pytest

from cheetah_pose_estimation_amd import skeleton

G = 9.81


def _flight_acceleration(oracle, sk, Z):
    q, fps, st = Z["q"], float(Z["fps"]), Z["stance"]
    com = oracle.com(sk, q)
    acc = (com[2:] - 2.0 * com[1:-1] + com[:-2]) * fps ** 2                      # at frames 1 .. N-2
    fl = [n for n in range(1, len(q) - 1) if st[n - 1:n + 2].sum() == 0]           # no paw down in any of the three frames of the difference
    return acc[[n - 1 for n in fl]], fl


@pytest.mark.parametrize("fixture", ["kinetic_pin_phantom2017.npz", "kinetic_pin_phantom0902.npz"])
def test_centre_of_mass_falls_with_g_in_the_flight_phases_of_the_stored_physics_results(oracle, fixture):
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))
    assert float(Z["worst_px"]) < 1e-4 and float(Z["fps"]) == 90.0
    sk = skeleton.build_skeleton("phantom", 24)
    A, fl = _flight_acceleration(oracle, sk, Z)
    assert len(fl) == 14                                                          # two aerial phases in each run
    mag = np.linalg.norm(A, axis=1)
    mean = A.mean(0)
    assert abs(np.linalg.norm(mean) - G) < 0.015 * G                               # 9.875 and 9.755 m/s^2: the mean is within 0.7 %
    assert np.abs(mag - G).max() < 0.06 * G                                        # every single flight frame within 5 % (second differences at 90 fps)
    down = -mean / np.linalg.norm(mean)
    assert np.degrees(np.arccos(np.clip(down @ Z["ground_normal"], -1, 1))) < 5.0  # and it points along the normal of the plane the paws touch
    # control: the same angles with the total mass spread evenly over the links -- no longer a body in free fall
    sku = skeleton.build_skeleton("phantom", 24)
    total = sum(sku.mass[i] for i in range(sku.n_links))
    for i in range(sku.n_links):
        sku.mass[i] = total / sku.n_links
    Au, _ = _flight_acceleration(oracle, sku, Z)
    assert np.abs(np.linalg.norm(Au, axis=1) - G).max() > 0.25 * G
    # control: the wrong frame rate (120 instead of 90 fps) scales the acceleration by 16 / 9
    assert abs(np.linalg.norm(mean) * (120.0 / 90.0) ** 2 - G) > 0.7 * G


@pytest.mark.parametrize("fixture", ["kinetic_pin_phantom2017.npz", "kinetic_pin_phantom0902.npz"])
def test_contact_rules_hold_in_the_stored_physics_results(oracle, fixture):
    """The reference's physics-based model keeps a foot within `height_uncertainty_m` = 0.1 m of the ground and (nearly) still while it is in one of
    the stored contact windows (acinoset_opt.py:780-812: `foot_height` bounds, `gamma <= 1`).  This repository reads "foot" as the hock-bottom
    markers (skeleton.FOOT_MARKERS).  On the reference's own stored results those markers behave as the rules say: inside the windows
    they stay in a band of 0.13 m around the median stance height and move at ~1 m/s (2.2 at most, backward differences at 90 fps); outside they
    are 0.13 m up (median) and move at ~9.5 m/s.  Another marker choice, or windows shifted by a few frames, would not show this."""
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))
    q, fps, st, up = Z["q"], float(Z["fps"]), Z["stance"], Z["ground_normal"]
    sk = skeleton.build_skeleton("phantom", 24)
    feet = [skeleton.MARKERS.index(m) for m in skeleton.FOOT_MARKERS]
    P = oracle.markers(sk, q)[:, feet]
    on = st == 1
    on[0] = False
    off = (st == 0) & (np.arange(len(q))[:, None] > 0)
    hgt = P @ up
    hgt = hgt - np.median(hgt[on])
    v = np.zeros_like(P); v[1:] = (P[1:] - P[:-1]) * fps
    vh = v - np.einsum("nkd,d->nk", v, up)[..., None] * up
    speed = np.linalg.norm(vh, axis=-1)
    assert on.sum() == 36                                                          # four windows of nine frames
    assert hgt[on].min() > -0.03 and hgt[on].max() < 0.12 and np.median(hgt[off]) > 0.12
    assert np.median(speed[on]) < 1.3 and speed[on].max() < 2.2 and np.median(speed[off]) > 8.0
    # control: the same windows moved by five frames catch the paws in the air
    sh = np.roll(on, 5, axis=0); sh[:5] = False
    assert np.median(speed[sh]) > 2.0 * np.median(speed[on])


def _stance_by_speed(P, fps, limit=3.0):
    v = np.zeros_like(P); v[1:] = (P[1:] - P[:-1]) * fps; v[0] = v[1]
    return np.linalg.norm(v, axis=-1) < limit, v


def test_kinetic_dataset_physics_result_obeys_the_mass_model_and_the_contact_rules(oracle):
    """The same two pins on the KINETIC DATASET (its `-02` skeletons, 200 fps and contact rules had no stored result behind them):
    `kinetic_dataset/2009_09_07/arabia/trial06/fte_kinetic/cam{1..4}_fte.csv` is the 2D reprojection of what the reference's physics-based stage found
    (run_kinetic, run_dataset.py:1092-1140).  tools/pin_kinetic_dataset_result.py recovered its joint angles with the trial's cameras held FIXED at what
    its kinematic result gave (tests/golden/fk_csv_pin_arabia.npz): 4.7e-6 px worst over 50 frames -- by itself a second pin of FK + `arabia-02` table +
    pinhole projection with no camera freedom.  No contact file of the kinetic dataset is shipped, so stance is read off the result: a hock-bottom
    marker (skeleton.FOOT_MARKERS) slower than 3 m/s (swing: 6 - 27 m/s) -- four windows of 9 - 11 frames, one per foot.  Then
      * all 39 stance positions lie within 12.4 mm of ONE plane (rms 4.4 mm): the kinetic dataset's `foot_height` tolerance is 0.03 (acinoset_opt.py:783-790);
      * inside the windows the marker moves along the plane's normal at 0.15 m/s (median), never faster than 1 m/s: `foot_z_vel <= 1` (:807-810, :864-866);
      * in the aerial phase between the fore and the hind stances the centre of mass of THIS repository's `arabia-02` mass table falls with 9.9 m/s^2
        on average (second differences at 200 fps, six frames, 9.08 ... 10.77), 7 degrees from the normal of that plane; with the mass spread evenly
        over the links it 'falls' with 22 - 93 m/s^2."""
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "kinetic_pin_arabia.npz"))
    assert float(Z["worst_px"]) < 1e-5 and float(Z["fps"]) == 200.0 and Z["q"].shape == (50, 54)
    sk = skeleton.build_skeleton("arabia-02", 24, kinetic_dataset=True)
    q, fps = Z["q"], float(Z["fps"])
    feet = [skeleton.MARKERS.index(m) for m in skeleton.FOOT_MARKERS]
    P = oracle.markers(sk, q)[:, feet]
    st, v = _stance_by_speed(P, fps)
    runs = []
    for k in range(4):
        on = np.nonzero(st[:, k])[0]
        assert 8 <= len(on) <= 12 and on[-1] - on[0] == len(on) - 1            # one contiguous window per foot
        runs.append((on[0], on[-1]))
    pts = P[st]
    c = pts.mean(0)
    n = np.linalg.svd(pts - c)[2][-1]
    n = n if n[2] > 0 else -n
    h = (P - c) @ n
    assert np.abs(h[st]).max() < 0.02 and np.sqrt((h[st] ** 2).mean()) < 0.008          # inside the 0.03 of the kinetic dataset
    assert np.median(h[~st]) > 0.1                                                      # swing: well above
    inner = st.copy()
    for k, (a, b) in enumerate(runs):
        inner[a, k] = inner[b, k] = False                                               # the touch-down and lift-off frames carry the swing's speed
    vn = v @ n
    assert np.abs(vn[inner]).max() < 1.0 and np.median(np.abs(vn[inner])) < 0.3
    # aerial phase: no foot in stance within two frames
    anyst = st.any(1)
    N = len(q)
    fl = [m for m in range(3, N - 3) if not anyst[m - 2:m + 3].any()]
    assert len(fl) >= 4
    com = oracle.com(sk, q)
    acc = (com[2:] - 2.0 * com[1:-1] + com[:-2]) * fps ** 2
    A = acc[[m - 1 for m in fl]]
    mean = A.mean(0)
    assert abs(np.linalg.norm(mean) - G) < 0.03 * G, np.linalg.norm(mean)
    assert np.abs(np.linalg.norm(A, axis=1) - G).max() < 0.11 * G                       # 9.08 ... 10.77: second differences at 200 fps
    assert np.degrees(np.arccos(np.clip(-mean @ n / np.linalg.norm(mean), -1, 1))) < 5.0
    sku = skeleton.build_skeleton("arabia-02", 24, kinetic_dataset=True)
    total = sum(sku.mass[i] for i in range(sku.n_links))
    for i in range(sku.n_links):
        sku.mass[i] = total / sku.n_links
    cu = oracle.com(sku, q)
    au = ((cu[2:] - 2.0 * cu[1:-1] + cu[:-2]) * fps ** 2)[[m - 1 for m in fl]]
    assert np.linalg.norm(au, axis=1).min() > 2.0 * G
    assert abs(np.linalg.norm(mean) * (120.0 / 200.0) ** 2 - G) > 0.5 * G               # the frame rate is part of the statement
    # The world frame of these pins was fixed by the joint-angle bounds of the KINEMATIC result alone (tools/reframe_kinetic_pin.py, tests/test_fk_pin.py).
    # Three independent things agree in it: the physics-based result obeys the same bounds (to IPOPT's tolerance), the ground the paws touch is
    # horizontal, and the centre of mass falls along -z.
    viol = 0.0
    for b in range(sk.n_bounds):
        ia, ib = sk.bound_a[b], sk.bound_b[b]
        dlt = q[:, ia] - (q[:, ib] if ib >= 0 else 0.0)
        dlt = (dlt + np.pi) % (2 * np.pi) - np.pi
        viol = max(viol, float(np.maximum(dlt - sk.bound_up[b], sk.bound_lo[b] - dlt).max()))
    assert viol < 2e-3, viol                                                            # measured 5.4e-4 (fte_kinetic), 9.9e-4 (fte_grf)
    assert np.degrees(np.arccos(np.clip(n[2], -1, 1))) < 2.0                            # stance plane: 1.0 degree from z
    assert np.degrees(np.arccos(np.clip(-mean[2] / np.linalg.norm(mean), -1, 1))) < 3.0 # free fall: 1.7 degrees from -z
    assert P[st][:, 2].max() - P[st][:, 2].min() < 0.03                                 # all stance heights within 24 mm in z


def test_last_stage_of_the_kinetic_dataset_driver_is_reproduced_in_2d(oracle):
    """`fte_grf/cam{1..4}_fte.csv` of the same trial (module-level estimate_grf, run_dataset.py:1138): angles recovered with the same fixed cameras to
    4.3e-6 px; FK + `arabia-02` + pinhole reproduce the stored pixels through the oracle"""
    from test_fk_pin import _cams_pinhole
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "kinetic_pin_arabia_grf.npz"))
    assert float(Z["worst_px"]) < 1e-5
    sk = skeleton.build_skeleton("arabia-02", 24, kinetic_dataset=True)
    cams = _cams_pinhole(Z)
    pos = oracle.markers(sk, Z["q"])
    uv = Z["uv"]
    seen = ~np.isnan(uv).any(-1)
    got = np.array([[[oracle.project(cams[c], pos[m, l]) for l in range(24)] for c in range(4)] for m in range(len(pos))])
    assert np.abs(got - uv)[seen].max() < 1e-4
