"""FK / marker / joint-equality / fisheye model pinned against the reference's STORED outputs.

tests/golden/fk_csv_pin.npz holds `uv` = the numbers of data/test_set/2019_03_07/phantom/run/fte_kinematic/
cam{1..6}_fte.csv (the reference's own 2D reprojection of its own FK at its solution, 57 frames x 6 cameras x
24 markers) together with joint angles q and 6 x 14 camera parameters RECOVERED from those numbers alone by
tools/pin_fk_from_csv.py (two-view geometry -> bundle adjustment -> skeleton fit -> joint refinement).
If the FK chain, a link length, a marker offset, a joint axis or the projection formula were restated wrongly,
no (q, cameras) could reproduce the 16 416 stored values; the recovered ones do to < 1e-4 px."""
import os

import numpy as np
import pytest

from cheetah_pose_estimation_amd import abi, skeleton, synth

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "fk_csv_pin.npz"))


def _rodrigues(r):
    th = np.linalg.norm(r)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def _cams(Zp=None):
    Zp = Z if Zp is None else Zp
    cams = (abi.Camera * 6)()
    for c in range(6):
        p = Zp["cams"][c]
        cam = cams[c]
        cam.model = abi.CAM_FISHEYE
        cam.fx, cam.fy, cam.cx, cam.cy = p[0:4]
        for i in range(4):
            cam.D[i] = p[4 + i]
        R = _rodrigues(p[8:11]) if np.linalg.norm(p[8:11]) > 0 else np.eye(3)
        for i in range(9):
            cam.R[i] = R.reshape(-1)[i]
        for i in range(3):
            cam.t[i] = p[11 + i]
        cam.mult = 1.0
    return cams


def test_oracle_fk_and_projection_reproduce_the_stored_2d_files(oracle):
    sk = skeleton.build_skeleton(str(Z["animal"]), 24)
    q, uv = Z["q"], Z["uv"]
    assert uv.shape == (57, 6, 24, 2) and q.shape == (57, 54)
    cams = _cams()
    pos = oracle.markers(sk, q)                                   # C oracle FK + marker model
    got = np.array([[[oracle.project(cams[c], pos[n, l]) for l in range(24)] for c in range(6)] for n in range(57)])
    err = np.abs(got - uv)
    assert err.max() < 1e-4, err.max()                            # all 16 416 stored numbers
    assert np.sqrt((err ** 2).mean()) < 5e-6
    # the recovered angles satisfy the reference's 26 joint equalities (both cos(phi) branches occur in this run)
    c = np.array([np.abs(oracle.constraints(sk, x)).max() for x in q])
    assert c.max() < 1e-12
    assert (Z["branch"] < 0).sum() > 0
    # link lengths implied by the stored files equal cheetah_params.py (the 3D markers of one link keep their distance)
    p = skeleton.load_params("phantom")
    M = {m: i for i, m in enumerate(skeleton.MARKERS)}
    for a, b, Lref in (("tail1", "tail2", p["tail1"]["length"]), ("tail_base", "tail1", p["tail0"]["length"]),
                       ("spine", "tail_base", p["body_B"]["length"]), ("spine", "neck_base", p["body_F"]["length"]),
                       ("r_front_knee", "r_front_ankle", p["front"]["calf"]["length"]), ("l_back_ankle", "l_back_paw", p["back"]["hock"]["length"])):
        dist = np.linalg.norm(pos[:, M[a]] - pos[:, M[b]], axis=1)
        assert np.abs(dist - Lref).max() < 1e-12


ZJ = np.load(os.path.join(os.path.dirname(__file__), "golden", "fk_csv_pin_jules.npz"))


def test_second_animal_and_recording_year(oracle):
    """the same pin on another animal, another rig and another year -- data/test_set/2017_08_29/top/jules/run1_1/
    fte_kinematic/cam{1..6}_fte.csv (30 frames x 6 cameras x 24 markers, 1920 x 1080 read-out, 90 fps), recovered by
    `tools/pin_fk_from_csv.py 2017_08_29/top/jules/run1_1 fte_kinematic fk_csv_pin_jules.npz`: jules' link table of
    cheetah_params.py + the FK chain + the marker offsets + the joint equalities + the fisheye model reproduce all 8 640 stored
    numbers to < 1e-5 px (rms 4.7e-8 px)."""
    sk = skeleton.build_skeleton(str(ZJ["animal"]), 24)
    assert str(ZJ["animal"]) == "jules" and ZJ["uv"].shape == (30, 6, 24, 2) and not np.isnan(ZJ["uv"]).any()
    cams = _cams(ZJ)
    pos = oracle.markers(sk, ZJ["q"])
    got = np.array([[[oracle.project(cams[c], pos[n, l]) for l in range(24)] for c in range(6)] for n in range(30)])
    err = np.abs(got - ZJ["uv"])
    assert err.max() < 1e-5 and np.sqrt((err ** 2).mean()) < 1e-6
    assert max(np.abs(oracle.constraints(sk, x)).max() for x in ZJ["q"]) < 1e-12
    pj = skeleton.load_params("jules")
    M = {m: i for i, m in enumerate(skeleton.MARKERS)}
    for a, b, Lref in (("spine", "tail_base", pj["body_B"]["length"]), ("spine", "neck_base", pj["body_F"]["length"]),
                       ("r_back_knee", "r_back_ankle", pj["back"]["calf"]["length"]), ("l_front_ankle", "l_front_paw", pj["front"]["hock"]["length"])):
        assert np.abs(np.linalg.norm(pos[:, M[a]] - pos[:, M[b]], axis=1) - Lref).max() < 1e-12
    # and it is NOT the phantom table that fits: with phantom's link lengths the same angles miss by pixels
    skp = skeleton.build_skeleton("phantom", 24)
    posp = oracle.markers(skp, ZJ["q"])
    gotp = np.array([[oracle.project(cams[0], posp[n, l]) for l in range(24)] for n in range(30)])
    assert np.abs(gotp - ZJ["uv"][:, 0]).max() > 1.0


def test_third_rig_with_pixels_outside_the_image(oracle):
    """`2017_09_02/top/jules/run1/fte_kinematic` (third rig; `tools/pin_fk_from_csv.py 2017_09_02/top/jules/run1 fte_kinematic fk_csv_pin_0902top.npz`):
    30 frames x 6 cameras x 24 markers, of which 20 (marker, frame) pairs of camera 5 are stored as NaN (the reference's writer leaves pixels
    outside the image empty); every stored number is reproduced to < 1e-5 px."""
    Z3 = np.load(os.path.join(os.path.dirname(__file__), "golden", "fk_csv_pin_0902top.npz"))
    sk = skeleton.build_skeleton(str(Z3["animal"]), 24)
    assert str(Z3["animal"]) == "jules" and Z3["uv"].shape == (30, 6, 24, 2)
    missing = np.isnan(Z3["uv"]).any(-1)
    assert missing.sum() == 20 and missing[:, 4].sum() == 20                       # all in camera 5
    cams = _cams(Z3)
    pos = oracle.markers(sk, Z3["q"])
    got = np.array([[[oracle.project(cams[c], pos[n, l]) for l in range(24)] for c in range(6)] for n in range(30)])
    err = np.abs(got - Z3["uv"])[~missing]
    assert err.max() < 1e-5 and np.sqrt((err ** 2).mean()) < 1e-6
    assert max(np.abs(oracle.constraints(sk, x)).max() for x in Z3["q"]) < 1e-12


def test_fourth_rig(oracle):
    """`2017_09_02/bottom/jules/run2/fte_kinematic` (`tools/pin_fk_from_csv.py 2017_09_02/bottom/jules/run2 fte_kinematic fk_csv_pin_0902bot.npz`):
    33 frames x 6 cameras x 24 markers reproduced to 2.7e-5 px worst (rms 2.1e-6 px)."""
    Z4 = np.load(os.path.join(os.path.dirname(__file__), "golden", "fk_csv_pin_0902bot.npz"))
    sk = skeleton.build_skeleton(str(Z4["animal"]), 24)
    assert str(Z4["animal"]) == "jules" and Z4["uv"].shape == (33, 6, 24, 2) and not np.isnan(Z4["uv"]).any()
    cams = _cams(Z4)
    pos = oracle.markers(sk, Z4["q"])
    got = np.array([[[oracle.project(cams[c], pos[n, l]) for l in range(24)] for c in range(6)] for n in range(33)])
    err = np.abs(got - Z4["uv"])
    assert err.max() < 1e-4 and np.sqrt((err ** 2).mean()) < 1e-5
    assert max(np.abs(oracle.constraints(sk, x)).max() for x in Z4["q"]) < 1e-12


@pytest.mark.parametrize("fixture,animal,n_frames,missing_cams,tol", [
    ("fk_csv_pin_0303.npz", "phantom", 42, (3, 4), 1e-4),        # 2019_03_03/phantom/run: a 4-camera scene (cameras 4, 5 have no stored files)
    ("fk_csv_pin_1209.npz", "jules", 30, (5,), 5e-5),            # 2017_12_09/bottom/jules/flick2: 5 cameras
    ("fk_csv_pin_0309.npz", "jules", 34, (), 5e-5),              # 2019_03_09/jules/flick1: camera 2 partly outside the image (88 empty pixels)
])
def test_remaining_stored_sequences(oracle, fixture, animal, n_frames, missing_cams, tol):
    """the other three sequences of data/test_set (`tools/pin_fk_from_csv.py <seq> fte_kinematic <fixture>`): cameras without stored files count as
    empty pixels, cameras that see part of the run outside the image keep their stored NaNs; every stored number is reproduced to `tol` pixels.
    With these, the FK / marker / joint / projection model is pinned on a stored result of EVERY sequence the reference ships (10 sequences, 7
    recording days / rigs, both animals, 90 and 120 fps): 7 multi-view results with self-calibrated cameras (this file), the other 3 through
    their monocular results with the cameras of a sibling sequence (tests/test_contacts.py)."""
    Zs = np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))
    sk = skeleton.build_skeleton(str(Zs["animal"]), 24)
    assert str(Zs["animal"]) == animal and Zs["uv"].shape == (n_frames, 6, 24, 2)
    missing = np.isnan(Zs["uv"]).any(-1)
    for c in range(6):
        assert missing[:, c].all() == (c in missing_cams)
    cams = _cams(Zs)
    pos = oracle.markers(sk, Zs["q"])
    got = np.array([[[oracle.project(cams[c], pos[n, l]) for l in range(24)] for c in range(6)] for n in range(n_frames)])
    err = np.abs(got - Zs["uv"])[~missing]
    assert err.size >= n_frames * 24 * 2 * 3 and err.max() < tol and np.sqrt((err ** 2).mean()) < 0.1 * tol
    assert max(np.abs(oracle.constraints(sk, x)).max() for x in Zs["q"]) < 1e-12


@pytest.mark.parametrize("key,n_frames,tol", [("ph17", 44, 5e-4), ("j2", 34, 2e-4), ("ph0902", 45, 5e-5)])
def test_multi_view_results_of_sibling_sequences_with_cameras_held_fixed(oracle, key, n_frames, tol):
    """The three sequences that share a rig with a self-calibrated one (`2017_08_29/top/phantom/run1_1`, `.../jules/run1_2` with the cameras of
    `.../jules/run1_1`; `2017_09_02/top/phantom/run1_2` with those of `.../jules/run1`): their stored MULTI-view results `fte_kinematic/cam*_fte.csv`
    are reproduced by this repository's FK + the animal's link table with only the pose free per frame -- no camera parameter is fitted to them
    (tests/golden/fk_fixed_cameras_pin.npz, `tools/pin_contacts_from_csv.py <seq> <animal> <camera fixture> fte_kinematic ...`).  With this every
    multi-view result the reference ships is pinned (7 self-calibrated + 3 here)."""
    Zf = np.load(os.path.join(os.path.dirname(__file__), "golden", "fk_fixed_cameras_pin.npz"))
    uv, q = Zf[key + "_uv"], Zf[key + "_q"]
    assert uv.shape == (n_frames, 6, 24, 2) and float(Zf[key + "_worst_px"]) < tol
    sk = skeleton.build_skeleton(str(Zf[key + "_animal"]), 24)
    cams = _cams(np.load(os.path.join(os.path.dirname(__file__), "golden", str(Zf[key + "_cams"]))))
    pos = oracle.markers(sk, q)
    got = np.array([[[oracle.project(cams[c], pos[n, l]) for l in range(24)] for c in range(6)] for n in range(n_frames)])
    ok = ~np.isnan(uv).any(-1)
    assert ok.mean() > 0.95 and np.abs(got - uv)[ok].max() < tol
    assert max(np.abs(oracle.constraints(sk, x)).max() for x in q) < 1e-12


def test_numpy_host_fk_agrees_on_the_recovered_angles():
    sk = skeleton.build_skeleton(str(Z["animal"]), 24)
    cams = _cams()
    pos, _ = synth.fk_numpy(sk, Z["q"])
    for c in range(6):
        uv, _ = synth.project_numpy(cams[c], pos)
        assert np.abs(uv - Z["uv"][:, c]).max() < 1e-4


def real_run_problem(noise_px=2.0, drop=0.4, outliers=0.05, init_noise=0.05, seed=0, fixture=None):
    """An estimation problem on REAL cheetah motion: the 57-frame trajectory and the 6 cameras recovered from the
    reference's stored run, its own reprojections as measurements plus DLC-like noise, drop-outs and outliers, and
    a perturbed start.  Limb links swing beyond the horizontal under a rolled trunk in this run.  `fixture`: another of the
    recovered runs (file name in tests/golden; it must have no empty pixels)."""
    Zr = Z if fixture is None else np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))
    sk = skeleton.build_skeleton(str(Zr["animal"]), 24)
    q, uv = Zr["q"], Zr["uv"]
    N = q.shape[0]
    rng = np.random.default_rng(seed)
    meas = uv + rng.normal(0, noise_px, uv.shape)
    weight = np.broadcast_to(1.0 / skeleton.measurement_sigma(24, False), (N, 6, 24)).copy()
    weight[rng.random((N, 6, 24)) < drop] = 0.0
    out = rng.random((N, 6, 24)) < outliers
    meas[out] += rng.normal(0, 200, (out.sum(), 2))
    ind = skeleton.independent_dofs(sk)
    q_init = q.copy()
    q_init[:, ind] += rng.normal(0, init_noise, (N, len(ind)))
    return sk, _cams(Zr), q_init, np.ascontiguousarray(meas), weight, q


def test_oracle_solver_on_the_real_run(oracle):
    """2 px noise, 40 % drop-outs, 5 % gross outliers on real motion: the solve converges, keeps the joint ranges and
    lands within millimetres of the reference's own stored solution."""
    sk, cams, q_init, meas, weight, q_ref = real_run_problem()
    opts = abi.default_options(90.0)
    res = oracle.solve(sk, cams, opts, None, q_init, meas, weight)
    st = res["stats"]
    assert st.status == abi.OK and st.iterations < 60
    assert st.max_bound_violation < 1e-5 and st.max_constraint < 1e-12
    err = np.sqrt(((res["positions"] - oracle.markers(sk, q_ref)) ** 2).sum(-1))
    assert np.sqrt((err ** 2).mean()) < 0.008 and err.max() < 0.04


def _cams_pinhole(Zp):
    """cameras of a kinetic-dataset fixture (tools/pin_fk_pinhole.py): [fx fy cx cy | k1 k2 p1 p2 k3 | rvec | t] in OpenCV's order.  The reference's
    pinhole model (acinoset_misc.py:1682-1696) is purely radial, 1 + D0 r^2 + D1 r^4 + D2 r^6; the fit left the tangential pair free and found
    |p1|, |p2| < 4e-7, which is how the stored files say that the model has no such terms.  They are dropped here (< 1e-4 px)."""
    C = Zp["cams"].shape[0]
    cams = (abi.Camera * C)()
    for c in range(C):
        p = Zp["cams"][c]
        cam = cams[c]
        cam.model = abi.CAM_PINHOLE
        cam.fx, cam.fy, cam.cx, cam.cy = p[0:4]
        cam.D[0], cam.D[1], cam.D[2], cam.D[3] = p[4], p[5], p[8], 0.0
        R = _rodrigues(p[9:12]) if np.linalg.norm(p[9:12]) > 0 else np.eye(3)
        for i in range(9):
            cam.R[i] = R.reshape(-1)[i]
        for i in range(3):
            cam.t[i] = p[12 + i]
        cam.mult = 1.0
    return cams


KINETIC_PINS = [f for f in ("fk_csv_pin_arabia.npz", "fk_csv_pin_shiraz.npz") if os.path.exists(os.path.join(os.path.dirname(__file__), "golden", f))]


@pytest.mark.parametrize("fixture", KINETIC_PINS)
def test_kinetic_dataset_pinhole_rig(oracle, fixture):
    """the only reference-held fixtures for the `-02` skeletons and the four-camera pinhole rig of the kinetic dataset --
    data/test_set/kinetic_dataset/<day>/<animal>/<trial>/fte_kinematic/cam{1..4}_fte.csv (the reference's reprojection of its own FK at its solution,
    empty cells where a marker leaves the 1280 x 720 image).  tools/pin_fk_pinhole.py recovered joint angles and camera parameters from those
    numbers alone (focal scan + two-view geometry -> bundle adjustment -> skeleton fit -> joint refinement in the solver's own leg-angle
    coordinates).  The `<animal>-02` link table, the FK chain, the marker offsets, the 26 joint equalities and the pinhole + radial projection
    reproduce every stored number to < 1e-4 px (arabia: 4 335 visible points, max 8.1e-5 px, rms 2.1e-5 px; the fit itself, with its two free
    tangential terms, closes to 6.7e-6 px)."""
    Zp = np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))
    animal = str(Zp["animal"])
    sk = skeleton.build_skeleton(f"{animal}-02", 24, kinetic_dataset=True)
    q, uv = Zp["q"], Zp["uv"]
    assert uv.shape[1:] == (4, 24, 2) and q.shape == (uv.shape[0], 54)
    cams = _cams_pinhole(Zp)
    pos = oracle.markers(sk, q)
    got = np.array([[[oracle.project(cams[c], pos[n, l]) for l in range(24)] for c in range(4)] for n in range(q.shape[0])])
    seen = ~np.isnan(uv).any(-1)
    assert seen.sum() > 3000 and 0.02 < (~seen).mean() < 0.3          # the gaps are part of the fixture
    err = np.abs(got - uv)[seen]
    # arabia trial06: max 8.1e-5 px.  shiraz trial01 (56 frames, 28 % of the cells empty, camera 3 sees 526 points of 1 344): the radial-only fit stands at
    # max 6.2e-4 px, rms 4.6e-5 px after seventeen rounds and still creeps (camera 3's higher distortion terms are barely observable from so few points) -- three orders below
    # what a wrong link length or marker offset leaves (pixels)
    tol, tol_rms = (1e-4, 3e-5) if animal == "arabia" else (1e-3, 1e-4)
    assert err.max() < tol, err.max()
    assert np.sqrt((err ** 2).mean()) < tol_rms
    assert max(np.abs(oracle.constraints(sk, x)).max() for x in q) < 1e-12
    assert np.abs(Zp["cams"][:, 6:8]).max() < 1e-6                     # no tangential distortion in the stored numbers
    # The world frame of such a pin is free (2D data fix cameras and animal only up to a rigid motion); tools/reframe_kinetic_pin.py chose its tilt -- two
    # angles -- so that the recovered angles violate this repository's `-02` bound table least (cheetah.py:306-352: |roll of the rear body| <= 0.05,
    # neck / spine / tail yaw and roll differences within +-0.05 ... +-0.1; none of them is invariant under a tilt).  In that frame the stored solution
    # satisfies EVERY bound in every frame -- arabia: five bounds that were violated by up to 0.10 rad before, shiraz: four by up to 0.27 rad, each over
    # 50 - 56 frames, with two free numbers: the table (which angles, which ranges) is the reference's.
    assert "world_tilt" in Zp.files and abs(np.linalg.det(Zp["world_tilt"]) - 1.0) < 1e-12
    worst = 0.0
    for b in range(sk.n_bounds):
        ia, ib = sk.bound_a[b], sk.bound_b[b]
        dlt = q[:, ia] - (q[:, ib] if ib >= 0 else 0.0)
        dlt = (dlt + np.pi) % (2 * np.pi) - np.pi
        worst = max(worst, float(np.maximum(dlt - sk.bound_up[b], sk.bound_lo[b] - dlt).max()))
    assert worst < 1e-9, worst
    assert np.degrees(np.arccos(np.clip(Zp["world_tilt"][2, 2], -1, 1))) > 5.0        # (the animal-fixed frame of the raw pin was 10 - 15 degrees off)



def kinetic_dataset_problem(fixture="fk_csv_pin_arabia.npz", noise_px=2.0, drop=0.3, outliers=0.05, init_noise=0.05, seed=0):
    """An estimation problem on REAL motion of the kinetic dataset: the trajectory and the four pinhole cameras recovered from the stored kinematic
    result of a trial (in the world frame its joint ranges fix), its own reprojections as measurements (empty cells: weight 0) plus noise, drop-outs
    and outliers, a perturbed start; the dataset's objective multipliers [1, 1, .6, .6] (acinoset_misc.py:463-465), its 7 px sigma and the `-02`
    skeleton with its tight ranges.  A 200 fps gallop with four leg links beyond the horizontal."""
    Zr = np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))
    sk = skeleton.build_skeleton(f"{str(Zr['animal'])}-02", 24, kinetic_dataset=True)
    q, uv = Zr["q"], Zr["uv"]
    N = q.shape[0]
    rng = np.random.default_rng(seed)
    seen = ~np.isnan(uv).any(-1)
    meas = np.nan_to_num(uv) + rng.normal(0, noise_px, uv.shape)
    weight = np.broadcast_to(1.0 / skeleton.measurement_sigma(24, True), (N, 4, 24)).copy()
    weight[~seen] = 0.0
    weight[rng.random((N, 4, 24)) < drop] = 0.0
    out = rng.random((N, 4, 24)) < outliers
    meas[out] += rng.normal(0, 200, (out.sum(), 2))
    cams = _cams_pinhole(Zr)
    for c, mlt in enumerate((1.0, 1.0, 0.6, 0.6)):
        cams[c].mult = mlt
    ind = skeleton.independent_dofs(sk)
    q_init = q.copy()
    q_init[:, ind] += rng.normal(0, init_noise, (N, len(ind)))
    return sk, cams, q_init, np.ascontiguousarray(meas), weight, q


def test_oracle_solver_on_a_real_trial_of_the_kinetic_dataset(oracle):
    """arabia trial06 at 200 fps: 2 px noise, 30 % drop-outs on top of the 10 % of cells the cameras do not see, 5 % gross outliers, start 0.05 rad off:
    the solve converges inside the `-02` joint ranges (four multiplier updates) and lands within a centimetre of the reference's own stored solution.
    Round 3, before the cost pitch: no stop within 200 iterations even from the stored solution -- the principal pitch the model term then
    saw turns around where these limbs pass the horizontal (15 420.9 against 102.5, tests/test_deviations.py)."""
    sk, cams, q_init, meas, weight, q_ref = kinetic_dataset_problem()
    opts = abi.default_options(200.0)
    res = oracle.solve(sk, cams, opts, None, q_init, meas, weight)
    st = res["stats"]
    err = np.sqrt(((res["positions"] - oracle.markers(sk, q_ref)) ** 2).sum(-1))
    print(f"real kinetic-dataset trial: {st.iterations} iterations / {st.outer} multiplier updates, model term {st.cost_model:.1f}, RMSE to the stored solution "
          f"{np.sqrt((err ** 2).mean()):.4f} m, worst {err.max():.3f} m")
    assert st.status == abi.OK and st.iterations < 60
    assert st.max_bound_violation < 1e-5 and st.max_constraint < 1e-12
    assert np.sqrt((err ** 2).mean()) < 0.012 and err.max() < 0.06
    assert np.abs(res["q"][:, 4::3]).max() <= np.pi / 2 + 1e-12                        # the returned angles are the principal triple (outputs, FK) ...
    assert np.abs(synth.cost_view_numpy(sk, res["q"])[:, 4::3]).max() > 1.7             # ... the cost terms saw pitches beyond the horizontal
