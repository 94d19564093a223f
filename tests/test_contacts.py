"""Contact heuristic and synthetic force templates (cheetah_pose_estimation_amd/contacts.py), host logic only.
Two kinds of tests: hand-worked cases of the rules stated in acinoset_misc.py:745-943, and comparisons with the outputs of the
reference's OWN functions (contact_detection, synth_grf_data, their helpers, traj_error) on seeded series, committed as
tests/golden/contacts_metrics_* by tools/gen_golden.py."""
import json
import os

import numpy as np
import pytest

from cheetah_pose_estimation_amd import contacts as ct

FEET = ["HFL_foot", "HFR_foot", "HBL_foot", "HBR_foot"]


def test_linear_models_and_stance_length():
    m, c = ct.line_through(ct.STANCE_TIME_PTS)
    assert abs(m + 0.006) < 1e-12 and abs(c - 0.144) < 1e-12
    assert ct.stance_frames(12.0, 120.0) == 9            # 0.072 s * 120 fps = 8.64
    assert ct.stance_frames(9.0, 200.0) == 18
    m, c = ct.line_through(ct.PEAK_FZ_PTS[("B", "leading")])
    assert abs(m * 12.0 + c - 2.35) < 1e-12


def test_helpers_follow_the_reference_index_conventions():
    runs = ct.runs_of_consecutive(np.array([3, 4, 5, 9, 10, 14]))
    assert [r.tolist() for r in runs] == [[3, 4, 5], [9, 10], [14]]
    assert [r.tolist() for r in ct.runs_of_consecutive(np.array([], dtype=int))] == [[]]
    v = np.array([-1.0, -0.5, 0.0, 0.3, 1.0, -1.0, 2.0])
    # zeros are dropped first: [-1, -.5, .3, 1, -1, 2] -> sign changes at 2 and 5 of the compacted series
    assert sorted(ct.upward_crossing_window(v).tolist()) == [0, 1, 2, 3, 3, 4, 4, 5, 6, 7]
    assert ct.upward_crossing_window(np.ones(5)).size == 0


def gait(N=60, fps=120.0, touch=(10, 22, 34, 46), dip=8):
    """foot heights: 0.2 m in swing, a parabolic dip to 0.01 m of 2*dip+1 frames centred on touch[i]; velocity = gradient"""
    n = np.arange(N)
    z = 0.2 + 0.02 * np.sin(0.37 * n[:, None] + np.arange(4)[None, :])       # swing: never exactly at rest (see the zero rule above)
    for i, c in enumerate(touch):
        w = np.abs(n - c) <= dip
        z[w, i] = 0.01 + 0.19 * ((n[w] - c) / dip) ** 2
    vz = np.gradient(z, 1.0 / fps, axis=0)
    return z, vz


def test_contact_windows_are_centred_on_the_lowest_point_and_labelled():
    z, vz = gait()
    contacts, by_height = ct.contact_detection(z, vz, FEET, start_frame=100, speed=12.0, fps=120.0)
    # stance 9 frames (odd) -> lowest point -4 .. +4
    assert contacts["HFL_foot"] == [[106, 114, 0, "trailing"]]
    assert contacts["HFR_foot"] == [[118, 126, 1, "leading"]]
    assert contacts["HBL_foot"] == [[130, 138, 2, "trailing"]]
    assert contacts["HBR_foot"] == [[142, 150, 3, "leading"]]
    # height-only variant: frames with z < 0.05: |n - c| <= 3 (0.01 + 0.19 (k/8)^2 < 0.05 for k <= 3)
    assert by_height["HFL_foot"] == [[107, 113, 0, "TBD"]]
    # an even stance length shifts the first frame by one (speed 9 m/s at 100 fps: 9 frames; at 200 fps: 18)
    contacts, _ = ct.contact_detection(z, vz, FEET, 0, 9.0, 200.0)
    assert contacts["HFR_foot"] == [[22 - 9 + 1, 22 + 9, 1, "leading"]]


def test_contact_windows_are_clamped_to_the_sequence_and_need_a_velocity_sign_change():
    z, vz = gait(touch=(2, 22, 34, 58))
    contacts, _ = ct.contact_detection(z, vz, FEET, 0, 12.0, 120.0)
    assert contacts["HFL_foot"] == [[0, 8, 0, "trailing"]]          # 2-4 < 0 -> shifted right
    # the reference's end clamp (first -= last - N - 1; last = N - 1): lowest 58 -> 54..62 -> first 54-(62-60-1)=53, last 59
    assert contacts["HBR_foot"] == [[53, 59, 3, "leading"]]
    # a foot that only descends (no upward sign change of vz) is not a contact even below the height threshold
    z2 = z.copy(); z2[:, 1] = np.linspace(0.2, 0.0, z.shape[0])
    vz2 = np.gradient(z2, 1 / 120.0, axis=0)
    contacts, by_height = ct.contact_detection(z2, vz2, FEET, 0, 12.0, 120.0)
    assert contacts["HFR_foot"] is None and by_height["HFR_foot"] is None
    assert contacts["HFL_foot"][0][3] == "TBD"                      # no partner -> no leading/trailing decision
    # never below the threshold
    z3 = np.full_like(z, 0.3)
    contacts, _ = ct.contact_detection(z3, vz, FEET, 0, 12.0, 120.0)
    assert all(v is None for v in contacts.values())


def test_two_contacts_of_one_foot():
    z, vz = gait(N=90, touch=(10, 22, 34, 46))
    n = np.arange(90)
    w = np.abs(n - 70) <= 8
    z[w, 0] = 0.01 + 0.19 * ((n[w] - 70) / 8) ** 2
    vz = np.gradient(z, 1 / 120.0, axis=0)
    contacts, by_height = ct.contact_detection(z, vz, FEET, 0, 12.0, 120.0)
    assert [c[:2] for c in contacts["HFL_foot"]] == [[6, 14], [66, 74]]
    assert contacts["HFL_foot"][0][3] == "trailing" and contacts["HFL_foot"][1][3] == "TBD"     # only the first contact is labelled
    assert [c[:2] for c in by_height["HFL_foot"]] == [[7, 13], [67, 73]]


def test_files_and_synthetic_forces(tmp_path):
    z, vz = gait()
    contacts, by_height = ct.contact_detection(z, vz, FEET, 100, 12.0, 120.0)
    grf_dir = os.path.join(str(tmp_path), "grf")
    ct.write_contacts(grf_dir, 100, 60, contacts, by_height)
    with open(os.path.join(grf_dir, "autogen-contact.json")) as f:
        cj = json.load(f)
    assert cj["start_frame"] == 100 and cj["end_frame"] == 160 and cj["contacts"]["HBR_foot"] == [[142, 150, 3, "leading"]]
    plates = ct.synth_grf(cj, FEET, speed=12.0, direction=-1.0)
    assert sorted(plates) == [-1, 0, 1, 2]                           # the reference keys the table by foot index - 1
    F = plates[0]                                                    # HFR: leading fore limb, frames 117 .. 127 -> rows 17 .. 27
    assert F.shape == (60, 3) and not F[:17].any() and not F[27:].any() and not F[:, 1].any()
    peak = 2.0 + (1.8 - 2.0) * (12.0 - 9.0) / 6.0
    n = 10
    t = np.linspace(0, n, n)
    assert np.abs(F[17:27, 2] - peak * np.sin(np.pi * t / n)).max() < 1e-12
    # Fx: quadratic interpolating spline through (0,0) (2,brake) (5,0) (7,push) (10,0): braking first (direction -1 -> negative)
    brake, push = -0.5 * peak, 0.25 * peak
    from scipy.interpolate import InterpolatedUnivariateSpline
    s = InterpolatedUnivariateSpline([0, 2, 5, 7, 10], [0, brake, 0, push, 0], k=2)
    assert np.abs(F[17:27, 0] - s(t)).max() < 1e-12 and F[18, 0] < 0 < F[25, 0]
    assert abs(F[17, 0]) < 1e-12 and abs(F[26, 0]) < 1e-12
    ct.write_synth_grf(os.path.join(grf_dir, "data_synth.csv"), plates)
    rows = np.genfromtxt(os.path.join(grf_dir, "data_synth.csv"), delimiter=",", skip_header=1)
    assert rows.shape == (4 * 60, 5) and np.abs(rows[60:120, 2:] - plates[0]).max() == 0
    # a contact that runs to the end of the sequence is skipped (last >= end_frame), like the reference
    cj["contacts"]["HBR_foot"] = [[150, 160, 3, "leading"]]
    assert 2 not in ct.synth_grf(cj, FEET, 12.0, -1.0)


def _golden():
    here = os.path.dirname(os.path.abspath(__file__))
    G = np.load(os.path.join(here, "golden", "contacts_metrics_golden.npz"))
    with open(os.path.join(here, "golden", "contacts_metrics_lists.json")) as f:
        return G, json.load(f)


def test_helpers_against_the_reference_own_functions():
    """tests/golden/contacts_metrics_*: outputs of acinoset_misc.SimpleLinearModel, positive_zero_crossings,
    group_by_consecutive_values and find_minimum_foot_height CALLED on seeded inputs (tools/gen_golden.py)"""
    G, Lj = _golden()
    for name, pts in (("stance", ct.STANCE_TIME_PTS), ("lfl", ct.PEAK_FZ_PTS[("F", "leading")]), ("lhl", ct.PEAK_FZ_PTS[("B", "leading")]),
                      ("nlfl", ct.PEAK_FZ_PTS[("F", "trailing")]), ("nlhl", ct.PEAK_FZ_PTS[("B", "trailing")])):
        m, c = ct.line_through(pts)
        assert np.abs(np.array([m, c, m * 11.3 + c]) - G[f"line_{name}"]).max() < 1e-13, name
    for k in range(4):
        got = ct.upward_crossing_window(G["zc_series"][k])
        want = Lj[f"zc_{k}"]
        assert sorted(got.tolist()) == sorted(want["idx"]) and len(got) == 5 * want["count"] and want["count"] > 0
    for k in range(4):
        rec = Lj[f"runs_{k}"]
        assert [r.tolist() for r in ct.runs_of_consecutive(np.array(rec["inp"], dtype=int))] == rec["out"]
    # find_minimum_foot_height = region start + argmin over the slice: the expression inside contact_detection
    h = G["minh_series"]
    for (lo, hi), want in zip(G["minh_regions"], G["minh_out"]):
        assert lo + int(np.argmin(h[lo:hi])) == want


def test_trajectory_metrics_against_the_reference_own_functions():
    from cheetah_pose_estimation_amd import metrics
    G, _ = _golden()
    X, Y = G["traj_X"], G["traj_Y"]
    assert abs(metrics.traj_smoothness(X, Y) - float(G["traj_smoothness"])) < 1e-15
    for centered, tag in ((False, "u"), (True, "c")):
        X0, Y0 = X.copy(), Y.copy()
        per_marker, per_frame, smooth = metrics.traj_error(X, Y, centered=centered)
        assert np.array_equal(X, X0) and np.array_equal(Y, Y0)                      # inputs untouched
        assert list(per_marker) == list(__import__("cheetah_pose_estimation_amd").skeleton.MARKERS)
        assert np.abs(np.array(list(per_marker.values())) - G[f"traj_mpjpe_{tag}"]).max() < 1e-11
        assert np.abs(per_frame - G[f"traj_frame_{tag}"]).max() < 1e-11
        assert abs(smooth - float(G[f"traj_smooth_{tag}"])) < 1e-11
    M = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "misc_golden.npz"))
    assert abs(metrics.rmse(M["metric_a"], M["metric_b"]) - float(M["metric_rmse"])) < 1e-15
    a = M["metric_a"].copy(); a[0, 0, 0] = np.nan
    assert np.isfinite(metrics.rmse(a, M["metric_b"]))


def test_assembled_rule_and_force_templates_against_the_reference_own_functions(tmp_path):
    """acinoset_misc.contact_detection and synth_grf_data themselves, run by tools/gen_golden.py on seeded foot-height / velocity
    series with stand-ins for the Pyomo accessors only (feet objects, get_vals, ground plane 0.0, to_hdf captured): same contact
    windows, labels, JSON files and force tables."""
    G, Lj = _golden()
    for case, rec in enumerate(Lj["contact_cases"]):
        z, vz = G[f"cd_height_{case}"], G[f"cd_velz_{case}"]
        contacts, by_height = ct.contact_detection(z, vz, FEET, rec["start_frame"], rec["speed"], rec["fps"])
        assert contacts == rec["contacts"], case
        assert by_height == rec["by_height"], case
        grf_dir = os.path.join(str(tmp_path), f"grf{case}")
        ct.write_contacts(grf_dir, rec["start_frame"], rec["N"], contacts, by_height)
        with open(os.path.join(grf_dir, "autogen-contact.json")) as f:
            cj = json.load(f)
        assert cj == rec["json"]
        plates = ct.synth_grf(cj, FEET, rec["speed"], rec["direction"])
        assert sorted(str(k) for k in plates) == sorted(rec["plates"])
        for k, F in plates.items():
            assert np.abs(F - np.array(rec["plates"][str(k)])).max() < 1e-12, (case, k)
    assert any(len(v) > 1 for rec in Lj["contact_cases"] for v in rec["contacts"].values() if v)      # a foot with two contacts is covered


# ---- pin on the reference's own STORED outputs ------------------------------------------------------------------
def _stored_run():
    """tests/golden/contacts_pin.npz (tools/pin_contacts_from_csv.py): the joint angles of the reference's monocular solution
    `2019_03_07/phantom/run/fte_kinematic_1`, recovered from its stored 2D files cam{1..6}_fte.csv (worst pixel error 1.4e-5) with the
    cameras of tests/golden/fk_csv_pin.npz, a ground plane fitted to the lowest paw positions (the calibration files that define the true
    world frame are not shipped), and the contents of the stored `grf/autogen-contact.json` / `-02.json`."""
    import os
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "contacts_pin.npz"))
    q = Z["q"]
    dq = np.zeros_like(q); dq[1:] = (q[1:] - q[:-1]) * 120.0; dq[0] = dq[1]     # implicit Euler; q'_0 is a free variable of the reference's NLP (not stored in 2D)
    return Z, q, dq


def _detect_in_fitted_frame(Z, pos, vel):
    from cheetah_pose_estimation_amd import skeleton
    up, off = Z["ground_normal"], float(Z["ground_offset"])
    feet = [skeleton.MARKERS.index(m) for m in skeleton.FOOT_MARKERS]
    names = [f"{f}_foot" for f in skeleton.FEET]
    com = pos.mean(1)                                                            # speed enters only through the stance-time line: any body point serves
    speed = float(np.linalg.norm(np.diff(com, axis=0) * 120.0, axis=1).mean())
    return ct.contact_detection(pos[:, feet] @ up - off, vel[:, feet] @ up, names, int(Z["start_frame"]), speed, 120.0), names


def test_contact_heuristic_reproduces_the_reference_stored_json(oracle):
    """FK + analytic foot velocity (oracle) + contact_detection on the reference's stored monocular solution give the windows and the
    leading / trailing labels the reference itself wrote to grf/autogen-contact.json: start 135, end 192, HFL [168, 180] leading,
    HFR [157, 169] trailing, HBL [155, 167] leading, HBR [144, 156] trailing -- exactly, and unchanged by +-2 cm of ground offset."""
    from cheetah_pose_estimation_amd import skeleton
    Z, q, dq = _stored_run()
    sk = skeleton.build_skeleton("phantom", 24)
    pos = oracle.markers(sk, q)
    vel = np.array([np.einsum("ldp,p->ld", oracle.markers_jac(sk, q[n])[1], dq[n]) for n in range(q.shape[0])])
    (contacts, by_height), names = _detect_in_fitted_frame(Z, pos, vel)
    assert int(Z["start_frame"]) == 135 and int(Z["end_frame"]) == 192 and q.shape[0] == 57
    for i, n in enumerate(names):
        assert len(contacts[n]) == 1 == int(Z["n_windows"][i])
        assert contacts[n][0][:2] == [int(Z["windows"][i][0]), int(Z["windows"][i][1])], (n, contacts[n])
        assert contacts[n][0][3] == str(Z["labels"][i])
    assert [contacts[n][0][:2] for n in names] == [[168, 180], [157, 169], [155, 167], [144, 156]]
    for shift in (-0.02, 0.02):                                                  # the fitted plane is within centimetres of the true one: the result does not depend on that
        Z2 = dict(Z); Z2["ground_offset"] = float(Z["ground_offset"]) + shift
        (c2, _), _ = _detect_in_fitted_frame(Z2, pos, vel)
        assert [c2[n][0][:2] for n in names] == [[168, 180], [157, 169], [155, 167], [144, 156]]


def test_contact_heuristic_on_a_second_stored_run(oracle):
    """Second sequence, other animal, other frame rate: `2017_08_29/top/jules/run1_1` (90 fps, 30 frames), fixture
    tests/golden/contacts_pin_jules.npz (tools/pin_contacts_from_csv.py ... stance 90; worst pixel error 1.9e-5).  The run is short and
    its ends dip below the stance height, so the ground plane is fitted through the lowest interior minimum of every paw's height trace.
    Stored JSON: start 53, end 83, HFL [58, 64] leading, HFR [54, 60] trailing, HBL [69, 75] trailing, HBR [72, 78] leading.  Three
    windows and all four labels are reproduced exactly; HFR's two lowest frames differ by 5e-5 m in the fitted frame, so its window is
    pinned to within one frame (which of the two is lower is decided by the unshipped calibration's true ground plane)."""
    import os
    from cheetah_pose_estimation_amd import skeleton
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "contacts_pin_jules.npz"))
    q, fps = Z["q"], float(Z["fps"])
    assert fps == 90.0 and q.shape[0] == 30 and int(Z["start_frame"]) == 53 and int(Z["end_frame"]) == 83
    dq = np.zeros_like(q); dq[1:] = (q[1:] - q[:-1]) * fps; dq[0] = dq[1]
    sk = skeleton.build_skeleton("jules", 24)
    pos = oracle.markers(sk, q)
    vel = np.array([np.einsum("ldp,p->ld", oracle.markers_jac(sk, q[n])[1], dq[n]) for n in range(q.shape[0])])
    up, off = Z["ground_normal"], float(Z["ground_offset"])
    feet = [skeleton.MARKERS.index(m) for m in skeleton.FOOT_MARKERS]
    names = [f"{f}_foot" for f in skeleton.FEET]
    speed = float(np.linalg.norm(np.diff(pos.mean(1), axis=0) * fps, axis=1).mean())
    assert ct.stance_frames(speed, fps) == 7                                     # every stored window spans 7 frames
    for shift in (-0.02, 0.0, 0.02):
        contacts, by_height = ct.contact_detection(pos[:, feet] @ up - off - shift, vel[:, feet] @ up, names, 53, speed, fps)
        for i, n in enumerate(names):
            assert len(contacts[n]) == 1 == int(Z["n_windows"][i])
            assert contacts[n][0][3] == str(Z["labels"][i]), (n, contacts[n])
            d = np.array(contacts[n][0][:2]) - Z["windows"][i]
            assert d[0] == d[1] and abs(int(d[0])) <= (1 if n == "HFR_foot" else 0), (n, contacts[n], Z["windows"][i])
            if shift == 0.02:                                                    # the below-threshold runs measure the ground offset itself: 2 cm lower matches the stored ones to a frame
                assert np.abs(np.array(by_height[n][0][:2]) - Z["windows_height_only"][i]).max() <= 1, (n, by_height[n])


@pytest.mark.parametrize("fixture,animal,n_frames,start,end,stance,windows,labels,shifts,loose", [
    ("contacts_pin_phantom2017.npz", "phantom", 44, 59, 103, 9, [[84, 92], [75, 83], [76, 84], [67, 75]], ["leading", "trailing", "leading", "trailing"], (-0.02, 0.0, 0.02), None),
    ("contacts_pin_jules2.npz", "jules", 34, 80, 114, 8, [[103, 110], [97, 104], [83, 90], [87, 94]], ["leading", "trailing", "trailing", "leading"], (-0.02, 0.0, 0.02), None),
    # third rig (2017_09_02/top; cameras of tests/golden/fk_csv_pin_0902top.npz, recovered from the MULTI-view result of jules/run1): the monocular
    # result of the same run (fte_kinematic_0, 2.7e-6 px) and the phantom run of that day (run1_2, fte_kinematic_1, 1.7e-5 px)
    ("contacts_pin_0902_jules.npz", "jules", 30, 68, 98, 6, [[89, 94], [85, 90], [72, 77], [74, 79]], ["leading", "trailing", "trailing", "leading"], (-0.02, 0.0, 0.02), None),
    ("contacts_pin_0902_phantom.npz", "phantom", 45, 39, 84, 9, [[58, 66], [67, 75], [49, 57], [54, 62]], ["trailing", "leading", "trailing", "leading"], (-0.02, 0.0, 0.02), None),
    # fourth rig (2017_09_02/bottom/jules/run2; cameras of fk_csv_pin_0902bot.npz from its multi-view result, monocular result fte_kinematic_1 at 1.5e-5 px)
    ("contacts_pin_0902bot.npz", "jules", 33, 91, 124, 8, [[112, 119], [108, 115], [96, 103], [97, 104]], ["leading", "trailing", "trailing", "leading"], (-0.02, 0.0, 0.02), None),
    # 2017_12_09/bottom/jules/flick2 (a 5-camera scene: camera 6 has no stored file; fk_csv_pin_1209.npz; fte_kinematic_4 at 1.9e-4 px): a flick, 5-frame stance
    ("contacts_pin_1209.npz", "jules", 30, 19, 49, 5, [[42, 46], [39, 43], [28, 32], [25, 29]], ["leading", "trailing", "leading", "trailing"], (-0.02, 0.0, 0.02), None),
    # 2019_03_09/jules/flick1 (120 fps, camera 2 partly outside the image; fk_csv_pin_0309.npz; fte_kinematic_4 at 2.1e-5 px): exact on the fitted
    # plane and 2 cm below it; 2 cm above it the right fore paw's lowest point no longer reaches the 5 cm threshold
    ("contacts_pin_0309.npz", "jules", 34, 125, 159, 8, [[143, 150], [141, 148], [127, 134], [131, 138]], ["leading", "trailing", "trailing", "leading"], (0.0, 0.02), None),
    # 2019_03_03/phantom/run (a 4-camera scene: cameras 4 and 5 have no stored files; fk_csv_pin_0303.npz; fte_kinematic_1 at 7e-4 px): the left fore
    # paw's two lowest frames differ by 2 mm in the fitted frame: its window within one frame, everything else exact
    ("contacts_pin_0303.npz", "phantom", 42, 152, 194, 12, [[172, 183], [180, 191], [156, 167], [162, 173]], ["trailing", "leading", "trailing", "leading"], (-0.01, 0.0, 0.02), "HFL_foot"),
])
def test_contact_heuristic_on_stored_runs_with_cameras_from_another_sequence(oracle, fixture, animal, n_frames, start, end, stance, windows, labels, shifts, loose):
    """`2017_08_29/top/phantom/run1_1` (fte_kinematic_4) and `2017_08_29/top/jules/run1_2` (fte_kinematic_1), 90 fps: the cameras are NOT fitted to these
    sequences -- they are the ones recovered from jules run1_1 of the same day and rig (tests/golden/fk_csv_pin_jules.npz) -- and this repository's
    FK + link lengths of the animal + joint equalities still reproduce the reference's stored monocular 2D files cam{1..6}_fte.csv to 2.2e-5 px
    (phantom) / 2.3e-4 px (jules run1_2) worst (fixture worst_px; tools/pin_contacts_from_csv.py <seq> <animal> fk_csv_pin_jules.npz <dir> <out>
    stance 90): an FK / marker / projection pin with no camera freedom left.  On the recovered angles the contact heuristic gives the stored
    `grf/autogen-contact.json` EXACTLY -- start and end frame, every window, every leading / trailing label, the stance length at the run's
    speed -- unchanged by +-2 cm of ground offset."""
    import os
    from cheetah_pose_estimation_amd import skeleton
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))
    q, fps = Z["q"], float(Z["fps"])
    assert fps in (90.0, 120.0) and q.shape[0] == n_frames and int(Z["start_frame"]) == start and int(Z["end_frame"]) == end and float(Z["worst_px"]) < 1e-3
    dq = np.zeros_like(q); dq[1:] = (q[1:] - q[:-1]) * fps; dq[0] = dq[1]
    sk = skeleton.build_skeleton(animal, 24)
    pos = oracle.markers(sk, q)
    vel = np.array([np.einsum("ldp,p->ld", oracle.markers_jac(sk, q[n])[1], dq[n]) for n in range(q.shape[0])])
    up, off = Z["ground_normal"], float(Z["ground_offset"])
    feet = [skeleton.MARKERS.index(m) for m in skeleton.FOOT_MARKERS]
    names = [f"{f}_foot" for f in skeleton.FEET]
    speed = float(np.linalg.norm(np.diff(pos.mean(1), axis=0) * fps, axis=1).mean())
    assert ct.stance_frames(speed, fps) == stance
    assert windows == [list(map(int, w)) for w in Z["windows"]] and labels == [str(l) for l in Z["labels"]]      # the fixture holds the stored file
    for shift in shifts:
        contacts, _ = ct.contact_detection(pos[:, feet] @ up - off - shift, vel[:, feet] @ up, names, start, speed, fps)
        for i, n in enumerate(names):
            assert len(contacts[n]) == 1 and contacts[n][0][3] == labels[i], (n, contacts[n])
            d = np.array(contacts[n][0][:2]) - np.array(windows[i])
            assert d[0] == d[1] and abs(int(d[0])) <= (1 if n == loose else 0), (n, contacts[n], windows[i])
