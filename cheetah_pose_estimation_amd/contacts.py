"""Contact heuristic and synthetic ground-reaction-force templates.

Host-side decision logic over per-foot time series of a few hundred samples; the series themselves (foot height and
analytic foot velocity) come from the device (cpe_forward_kinematics / cpe_marker_velocities).  Follows
acinoset_misc.py:745-862 (contact_detection), :865-943 (synth_grf_data) and the helpers :69-81, :2033-2057, including
their index conventions, so that the files written here can be consumed where the reference's are.

`pe.foot.Foot3D.ground_plane_height` lives in the un-vendored `physical_education` submodule; the ground is z = 0 in
every stored reconstruction, which is the default used here (parity of that constant: unpinned).  Everything else is pinned:
tests/golden/contacts_metrics_* hold the outputs of the reference's own contact_detection, synth_grf_data and helpers on seeded
series (tools/gen_golden.py), and tests/test_contacts.py compares this module with them.
"""
import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

HEIGHT_THRESHOLD = 0.05                 # m above the ground plane (acinoset_misc.py:757)
STANCE_TIME_PTS = ((9.0, 0.09), (14.0, 0.06))           # speed [m/s] -> stance time [s] (:749)
PEAK_FZ_PTS = {                          # speed -> peak vertical force [body weights] (:876-879)
    ("F", "leading"): ((9.0, 2.0), (15.0, 1.8)),
    ("B", "leading"): ((9.0, 2.1), (15.0, 2.6)),
    ("F", "trailing"): ((9.5, 2.1), (15.0, 2.0)),
    ("B", "trailing"): ((9.0, 1.7), (15.0, 2.5)),
}


def line_through(pts) -> Tuple[float, float]:
    """least-squares line (slope, intercept) through the points -- SimpleLinearModel, acinoset_misc.py:69-81"""
    x, y = np.asarray(pts, dtype=float).T
    slope, icpt = np.linalg.lstsq(np.stack([x, np.ones_like(x)], axis=1), y, rcond=None)[0]
    return float(slope), float(icpt)


def stance_frames(speed: float, fps: float) -> int:
    m, c = line_through(STANCE_TIME_PTS)
    return int(round((m * speed + c) * fps))


def runs_of_consecutive(idx: np.ndarray) -> List[np.ndarray]:
    """split a sorted index list where it jumps by more than one (:2049-2051); an empty input gives one empty run"""
    idx = np.asarray(idx)
    cuts = np.nonzero(np.diff(idx) > 1)[0] + 1
    return np.split(idx, cuts)


def upward_crossing_window(v: np.ndarray) -> np.ndarray:
    """indices within +-2 of a negative -> positive sign change of v, counted in the series with exact zeros removed
    (:2033-2046)"""
    x = v[v != 0]
    i = np.nonzero((x[:-1] < 0) & (x[1:] > 0))[0] + 1
    return (i[:, None] + np.arange(-2, 3)[None, :]).reshape(-1) if i.size else np.zeros(0, dtype=int)


def contact_detection(foot_height: np.ndarray, foot_vel_z: np.ndarray, foot_names: Sequence[str], start_frame: int,
                      speed: float, fps: float, ground_plane_height: float = 0.0):
    """foot_height, foot_vel_z: [N, n_feet].  Returns (contacts, contacts_height_only): {foot: [[first, last, foot
    index, label], ...] or None}.  Rule (acinoset_misc.py:745-842): a run of frames with the foot below the height
    threshold is a contact if the vertical foot velocity changes sign upwards within +-2 frames of the lowest point;
    the contact window is the expected stance duration at this speed centred on the lowest point."""
    N = foot_height.shape[0]
    stance = stance_frames(speed, fps)
    half, even = stance // 2, stance % 2 == 0
    contacts: Dict[str, Optional[list]] = {}
    by_height: Dict[str, Optional[list]] = {}
    for i, name in enumerate(foot_names):
        z = foot_height[:, i]
        runs = runs_of_consecutive(np.nonzero(z < ground_plane_height + HEIGHT_THRESHOLD)[0])
        crossing = upward_crossing_window(foot_vel_z[:, i])
        found, found_h = [], []
        lowest = -1
        for j, run in enumerate(runs):
            if run.size == 0:
                continue
            lo = lowest + 1
            hi = int(runs[j + 1][0]) if j + 1 < len(runs) else -1
            window = z[lo:hi] if z[lo:hi].size else z[lo:]      # (the reference's slice is empty, and raises, if a run starts at the last frame)
            lowest = lo + int(np.argmin(window))
            near = np.intersect1d(run, crossing)
            hit = any((lowest + k) in near for k in (-2, -1, 0, 1, 2))
            first, last = lowest - half + (1 if even else 0), lowest + half
            lowest = int(run[-1])               # the next search starts after this run
            if not hit:
                continue
            if first < 0:
                last -= first
                first = 0
            if last >= N:
                first -= last - N - 1
                last = N - 1
            found.append([int(start_frame + first), int(start_frame + last), i, "TBD"])
            found_h.append([int(start_frame + run[0]), int(start_frame + run[-1]), i, "TBD"])
        contacts[name] = found or None
        by_height[name] = found_h or None
    # rotary gallop: of a fore (hind) pair the limb that lands first is the trailing one (:846-862)
    for a, b in ((0, 1), (2, 3)):
        if len(foot_names) > max(a, b):
            ca, cb = contacts[foot_names[a]], contacts[foot_names[b]]
            if ca is not None and cb is not None:
                a_later = ca[0][0] > cb[0][0]
                ca[0][3] = "leading" if a_later else "trailing"
                cb[0][3] = "trailing" if a_later else "leading"
    return contacts, by_height


def write_contacts(grf_dir: str, start_frame: int, n_frames: int, contacts, by_height):
    """grf/autogen-contact.json and grf/autogen-contact-02.json (acinoset_misc.py:845-860)"""
    os.makedirs(grf_dir, exist_ok=True)
    for fname, c in (("autogen-contact.json", contacts), ("autogen-contact-02.json", by_height)):
        with open(os.path.join(grf_dir, fname), "w", encoding="utf-8") as f:
            json.dump({"start_frame": int(start_frame), "end_frame": int(start_frame + n_frames), "contacts": c}, f)


def synth_grf(contact_json: dict, foot_names: Sequence[str], speed: float, direction: float):
    """Template forces for the first contact of every foot (acinoset_misc.py:865-943): half-sine Fz with a
    speed-dependent peak, a braking then propulsive Fx (peaks 50 % of Fz and 50 % of that) through a quadratic
    interpolating spline, Fy = 0.  Returns {plate key: array [n_frames, 3] = (Fx, Fy, Fz)} in body weights; the plate
    key is the reference's `foot index - 1`."""
    from scipy.interpolate import InterpolatedUnivariateSpline
    start, end = contact_json["start_frame"], contact_json["end_frame"]
    out = {}
    for name in foot_names:
        rec = contact_json["contacts"].get(name)
        if rec is None or rec[0][1] >= end:
            continue
        first, last, idx, label = rec[0]
        a, b = max(first - 1, start), min(last + 1, end)
        n = b - a
        mid = n // 2
        t = np.linspace(0, n, n)
        pts = PEAK_FZ_PTS.get(("F" if "F" in name else "B", label))
        peak = 0.0
        if pts is not None:
            m, c = line_through(pts)
            peak = m * speed + c
        brake = direction * 0.5 * peak
        push = -0.5 * brake
        knots = np.array([[0, 0.0], [mid // 2, brake], [mid, 0.0], [mid + (n - mid) // 2, push], [n, 0.0]])
        fx = InterpolatedUnivariateSpline(knots[:, 0], knots[:, 1], k=2)(t)
        F = np.zeros((end - start, 3))
        F[a - start:b - start, 0] = fx
        F[a - start:b - start, 2] = peak * np.sin(np.pi * t / n)
        out[idx - 1] = F
    return out


def write_synth_grf(path_csv: str, plates: Dict[int, np.ndarray]):
    """The reference stores a (force_plate, frame) x (Fx, Fy, Fz) table as HDF5 (`to_hdf`, acinoset_misc.py:940-943); PyTables is not
    available here, so the same table is written as CSV with the same index and column names."""
    os.makedirs(os.path.dirname(path_csv), exist_ok=True)
    with open(path_csv, "w", encoding="utf-8") as f:
        f.write("force_plate,frame,Fx,Fy,Fz\n")
        for key, F in plates.items():
            for n, row in enumerate(F):
                f.write(f"{key},{n},{float(row[0])!r},{float(row[1])!r},{float(row[2])!r}\n")
