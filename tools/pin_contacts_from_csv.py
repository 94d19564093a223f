#!/usr/bin/env python3
"""Pin the contact heuristic end to end on the reference's STORED outputs (build container only).

`data/test_set/2019_03_07/phantom/run/grf/autogen-contact.json` is what the reference's determine_contacts wrote
(acinoset_opt.py:638-692 -> acinoset_misc.contact_detection) from its monocular solution `fte_kinematic_1`.  That solution is
stored as 2D text too: `fte_kinematic_1/cam{1..6}_fte.csv` = its 24 markers projected into all six cameras of the scene.  The
cameras of this scene are already recovered (tests/golden/fk_csv_pin.npz, tools/pin_fk_from_csv.py), so the joint angles of the
monocular solution follow per frame from 288 pixel values.  This script recovers them, expresses them in a z-up frame whose
ground plane is fitted to the lowest paw positions (the true world frame is in calibration files the reference does not ship)
and stores q, the frame and the JSON's contents (data) in tests/golden/contacts_pin.npz.  tests/test_contacts.py and the GPU
test then run FK -> foot heights / analytic foot velocities -> contact_detection and compare with the stored windows.
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from cheetah_pose_estimation_amd import skeleton, synth  # noqa: E402
import pin_fk_from_csv as P  # noqa: E402  (project, rodrigues)

# usage: pin_contacts_from_csv.py [sequence animal camera_fixture monocular_dir output_fixture plane fps]
#   plane = "lowest": ground plane through the four lowest positions of every paw (long runs: every paw has a clean stance);
#   plane = "stance": through the lowest interior local minimum of every paw's height trace (short runs whose ends dip below the stance height)
_A = sys.argv[1:] + ["2019_03_07/phantom/run", "phantom", "fk_csv_pin.npz", "fte_kinematic_1", "contacts_pin.npz", "lowest", "120"][len(sys.argv) - 1:]
SEQ, ANIMAL, CAM_FIXTURE, MONO, OUT, PLANE, FPS = _A[:7]
SRC = f"/root/reference/data/test_set/{SEQ}"


def load_uv(sub):
    arrs = []
    for c in range(1, 7):
        f = os.path.join(SRC, sub, f"cam{c}_fte.csv")
        if not os.path.exists(f):      # camera outside the stored result (4- and 5-camera scenes): empty pixels
            arrs.append(None); continue
        rows = np.genfromtxt(f, delimiter=",", skip_header=2)
        arrs.append(rows[:, 1:].reshape(len(rows), 24, 3)[:, :, :2])
    shape = next(a for a in arrs if a is not None).shape
    return np.stack([a if a is not None else np.full(shape, np.nan) for a in arrs], 1), int(rows[0, 0])


def main():
    Z = np.load(os.path.join(ROOT, "tests", "golden", CAM_FIXTURE))
    cams = Z["cams"]
    uv, idx0 = load_uv(MONO)
    N = uv.shape[0]
    sk = skeleton.build_skeleton(ANIMAL, 24)
    lay = synth.leg_layout(sk)
    ind = skeleton.independent_dofs(sk)
    trunk = [p for p in ind if not any(p == 3 + 3 * c + 1 for c, _ in lay)]       # independent Euler dofs that are not leg pitches

    def q_of(x, psi_ref):
        q = np.zeros((1, sk.nq)); q[0, trunk] = x[:len(trunk)]
        for i in range(sk.n_links):
            if (3 + 3 * i + 2) not in trunk:
                q[0, 3 + 3 * i + 2] = psi_ref
        q = synth.legs_from_alpha(sk, q, x[None, len(trunk):])
        qh = synth.project_dependents_numpy(sk, q)                                   # hooke joints (tails): phi in closed form
        for j in range(sk.n_joints):
            if sk.joint_kind[j] == 1:
                q[0, 3 + 3 * sk.joint_child[j]] = qh[0, 3 + 3 * sk.joint_child[j]]
        return q[0]

    def alpha_of(q):
        al = np.zeros(len(lay))
        for r, (c, B) in enumerate(lay):
            RB = synth.rot_zyx(q[3 + 3 * B:6 + 3 * B]); Rc = synth.rot_zyx(q[3 + 3 * c:6 + 3 * c])
            M = RB.T @ Rc
            al[r] = np.arctan2(M[0, 2], M[0, 0])
        return al

    q_out = np.zeros((N, sk.nq)); worst = 0.0
    errs = np.zeros(N)
    x = None
    for n in range(N):
        zq = Z["q"][min(n, len(Z["q"]) - 1)]                   # (the camera fixture may come from ANOTHER sequence of the same rig: any pose will do as a start)
        q0 = zq if x is None else q_out[n - 1]                 # multi-view solution of the same frames / previous frame as the start
        x0 = np.concatenate([q0[trunk], alpha_of(q0)])
        best = None
        for start in (x0, np.concatenate([zq[trunk], alpha_of(zq)])):
            f = lambda xx: np.concatenate([np.nan_to_num(P.project(cams[c], synth.fk_numpy(sk, q_of(xx, q0[5])[None])[0][0]) - uv[n, c]).ravel() for c in range(6)])     # (NaN = stored pixel outside the image)
            s = least_squares(f, start, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=4000)
            if best is None or np.abs(s.fun).max() < np.abs(best.fun).max():
                best = s
            if np.abs(best.fun).max() < 1e-6:
                break
        q_out[n] = q_of(best.x, q0[5]); x = best.x
        errs[n] = float(np.abs(best.fun).max())
        print(f"frame {n}: max |pixel error| {errs[n]:.3e}", flush=True)
    # frames the forward sweep left in a local minimum (a monocular result can start far from the multi-view one): once more, backwards, from the
    # recovered pose of the following frame
    for n in range(N - 2, -1, -1):
        if errs[n] > 1e-3 and errs[n + 1] <= 1e-3:
            qn = q_out[n + 1]
            f = lambda xx: np.concatenate([np.nan_to_num(P.project(cams[c], synth.fk_numpy(sk, q_of(xx, qn[5])[None])[0][0]) - uv[n, c]).ravel() for c in range(6)])
            s2 = least_squares(f, np.concatenate([qn[trunk], alpha_of(qn)]), method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=4000)
            if np.abs(s2.fun).max() < errs[n]:
                q_out[n] = q_of(s2.x, qn[5]); errs[n] = float(np.abs(s2.fun).max())
                print(f"frame {n} (backward): max |pixel error| {errs[n]:.3e}", flush=True)
    worst = float(errs.max())
    print("worst pixel error", worst)
    # z-up frame: up = normal of the plane through the lowest positions of the four paws, offset so that those lie at z = 0
    pos = synth.fk_numpy(sk, q_out)[0]
    feet = [skeleton.MARKERS.index(m) for m in skeleton.FOOT_MARKERS]
    up = np.array([0, 0, 1.0]); off = 0.0
    for _ in range(5):
        hgt = pos[:, feet] @ up
        if PLANE == "stance":
            pts = []
            for k in range(4):
                h = hgt[:, k]
                loc = [n for n in range(1, N - 1) if h[n] < h[n - 1] and h[n] <= h[n + 1]]
                pts.append(pos[min(loc, key=lambda n: h[n]), feet[k]])
            pts = np.array(pts)
        else:
            pts = np.concatenate([pos[np.argsort(hgt[:, k])[:4], feet[k]] for k in range(4)])
        c = pts.mean(0)
        w = np.linalg.svd(pts - c)[2][-1]
        up = w if w @ up > 0 else -w
        off = float(c @ up)
    print("ground normal in the recovered frame", up, "offset", off)
    with open(os.path.join(SRC, "grf", "autogen-contact.json")) as fh:
        cj = json.load(fh)
    with open(os.path.join(SRC, "grf", "autogen-contact-02.json")) as fh:
        cj2 = json.load(fh)
    names = [f"{f}_foot" for f in skeleton.FEET]
    win = np.array([[cj["contacts"][n][0][0], cj["contacts"][n][0][1]] if cj["contacts"][n] else [-1, -1] for n in names])
    lab = np.array([cj["contacts"][n][0][3] if cj["contacts"][n] else "" for n in names])
    win2 = np.array([[cj2["contacts"][n][0][0], cj2["contacts"][n][0][1]] if cj2["contacts"][n] else [-1, -1] for n in names])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", OUT), uv=uv, q=q_out, cams=cams, ground_normal=up, ground_offset=off,
                        start_frame=cj["start_frame"], end_frame=cj["end_frame"], windows=win, labels=lab, windows_height_only=win2,
                        first_index=idx0, worst_px=worst, fps=float(FPS), plane=PLANE, seq=SEQ, animal=ANIMAL, n_windows=np.array([len(cj["contacts"][n] or []) for n in names]))
    print("stored windows", win.tolist(), lab.tolist(), "height-only", win2.tolist())


if __name__ == "__main__":
    main()
