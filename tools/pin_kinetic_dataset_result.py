#!/usr/bin/env python3
"""Recover the joint angles of a stored PHYSICS-BASED result of the kinetic dataset (build container only).

`data/test_set/kinetic_dataset/<day>/<animal>/<trial>/fte_kinetic/cam{1..4}_fte.csv` (and `fte_grf/`) are the 2D reprojections of the trajectory the
reference's physics-based stages found (run_kinetic, run_dataset.py:1092-1140).  The cameras of the trial are already recovered from its KINEMATIC
result (tests/golden/fk_csv_pin_<animal>.npz, tools/pin_fk_pinhole.py) and are held fixed here: per frame only the pose is free (trunk angles + leg
angles, the solver's own coordinates), fitted to the <= 192 stored pixel values.  Output: tests/golden/<out> with q, the worst pixel error, the frame
rate; tests/test_free_flight_pin.py reads it.
usage: python tools/pin_kinetic_dataset_result.py <sequence> <animal> <subdir, e.g. fte_kinetic> <out.npz> [fps]"""
import os
import sys

import numpy as np
from scipy.optimize import least_squares

SEQ_, ANIMAL_, SUB, OUT_ = sys.argv[1:5]
FPS = float(sys.argv[5]) if len(sys.argv) > 5 else 200.0
sys.argv = [sys.argv[0], SEQ_, ANIMAL_, OUT_]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pin_fk_pinhole as P  # noqa: E402
from cheetah_pose_estimation_amd import skeleton, synth  # noqa: E402


def main():
    Z = np.load(os.path.join(P.ROOT, "tests", "golden", f"fk_csv_pin_{ANIMAL_}.npz"))
    cams = Z["cams"]
    src = f"/root/reference/data/test_set/{SEQ_}/{SUB}"
    arrs = []
    for c in range(1, 5):
        rows = np.genfromtxt(os.path.join(src, f"cam{c}_fte.csv"), delimiter=",", skip_header=2)
        arrs.append(rows[:, 1:].reshape(len(rows), 24, 3)[:, :, :2])
    uv = np.stack(arrs, 1)
    N = uv.shape[0]
    ok = ~np.isnan(uv).any(-1)
    sk = skeleton.build_skeleton(f"{ANIMAL_}-02", 24, kinetic_dataset=True)
    lay = synth.leg_layout(sk)
    ind = skeleton.independent_dofs(sk)
    trunk = [p for p in ind if not any(p == 3 + 3 * c + 1 for c, _ in lay)]

    def q_of(x, psi_ref):
        q = np.zeros((1, sk.nq)); q[0, trunk] = x[:len(trunk)]
        for i in range(sk.n_links):
            if (3 + 3 * i + 2) not in trunk:
                q[0, 3 + 3 * i + 2] = psi_ref
        q = synth.legs_from_alpha(sk, q, x[None, len(trunk):])
        return synth.project_dependents_numpy_hooke(sk, q)[0]

    def alpha_of(q):
        al = np.zeros(len(lay))
        for r, (c, B) in enumerate(lay):
            M = synth.rot_zyx(q[3 + 3 * B:6 + 3 * B]).T @ synth.rot_zyx(q[3 + 3 * c:6 + 3 * c])
            al[r] = np.arctan2(M[0, 2], M[0, 0])
        return al
    q_out = np.zeros((N, sk.nq)); errs = np.zeros(N)
    for n in range(N):
        starts = [Z["q"][min(n, len(Z["q"]) - 1)]] + ([q_out[n - 1]] if n else [])
        best = None
        for q0 in starts:
            f = lambda xx: np.concatenate([np.where(ok[n, c][:, None], P.project(cams[c], synth.fk_numpy(sk, q_of(xx, q0[5])[None])[0][0]) - np.nan_to_num(uv[n, c]), 0.0).ravel() for c in range(4)])
            s = least_squares(f, np.concatenate([q0[trunk], alpha_of(q0)]), method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=4000)
            if best is None or np.abs(s.fun).max() < np.abs(best[0].fun).max():
                best = (s, q0[5])
            if np.abs(best[0].fun).max() < 1e-6:
                break
        q_out[n] = q_of(best[0].x, best[1]); errs[n] = float(np.abs(best[0].fun).max())
        print(f"frame {n}: max |pixel error| {errs[n]:.3e} over {int(ok[n].sum())} visible points", flush=True)
    out = os.path.join(P.ROOT, "tests", "golden", OUT_)
    np.savez_compressed(out, uv=uv, q=q_out, cams=cams, worst_px=float(errs.max()), fps=FPS, seq=SEQ_, animal=ANIMAL_, sub=SUB, visible=ok.sum((1, 2)))
    print("worst pixel error", errs.max(), "wrote", out)


if __name__ == "__main__":
    main()
