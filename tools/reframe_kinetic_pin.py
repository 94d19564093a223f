#!/usr/bin/env python3
"""Choose the WORLD FRAME of the kinetic-dataset pins (build container only).

tools/pin_fk_pinhole.py recovers cameras and joint angles from 2D data alone, i.e. up to a rigid motion of the world; it parks the frame on the animal
(z along spine - paws).  The reference's kinetic-dataset model has constraints that are NOT invariant under a tilt of the world -- `spine_phi_0`
(|roll of the rear body| <= 0.05), neck / spine / tail yaw and roll differences within +-0.05 ... +-0.1 (cheetah.py:306-352) -- and its stored solutions
obey them in ITS world frame.  This script finds the tilt (two angles; a yaw about z changes none of those quantities) that minimises the violation of
this repository's `-02` bound table by the recovered angles of a trial's KINEMATIC result, re-expresses the angles (R_i -> R_w R_i, x -> R_w x) and the
cameras (R_c -> R_c R_w^T) of that pin and of the pins that share its cameras, and stores R_w beside them.  What is left of the violations, and how the
new z axis compares with the direction in which the centre of mass falls in the aerial phase, are for the tests (tests/test_free_flight_pin.py).
usage: python tools/reframe_kinetic_pin.py <animal> <kinematic fixture> [more fixtures with the same cameras ...]"""
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from cheetah_pose_estimation_amd import skeleton, synth  # noqa: E402

ANIMAL = sys.argv[1]
FIX = sys.argv[2:]
GOLD = os.path.join(ROOT, "tests", "golden")


def rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-14:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def inv_rodrigues(R):
    th = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
    if th < 1e-12:
        return np.zeros(3)
    return th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


def reframe_q(sk, q, Rw):
    """the same poses in the world frame rotated by Rw: principal ZYX triple of Rw R_i per link, yaw on the sheet next to the old one"""
    out = q.copy()
    out[:, 0:3] = q[:, 0:3] @ Rw.T
    for i in range(sk.n_links):
        R = np.einsum("ab,nbc->nac", Rw, synth.rot_zyx(q[:, 3 + 3 * i:6 + 3 * i]))
        th = np.arcsin(np.clip(-R[:, 2, 0], -1, 1)); ph = np.arctan2(R[:, 2, 1], R[:, 2, 2]); ps = np.arctan2(R[:, 1, 0], R[:, 0, 0])
        ps = ps + 2 * np.pi * np.round((q[:, 3 + 3 * i + 2] - ps) / (2 * np.pi))
        out[:, 3 + 3 * i] = ph; out[:, 4 + 3 * i] = th; out[:, 5 + 3 * i] = ps
    return out


def violations(sk, q):
    v = []
    for b in range(sk.n_bounds):
        ia, ib = sk.bound_a[b], sk.bound_b[b]
        d = q[:, ia] - (q[:, ib] if ib >= 0 else 0.0)
        d = (d + np.pi) % (2 * np.pi) - np.pi
        v.append(np.maximum(np.maximum(d - sk.bound_up[b], sk.bound_lo[b] - d), 0.0))
    return np.array(v)


def main():
    sk = skeleton.build_skeleton(f"{ANIMAL}-02", 24, kinetic_dataset=True)
    Z0 = dict(np.load(os.path.join(GOLD, FIX[0])))
    if "world_tilt" in Z0:
        print(FIX[0], "is already in a chosen world frame"); return
    q0 = Z0["q"]
    tilt = lambda p: rodrigues(np.array([p[0], p[1], 0.0]))
    f = lambda p: violations(sk, reframe_q(sk, q0, tilt(p))).ravel()
    best = None
    for start in ((0.0, 0.0), (0.1, 0.0), (-0.1, 0.0), (0.0, 0.1), (0.0, -0.1)):
        s = least_squares(f, np.array(start), xtol=1e-14, ftol=1e-14, gtol=1e-14)
        if best is None or s.cost < best.cost:
            best = s
    Rw = tilt(best.x)
    before, after = violations(sk, q0), violations(sk, reframe_q(sk, q0, Rw))
    print(f"tilt {np.degrees(best.x)} deg; worst violation of the -02 bounds {before.max():.4f} -> {after.max():.4f} rad; per bound before / after:")
    for b in range(sk.n_bounds):
        if before[b].max() > 1e-4 or after[b].max() > 1e-4:
            print(f"   bound {b} (dofs {sk.bound_a[b]}, {sk.bound_b[b]}; [{sk.bound_lo[b]:.3f}, {sk.bound_up[b]:.3f}]): {before[b].max():.4f} -> {after[b].max():.4f}")
    for fx in FIX:
        Z = dict(np.load(os.path.join(GOLD, fx)))
        Z["q"] = reframe_q(sk, Z["q"], Rw)
        cams = Z["cams"].copy()
        for c in range(cams.shape[0]):
            Rc = rodrigues(cams[c, 9:12])
            cams[c, 9:12] = inv_rodrigues(Rc @ Rw.T)
        Z["cams"] = cams
        Z["world_tilt"] = Rw
        # the poses are the same: FK in the new frame against the old positions
        p_old = synth.fk_numpy(sk, np.load(os.path.join(GOLD, fx))["q"])[0]
        p_new = synth.fk_numpy(sk, Z["q"])[0]
        print(fx, "max |R_w p_old - p_new| =", np.abs(p_old @ Rw.T - p_new).max())
        np.savez_compressed(os.path.join(GOLD, fx), **Z)


if __name__ == "__main__":
    main()
