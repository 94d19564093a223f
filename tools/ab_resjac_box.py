#!/usr/bin/env python3
"""Developer tool: k_resjac on THIS box -- the shipped kernel against the build that computes J but does not store it (-DCPE_RJ_NOSTORE), interleaved.
Tells whether a slow box is slow in its clocks (both numbers move) or in its memory (only the shipped one does)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from cheetah_pose_estimation_amd import _lib, abi, skeleton, synth
import ctypes as C
sk = skeleton.build_skeleton("phantom", 25); cams = synth.make_cameras(6)
d = synth.make_batch(sk, cams, B=16, N=200, seed=1)
dev = torch.device("cuda", 0)
T = {k: torch.tensor(d[k], device=dev).repeat((128,) + (1,) * (d[k].ndim - 1)).contiguous() for k in ("q_true", "meas", "weight")}
B, N = T["q_true"].shape[:2]
res = {}
here = os.path.dirname(os.path.abspath(_lib.__file__))
for name in ("libcpe.so", "libcpe_nostore.so", "libcpe.so", "libcpe_nostore.so"):
    _lib.LIB_PATH = os.path.join(here, name); _lib._LIB = None
    h = _lib.Handle(sk, cams, abi.default_options())
    S = h.jacobian_slots() if hasattr(h, "jacobian_slots") else 276
    r = torch.empty((B, N, 6, 25, 2), dtype=torch.float64, device=dev); J = torch.empty((B, N, 6, S, 2), dtype=torch.float64, device=dev); eps = torch.empty((B, N, 54), dtype=torch.float64, device=dev)
    for _ in range(int(os.environ.get("AB_WARM", "3"))): h.eval_resjac(T["q_true"], T["meas"], T["weight"], r, J, eps)
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(int(os.environ.get("AB_STEPS", "20"))): h.eval_resjac(T["q_true"], T["meas"], T["weight"], r, J, eps)
    h.synchronize(); ms = (time.perf_counter() - t0) / int(os.environ.get("AB_STEPS", "20")) * 1e3
    res.setdefault(name, []).append(ms); h.close()
print({k: [round(v, 3) for v in vs] for k, vs in res.items()})
