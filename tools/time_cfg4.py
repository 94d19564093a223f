#!/usr/bin/env python3
"""Developer tool (not part of the product or the tests): per-kernel milliseconds per launch of the physics-based solve (config 4: 200 frames, six
cameras, phantom, gallop) and, with a -DCPE_LM_STAMPS build given by --lib, the shader-clock shares of k_dyn_eval's phases (node 5 of sequence 0)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--N", type=int, default=200)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
from cheetah_pose_estimation_amd import _lib, abi, skeleton, synth
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
sk = skeleton.without_motion_model(skeleton.build_skeleton("phantom", 24))
cams = synth.make_cameras(6)
d = synth.make_gallop_batch(sk, cams, B=8, N=args.N, seed=4321)
rep = (args.B + 7) // 8
D = {k: np.ascontiguousarray(np.concatenate([d[k]] * rep)[:args.B]) for k in ("q_init", "meas", "weight", "stance")}
hk = _lib.Handle(skeleton.build_skeleton("phantom", 24), cams, abi.default_options(120.0))
kin = hk.solve_host(D["q_init"], D["meas"], D["weight"]); hk.close()
opts = abi.default_options(120.0); opts.tol_cost, opts.tol_step, opts.max_iter, opts.max_outer = 0.0, 0.0, args.iters, 0
ko = abi.default_kinetic_options(skeleton.dyn_options("phantom"), 120.0)
h = _lib.Handle(sk, cams, opts)
for rnd in range(2):
    if rnd == 1:
        h.profile(True)
    t0 = time.perf_counter()
    r = h.solve_kinetic_host(ko, kin["q"], D["meas"], D["weight"], D["stance"])
    el = time.perf_counter() - t0
its = np.mean([s.iterations for s in r["stats"]])
print(f"B={args.B} N={args.N} iterations {its:.1f}: {el * 1e3:.1f} ms (host copies included)")
print("   per-kernel ms/launch:", ", ".join(f"{k} {ms / max(n, 1):.3f} x{n}" for k, (ms, n) in h.profile_totals().items() if n))
if hasattr(h.lib, "cpe_debug_fn_stamps"):
    import ctypes as C
    z = (C.c_ulonglong * 16)()
    h.lib.cpe_debug_fn_stamps(z)
    names = {9: "load + base positions + base evaluations", 10: "A", 14: "G, b, M", 15: "partial Cholesky", 11: "Newton on F, x", 12: "evaluation at f*, active sets", 13: "H_ff, cost, rows"}
    tot = sum(z[i] for i in names) or 1
    jn = {0: "states, sin / cos, rotations, nearest", 1: "dB per link and local variable", 2: "A_i, D_i", 3: "block matrices K, C, M (+ d(Af)/dq)", 4: "coordinate map", 5: "row tables to registers", 6: "84 columns"}
    tj = sum(z[i] for i in jn) or 1
    if os.environ.get("JAC_STAMPS"):
        print("k_dyn_jac phases (node 5): " + ", ".join(f"{n} {100.0 * z[i] / tj:.1f}%" for i, n in jn.items()) + f"  [{tj} ticks in all]")
    print("k_dyn_eval phases (node 5): " + ", ".join(f"{n} {100.0 * z[i] / tot:.1f}%" for i, n in names.items()) + f"  [{tot / (2.0 * (args.iters + 1)):.0f} cycles per evaluation]")
