#!/usr/bin/env python3
"""Developer tool: time a fixed number of LM iterations of cpe_solve (per-kernel ablation builds can be
selected with --lib).  Not part of the product or the tests."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--N", type=int, default=200)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
import torch
from cheetah_pose_estimation_amd import _lib, abi, skeleton, synth
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
sk = skeleton.build_skeleton("phantom", 25); cams = synth.make_cameras(6)
opts = abi.default_options(); opts.max_iter = args.iters; opts.tol_step = 0.0; opts.tol_cost = 0.0; opts.max_outer = 0
h = _lib.Handle(sk, cams, opts)
d = synth.make_batch(sk, cams, B=16, N=args.N, seed=1)
dev = torch.device("cuda", 0)
rep = (args.B + 15) // 16
T = {k: torch.tensor(d[k], device=dev).repeat((rep,) + (1,) * (d[k].ndim - 1))[:args.B].contiguous() for k in ("q_init", "meas", "weight")}
q = torch.empty_like(T["q_init"]); dq = torch.empty_like(q); ddq = torch.empty_like(q)
pos = torch.empty((args.B, args.N, 25, 3), dtype=torch.float64, device=dev); me = torch.empty((args.B, args.N, 6, 25, 2), dtype=torch.float64, device=dev)
for rnd in range(3):
    if rnd == 2:
        h.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st, stats = h.solve(T["q_init"], T["meas"], T["weight"], q, dq, ddq, pos, me)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
its = np.mean([s.iterations for s in stats])
print("   per-kernel ms/launch:", ", ".join(f"{k} {ms / n:.3f} x{n}" for k, (ms, n) in h.profile_totals().items()))
if hasattr(h.lib, "cpe_debug_lm_stamps"):
    import ctypes as C
    z = (C.c_ulonglong * 32)()
    h.lib.cpe_debug_lm_stamps(z)
    # three timelines of workgroup 0: the factor wave (thread 0), an update wave (thread 64), and k_lm_back's two waves
    groups = {"k_lm_step factor wave": {0: "accept/reduce", 1: "stage rows", 2: "sweep", 3: "wait for P1", 6: "A11 tiles + rhs"},
              "k_lm_step update wave": {10: "wait for sweep", 4: "P1 panel product", 9: "sync", 5: "P2 trailing update", 7: "sync",
                                        12: "wait staged", 14: "R singles", 15: "R heavy", 11: "R gradient", 13: "sync"},
              "k_lm_back arithmetic wave": {22: "partial sums", 23: "substitution", 28: "stores", 29: "barrier"},
              "k_lm_back loader wave": {25: "issue", 30: "vmcnt wait", 31: "barrier"}}
    for gname, items in groups.items():
        tot = sum(z[i] for i in items) or 1
        print(f"{gname}: " + ", ".join(f"{n} {100.0 * z[i] / tot:.1f}%" for i, n in items.items()) + f"  [{tot / max(its + 1, 1):.3g} cycles/launch]")
    print(f"factorisation failures (all workgroups): {z[8]}")
if hasattr(h.lib, "cpe_debug_fn_stamps"):
    import ctypes as C
    z = (C.c_ulonglong * 16)()
    h.lib.cpe_debug_fn_stamps(z)
    tot = sum(z) or 1
    names = ["load+sincos+trunk R+leg euler+dyn", "hooke S rows", "positions", "slot dp + Dp", "pairs", "H accumulate", "bounds", "gmm", "write out"]
    print("k_frame_normal phase shares (block 0):", ", ".join(f"{n} {100.0 * z[i] / tot:.1f}%" for i, n in enumerate(names)))
print(f"{os.path.basename(args.lib or 'libcpe.so'):28s} B={args.B} iters={its:.1f} total {el*1e3:8.2f} ms  per-iteration {el*1e3/max(its,1):7.3f} ms  cost0 {stats[0].cost:.6g}")
