import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cheetah_pose_estimation_amd import _lib, abi, skeleton, synth
from oracle import oracle as O
sk = skeleton.build_skeleton("phantom", 25); sk.n_bounds = 0
cams = synth.make_cameras(6)
d = synth.make_batch(sk, cams, B=1, N=24, seed=61, wide_limbs=True)
rng = np.random.default_rng(0)
q0 = d["q_true"] + rng.normal(0, 0.01, d["q_true"].shape)
for it in (10, 15, 20, 30, 45, 70, 200):
    opts = abi.default_options(); opts.max_iter = it; pass
    h = _lib.Handle(sk, cams, opts)
    out = h.solve_host(q0, d["meas"], d["weight"]); h.close()
    ref = O.solve(sk, cams, opts, None, q0[0], d["meas"][0], d["weight"][0])
    dq = np.abs(out["q"][0] - ref["q"])
    print(it, "max|dq|", dq.max(), "argmax", np.unravel_index(dq.argmax(), dq.shape), "cost gpu/oracle", out["stats"][0].cost, ref["stats"].cost, "iters", out["stats"][0].iterations, ref["stats"].iterations)
