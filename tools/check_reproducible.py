"""Developer tool (not part of the product or the tests): solves the same inputs repeatedly, alone and inside a batch, and reports
whether the trajectories agree bit for bit (profiles/r01_solver_notes.md)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cheetah_pose_estimation_amd import _lib, abi, skeleton, synth
sk = skeleton.build_skeleton("phantom", 25); cams = synth.make_cameras(6); opts = abi.default_options()
h = _lib.Handle(sk, cams, opts, device=0)
d = synth.make_batch(sk, cams, B=24, N=200, seed=1234)
outs = [h.solve_host(d["q_init"], d["meas"], d["weight"]) for _ in range(3)]
print("bitwise equal q over 3 runs:", all(np.array_equal(outs[0]["q"], o["q"]) for o in outs[1:]))
print("iterations", [s.iterations for s in outs[0]["stats"]][:8], [s.iterations for s in outs[1]["stats"]][:8])
# batching independence: sequence 5 alone vs inside the batch
o1 = h.solve_host(d["q_init"][5:6], d["meas"][5:6], d["weight"][5:6])
print("alone == in batch:", np.array_equal(o1["q"][0], outs[0]["q"][5]), np.abs(o1["q"][0] - outs[0]["q"][5]).max())
d2 = synth.make_batch(sk, cams, B=1, N=450, seed=123)
its = []
for k in range(4):
    o = h.solve_host(d2["q_init"], d2["meas"], d2["weight"]); its.append((o["stats"][0].status, o["stats"][0].iterations, o["stats"][0].cost))
print("long ill-posed sequence, 4 runs:", its)
