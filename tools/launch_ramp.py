"""Developer tool (not part of the product or the tests): 80 back-to-back cpe_eval_resjac launches from an idle GPU, HIP-event time
per launch -- shows the clock ramp recorded in profiles/r01_resjac_ablation.md."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from cheetah_pose_estimation_amd import _lib, abi, skeleton, synth
sk = skeleton.build_skeleton("phantom", 25); cams = synth.make_cameras(6); opts = abi.default_options()
h = _lib.Handle(sk, cams, opts, device=0)
dev = torch.device("cuda", 0)
d = synth.make_batch(sk, cams, B=32, N=200, seed=1234)
B = 2048
rep = lambda a: torch.tensor(np.ascontiguousarray(np.tile(a, (B // 32,) + (1,) * (a.ndim - 1))), device=dev)
q, me, we = rep(d["q_true"]), rep(d["meas"]), rep(d["weight"])
r = torch.empty((B, 200, 6, 25, 2), dtype=torch.float64, device=dev); J = torch.empty((B, 200, 6, h.S, 2), dtype=torch.float64, device=dev); eps = torch.empty((B, 200, sk.nq), dtype=torch.float64, device=dev)
stream = torch.cuda.ExternalStream(h.stream, device=dev)
torch.cuda.synchronize(); time.sleep(2.0)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(80)]
for a, b in ev:
    a.record(stream); h.eval_resjac(q, me, we, r, J, eps, None); b.record(stream)
h.synchronize()
t = [a.elapsed_time(b) for a, b in ev]
print("launch ms:", " ".join(f"{x:.2f}" for x in t))
print("mean of last 40: %.3f  min %.3f" % (np.mean(t[40:]), min(t)))
