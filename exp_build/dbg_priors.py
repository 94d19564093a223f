import numpy as np, sys
sys.path.insert(0,'/root/repo')
from cheetah_pose_estimation_amd import skeleton, synth, abi, priors, _lib
from oracle import oracle
sk = skeleton.build_skeleton("phantom", 24)
cams6 = synth.make_cameras(6); cam1=(abi.Camera*1)(cams6[2]); opts=abi.default_options()
for which in ("pose","motion","both"):
    pr = priors.load_priors(pose=which in ("pose","both"), motion=which in ("motion","both"))
    h=_lib.Handle(sk, cam1, opts, pr)
    for (N,seed,inoise) in ((30,91,0.03),(40,5,0.03),(60,7,0.02)):
        d = synth.make_batch(sk, cam1, B=2, N=N, seed=seed, init_noise=inoise)
        out = h.solve_host(d["q_init"], d["meas"], d["weight"])
        for b in range(2):
            ref = oracle.solve(sk, cam1, opts, pr, d["q_init"][b], d["meas"][b], d["weight"][b])
            st, rs = out["stats"][b], ref["stats"]
            rmse = np.sqrt(((out["positions"][b] - ref["positions"]) ** 2).sum(-1).mean())
            f, g, _, terms, _ = oracle.objective(sk, cam1, opts, pr, out["q"][b], d["meas"][b], d["weight"][b], want_grad=True)
            f2, g2, _, _, _ = oracle.objective(sk, cam1, opts, pr, ref["q"], d["meas"][b], d["weight"][b], want_grad=True)
            print(which, N, b, "gpu st %d it %d cost %.8f | ora st %d it %d cost %.8f | rmse %.2e | gmax gpu %.2e ora %.2e | pose %.3f/%.3f motion %.3f/%.3f" % (
                st.status, st.iterations, st.cost, rs.status, rs.iterations, rs.cost, rmse, np.abs(g).max(), np.abs(g2).max(), st.cost_pose, rs.cost_pose, st.cost_motion, rs.cost_motion), flush=True)
    h.close()
