"""bench.py's host-side contract pieces that need no GPU: the algorithmic byte table, the PMC traffic record and
the cpu_baseline object (the only place outside tests / smoke where the oracle is timed)."""
import importlib.util
import json
import os

import numpy as np

from cheetah_pose_estimation_amd import abi, skeleton, synth

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_and_traffic_record():
    b = _bench()
    # q 432 + meas 2400 + weight 1200 + r 2400 + J 276*2*6*8 + eps 432 = 33 360 -- but k_resjac<false>, the timed kernel,
    # never reads the weights: its numerator is 32 160; the cost variant reads them and writes 8 B more
    assert b.resjac_bytes_per_frame(6, 25, 276, 54, False) == 432 + 2400 + 2400 + 276 * 2 * 6 * 8 + 432 == 32160
    assert b.resjac_bytes_per_frame(6, 25, 276, 54, True) == 33360 + 8
    assert b.resjac_bytes_per_frame(6, 24, 272, 54, False) == 432 + 2304 + 2304 + 272 * 2 * 6 * 8 + 432    # 270 slots padded to 272
    fn, lm = b.solve_bytes_per_frame_iteration(6, 25)
    assert fn == 8 * (66 + 300 + 150 + 784 + 28 + 48 + 8 + 54) and lm > 2 * 4 * 784 * 8      # factor column written and read back
    assert 2.5e5 < b.lm_flops_per_frame_iteration() < 3.5e5
    assert b.HBM_PEAK == 8.0e12
    t = b.pmc_traffic(2048, 200, 6, 25)
    with open(os.path.join(ROOT, "profiles", "r03_pmc.json")) as f:      # the newest round's record wins (bounded workload: scaled per frame)
        rec = json.load(f)
    per_frame = rec["kernels"]["k_resjac<false"]["total_corrected"] / rec["frames_per_launch"]
    assert abs(t - per_frame * 2048 * 200) < 1e-6 * t and 0.9 < t / (32160 * 2048 * 200) < 1.1
    lm = b.pmc_traffic(2048, 200, 6, 25, kernel="k_lm_step")
    bk = b.pmc_traffic(2048, 200, 6, 25, kernel="k_lm_back")
    assert lm is not None and bk is not None and lm + bk > t              # factor + solve kernels together move more bytes than the residual kernel
    assert abs(b.pmc_traffic(1024, 200, 6, 25) - t / 2) < 1e-6 * t       # scaled per frame
    assert b.pmc_traffic(2048, 200, 6, 24) is None                       # no PMC passes committed for that shape


def test_cpu_baseline_object():
    b = _bench()
    sk = skeleton.build_skeleton("phantom", 25)
    cams = synth.make_cameras(6)
    d = synth.make_batch(sk, cams, B=2, N=20, seed=1234)
    out = b.cpu_baseline(sk, cams, abi.default_options(120.0), d, budget_s=0.3)
    assert out["unit"] == "frames/s" and out["kind"] == "port" and out["cores"] == 1 and out["value"] > 0
    assert "single thread" in out["sample"] and out["solves_per_s"] > 0 and out["solve_iterations"] > 0
    mt = out["multi_thread"]
    assert mt["unit"] == "frames/s" and 1 <= mt["cores"] <= len(os.sched_getaffinity(0)) and mt["value"] > 0
    assert mt["solves_per_s"] > 0 and "threads" in mt["solve_sample"]


def test_plain_invocation_spawns_its_ranks():
    """`python bench.py --gpus 2` with no launcher starts two fresh ranks itself (the parent never touches the
    GPU), they rendezvous on 127.0.0.1 (gloo here: CPE_BENCH_DRYRUN exercises the plumbing without a GPU), every sequence is
    owned by exactly one rank, and rank 0 alone prints one JSON line."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["CPE_BENCH_DRYRUN"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["dryrun"] and j["n_gpus"] == 2 and j["owned"] == list(range(8)) and j["max_over_ranks"] == 2.0
