#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload = "cfg2"): synthetic 200-frame x 6-camera x 25-marker sequences,
phantom skeleton, fp64.  One "step" = one pass of the residual+Jacobian hot path (k_resjac) over a batch of
B sequences resident in HBM.  `value` = frames/s of that pass over all ranks; full-trajectory solves/s
(cpe_solve, LM + block-banded Cholesky) are timed next to it and reported in "solves".
Sequences are independent: ranks shard them, there is no collective on the data path (weak scaling).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

BYTES_PER_FRAME = {25: 33360, 24: 32544}      # SURVEY 8d algorithmic bytes, C=6
HBM_PEAK = 8.0e12


def tile_batch(torch, d, B, dev, seed):
    """Upload P unique sequences and tile them to B on the device (q gets a small seeded perturbation so
    no two sequences are bit-identical)."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    P = d["q_true"].shape[0]
    reps = (B + P - 1) // P
    out = {}
    for k in ("q_true", "q_init", "meas", "weight"):
        t = torch.tensor(d[k], device=dev)
        out[k] = t.repeat((reps,) + (1,) * (t.dim() - 1))[:B].contiguous()
    out["q_true"] += 1e-3 * torch.randn(out["q_true"].shape, generator=g, device=dev, dtype=torch.float64)
    out["q_init"][..., :3] += 1e-3 * torch.randn(out["q_init"][..., :3].shape, generator=g, device=dev, dtype=torch.float64)
    return out


def cpu_baseline(sk, cams, opts, d, budget_s=8.0):
    """CPU oracle (oracle/, plain C) timed on a bounded sample of the same workload: one thread (the scalar port) and
    OpenMP over sequences on this box's CPU share; the same C loop for both, output buffers reused across sequences."""
    from oracle import oracle as O
    O.lib()
    Bq, N = d["q_true"].shape[0], d["q_true"].shape[1]
    args = (sk, cams, opts, d["q_true"], d["meas"], d["weight"])

    def timed(threads, seconds, rate_guess):
        reps = max(1, int(round(seconds * rate_guess / (Bq * N))))
        t0 = time.perf_counter()
        used, _ = O.eval_resjac_batch(*args, reps=reps, threads=threads)
        dt = time.perf_counter() - t0
        return used, reps * Bq, dt, reps * Bq * N / dt

    _, _, _, probe = timed(1, 0.0, 1.0)                     # one pass: calibrates the sample size
    u1, n1, dt1, rate1 = timed(1, budget_s, probe)
    share = min(len(os.sched_getaffinity(0)), 16)           # a one-GPU box gives a job 16 host cores
    um, nm, dtm, ratem = timed(share, 0.5 * budget_s, rate1 * share)
    t1 = time.perf_counter()
    res = O.solve(sk, cams, opts, None, d["q_init"][0], d["meas"][0], d["weight"][0])
    ts = time.perf_counter() - t1
    return dict(value=rate1, unit="frames/s", cores=1, kind="port",
                sample=f"{n1} sequences x {N} frames of the same synthetic workload, oracle/cpe_oracle.c single thread, {dt1:.1f} s",
                solves_per_s=1.0 / ts, solve_iterations=int(res["stats"].iterations), host_cores_available=os.cpu_count(),
                multi_thread=dict(value=ratem, unit="frames/s", cores=um, sample=f"{nm} sequences, OpenMP over sequences, {dtm:.1f} s"))


def pmc_traffic(B, N, C, L):
    """HBM bytes per k_resjac launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc_resjac.json:
    separate FETCH_SIZE / WRITE_SIZE runs, read side doubled per MI355X_MICROARCH.md), scaled per frame.
    None if no profile for this marker/camera count is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_resjac.json"))):
        try:
            with open(f) as fh:
                j = json.load(fh)
            c = j["config"]
            if c["C"] == C and c["L"] == L:
                best = j["hbm_bytes_per_launch"]["total_corrected"] / (c["B"] * c["N"]) * (B * N)
        except Exception:
            pass
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10, help="untimed launches; the clocks of an idle MI355X need ~10 launches (30 ms) to ramp")
    ap.add_argument("--batch", type=int, default=2048, help="sequences per GPU for the residual+Jacobian pass")
    ap.add_argument("--solve-batch", type=int, default=2048, help="sequences per GPU for the solve timing")
    ap.add_argument("--markers", type=int, default=25)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-solve", action="store_true")
    ap.add_argument("--no-l24", action="store_true", help="skip the extra 24-marker residual+Jacobian measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cheetah_pose_estimation_amd import _lib, abi, skeleton, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # CPE_BENCH_REHEARSAL=1: rehearse the multi-rank path on a ONE-GPU box -- every rank uses cuda:0 and the two collectives
    # (timing barrier, MAX over ranks) go over gloo.  The numbers of such a run mean nothing; it checks the plumbing.
    rehearsal = os.environ.get("CPE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL; used for the timing barrier and the MAX over ranks only

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    L, N, C = args.markers, args.frames, 6
    sk = skeleton.build_skeleton("phantom", L)
    cams = synth.make_cameras(C)
    opts = abi.default_options(120.0)
    h = _lib.Handle(sk, cams, opts, device=local)
    S = h.S
    P = 32
    d = synth.make_batch(sk, cams, B=P, N=N, seed=1234 + 1000 * rank)     # sequence b uses seed 1234 + b (+ rank offset)
    B = args.batch
    t = tile_batch(torch, d, B, dev, seed=rank)
    r = torch.empty((B, N, C, L, 2), dtype=torch.float64, device=dev)
    J = torch.empty((B, N, C, S, 2), dtype=torch.float64, device=dev)
    eps = torch.empty((B, N, sk.nq), dtype=torch.float64, device=dev)
    stream = torch.cuda.ExternalStream(h.stream, device=dev)               # events on the stream the kernels run on

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        h.eval_resjac(t["q_true"], t["meas"], t["weight"], r, J, eps, None)
    h.synchronize()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        h.eval_resjac(t["q_true"], t["meas"], t["weight"], r, J, eps, None)
        ev[k][1].record(stream)
    h.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    elapsed = max_over_ranks(elapsed)
    frames_total = world * B * N * args.steps
    value = frames_total / elapsed

    # "also report L=24" -- the reference's own 24 markers, same cameras and sequence shape, same kernel
    l24 = None
    if L == 25 and world == 1 and not args.no_l24:
        del r, J, eps
        sk24 = skeleton.build_skeleton("phantom", 24)
        h24 = _lib.Handle(sk24, cams, opts, device=local)
        d24 = synth.make_batch(sk24, cams, B=P, N=N, seed=1234)
        t24 = tile_batch(torch, d24, B, dev, seed=7)
        r = torch.empty((B, N, C, 24, 2), dtype=torch.float64, device=dev)
        J = torch.empty((B, N, C, h24.S, 2), dtype=torch.float64, device=dev)
        eps = torch.empty((B, N, sk24.nq), dtype=torch.float64, device=dev)
        s24 = torch.cuda.ExternalStream(h24.stream, device=dev)
        for _ in range(max(2, args.warmup)):
            h24.eval_resjac(t24["q_true"], t24["meas"], t24["weight"], r, J, eps, None)
        h24.synchronize()
        e24 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in e24:
            a.record(s24)
            h24.eval_resjac(t24["q_true"], t24["meas"], t24["weight"], r, J, eps, None)
            b.record(s24)
        h24.synchronize()
        ms24 = float(np.mean([a.elapsed_time(b) for a, b in e24]))
        l24 = dict(value=B * N / (ms24 * 1e-3), unit="frames/s", kernel_ms=ms24, bytes_per_frame=BYTES_PER_FRAME[24],
                   frac=BYTES_PER_FRAME[24] * B * N / (ms24 * 1e-3) / HBM_PEAK)
        h24.close()
        del t24

    solves = None
    if not args.no_solve:
        Bs = args.solve_batch
        ts_ = tile_batch(torch, d, Bs, dev, seed=100 + rank)
        r = J = eps = None
        q = torch.empty((Bs, N, sk.nq), dtype=torch.float64, device=dev); dq = torch.empty_like(q); ddq = torch.empty_like(q)
        pos = torch.empty((Bs, N, L, 3), dtype=torch.float64, device=dev); me = torch.empty((Bs, N, C, L, 2), dtype=torch.float64, device=dev)
        h.solve(ts_["q_init"], ts_["meas"], ts_["weight"], q, dq, ddq, pos, me)   # warm-up at full size: the solver workspace (16 GB for 2048 sequences) is allocated here
        barrier()
        t1 = time.perf_counter()
        st, stats = h.solve(ts_["q_init"], ts_["meas"], ts_["weight"], q, dq, ddq, pos, me)
        barrier()
        el = time.perf_counter() - t1
        el = max_over_ranks(el)
        its = np.array([s.iterations for s in stats])
        stt = np.array([s.status for s in stats])
        solves = dict(value=world * Bs / el, unit="solves/s", batch_per_gpu=Bs, seconds=el, iterations_mean=float(its.mean()),
                      iterations_max=int(its.max()), converged_frac=float((stt == 0).mean()))

    if rank == 0:
        bpf = BYTES_PER_FRAME.get(L, 33360)
        ach = bpf * B * N / (kern_ms * 1e-3)
        out = {
            "metric": "frames/sec residual+Jacobian eval + full-traj solves/sec, 200-frame 6-cam seq",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg2: synthetic 200-frame x 6-cam x 25-marker sequences, phantom skeleton, const-accel model",
                       "frames": N, "cams": C, "markers": L, "sequences_per_gpu": B, "parallelism": f"shard{world} (independent sequences, no collective)"},
            "solves": solves, "markers24": l24,
            "roofline": {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                         "traffic": pmc_traffic(B, N, C, L), "kernel": "k_resjac<false>", "kernel_ms": kern_ms, "bytes_per_frame": bpf},
        }
        if not args.no_cpu and world == 1:          # the CPU leg is timed on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(sk, cams, opts, d)
        print(json.dumps(out))
    h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
