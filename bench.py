#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: started by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK /
WORLD_SIZE in the environment), or started plainly as `python bench.py --gpus N`, in which case THIS process starts the N ranks
as child processes before anything here has touched the GPU, waits for them and exits with their status.

Workload (config.workload = "cfg2"): synthetic 200-frame x 6-camera x 25-marker sequences, phantom skeleton, fp64.
One "step" = one pass of the residual+Jacobian hot path (k_resjac) over a batch of B sequences resident in HBM.  `value` =
frames/s of that pass over all ranks; full-trajectory solves/s (cpe_solve: LM + block-banded Cholesky) are timed next to it and
reported in "solves" with their own roofline object.  Sequences are independent: the global list of sequences is dealt to the
ranks round-robin (cheetah_pose_estimation_amd.sharding), there is no collective on the data path (weak scaling).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12
FP64_PEAK = 78.6e12     # MI355X vector fp64 = matrix fp64 (no MFMA advantage for fp64 on gfx950)


def resjac_bytes_per_frame(C, L, S, nq, with_cost):
    """ALGORITHMIC bytes of one frame of k_resjac: read q, meas (and weight only when the robust cost is asked for --
    k_resjac<false> never loads it); write residual, the S structurally non-zero Jacobian slots x 2 rows x C cameras, the
    acceleration slack (and the cost).  C=6, L=25, S=276: 32 160 B without cost, 33 368 B with."""
    b = 8 * nq + 16 * C * L + 16 * C * L + 16 * C * S + 8 * nq
    return b + (8 * C * L + 8 if with_cost else 0)


def solve_bytes_per_frame_iteration(C, L, nq=54, nu=28, nrev=12, pb=3):
    """ALGORITHMIC bytes one LM iteration moves per frame: k_frame_normal reads the state (nq + nrev), meas, weight and
    writes B (nu^2), g, Gamma (4 nrev), the cost record (8) and the consistent Euler angles; k_lm_step reads B, g, Gamma, the state
    and the cost record, writes the factor column ((pb+1) nu^2), z, the total gradient and the undamped diagonal; k_lm_back reads the
    factor column and z back, writes delta, reads gradient + diagonal for the predicted reduction, and reads + writes the state for
    the trial iterate.  (lm = the two kernels of the step together.)"""
    ns = nq + nrev
    fn = 8 * (ns + 2 * C * L + C * L + nu * nu + nu + 4 * nrev + 8 + nq)
    lm = 8 * (nu * nu + nu + 4 * nrev + ns + 8 + (pb + 1) * nu * nu + 3 * nu + (pb + 1) * nu * nu + nu + nu + 2 * nu + 2 * ns)
    return fn, lm


def lm_flops_per_frame_iteration(nu=28, pb=3):
    """fp64 flops of k_lm_step per frame: Cholesky of the diagonal block, panel solve of pb blocks, trailing update of the window
    (lower triangle of the pb diagonal blocks, all of the pb (pb-1)/2 others), right-hand side, backward substitution"""
    chol = nu ** 3 / 3.0
    panel = pb * nu * nu * nu
    trail = pb * nu * (nu + 1) * nu + (pb * (pb - 1) // 2) * 2 * nu ** 3
    rhs = 2 * pb * nu * nu + 2 * (pb + 1) * nu * nu
    return chol + panel + trail + rhs


def _gen_sequence(job):
    """one synthetic sequence (numpy only; runs in a worker process forked before anything touches the GPU)"""
    kind, L, cam_sel, N, seed = job
    from cheetah_pose_estimation_amd import abi, skeleton, synth
    cams = synth.make_cameras(6)
    if cam_sel is not None:
        cams = (abi.Camera * len(cam_sel))(*[cams[c] for c in cam_sel])
    if kind == "gallop":
        sk = skeleton.without_motion_model(skeleton.build_skeleton("phantom", L))
        return synth.make_gallop_batch(sk, cams, B=1, N=N, seed=seed)
    return synth.make_batch(skeleton.build_skeleton("phantom", L), cams, B=1, N=N, seed=seed)


def make_sequences(kind, L, cam_sel, N, seeds, workers):
    """sequence with global index b is generated from its own seed -- no tiling.  Host generation costs 9 ms (18 ms for the
    planted gallop) per sequence: spread over `workers` forked processes (numpy only) when there are many."""
    jobs = [(kind, L, cam_sel, N, int(sd)) for sd in seeds]
    parts = None
    if workers > 1 and len(jobs) >= 64:
        try:
            import multiprocessing as mp
            with mp.get_context("fork").Pool(workers) as pool:
                parts = pool.map(_gen_sequence, jobs, chunksize=max(1, len(jobs) // (4 * workers)))
        except Exception:
            parts = None
    if parts is None:
        parts = [_gen_sequence(j) for j in jobs]
    return {k: np.ascontiguousarray(np.concatenate([p_[k] for p_ in parts])) for k in parts[0]}


def upload(torch, d, dev, keys=("q_true", "q_init", "meas", "weight")):
    out = {k: torch.tensor(d[k], device=dev) for k in keys if k in d}
    torch.cuda.synchronize(dev)            # the handle launches on its own stream: nothing of the above may still be in flight
    return out


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
    torch.cuda.synchronize(dev)            # the handle launches on its own stream: nothing of the above may still be in flight
    return out


def usable_cores():
    """(cores this process can really run on, how that was found): the scheduler affinity, capped by the cgroup CPU quota when the
    container has one (a one-GPU box shows 256 host threads in its affinity mask but gives the job a 16-core quota; 256 OpenMP
    threads on 16 cores measured 2.4x SLOWER than 16)."""
    n = len(os.sched_getaffinity(0))
    how = f"sched_getaffinity={n}"
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:            # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = fh.read().split()[:2]
        if quota != "max":
            q = max(1, int(round(int(quota) / int(period))))
            how += f", cgroup cpu.max={q}"
            n = min(n, q)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                quota = int(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                period = int(fh.read())
            if quota > 0:
                q = max(1, int(round(quota / period)))
                how += f", cgroup cfs_quota={q}"
                n = min(n, q)
        except Exception:
            pass
    return n, how


def cpu_baseline(sk, cams, opts, d, budget_s=8.0):
    """CPU oracle (oracle/, plain C) timed on a bounded sample of the same workload: one thread (the scalar port) and OpenMP over
    sequences on every host core this process may run on (count stated); the same C code for both."""
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
    cores, how = usable_cores()                             # every core this process may use (affinity, cgroup quota if readable)
    # a quota the files above do not show (the one-GPU boxes: 256 threads in the mask, 16 cores' worth of time) would make `cores`
    # threads thrash: probe a few thread counts for ~0.3 s each and keep the fastest
    best = (0.0, 1)
    for tc in sorted({min(cores, c) for c in (8, 16, 32, 64, 128, 256)} | {cores}):
        _, _, _, r_tc = timed(tc, 0.3, rate1 * min(tc, 16))
        if r_tc > best[0]:
            best = (r_tc, tc)
    how += f", fastest of the probed thread counts: {best[1]}"
    cores = best[1]
    um, nm, dtm, ratem = timed(cores, 0.5 * budget_s, best[0])
    t1 = time.perf_counter()
    res = O.solve(sk, cams, opts, None, d["q_init"][0], d["meas"][0], d["weight"][0])
    ts = time.perf_counter() - t1
    # multi-thread solves: one sequence per thread at a time, as many rounds as fit in half the budget (at least one)
    nb = cores * max(1, int(0.5 * budget_s / max(ts, 1e-3)))
    rep = (nb + Bq - 1) // Bq
    qi = np.tile(d["q_init"], (rep, 1, 1))[:nb]; me = np.tile(d["meas"], (rep, 1, 1, 1, 1))[:nb]; we = np.tile(d["weight"], (rep, 1, 1, 1))[:nb]
    t2 = time.perf_counter()
    us, _, its = O.solve_batch(sk, cams, opts, qi, me, we, threads=cores)
    tm = time.perf_counter() - t2
    return dict(value=rate1, unit="frames/s", cores=1, kind="port",
                sample=f"{n1} sequences x {N} frames of the same synthetic workload, oracle/cpe_oracle.c single thread, {dt1:.1f} s",
                solves_per_s=1.0 / ts, solve_iterations=int(res["stats"].iterations), host_cores_available=os.cpu_count(), cores_usable=how,
                multi_thread=dict(value=ratem, unit="frames/s", cores=um, sample=f"{nm} sequences, OpenMP over sequences, {dtm:.1f} s",
                                  solves_per_s=nb / tm, solve_sample=f"{nb} sequences on {us} threads, {tm:.1f} s, {float(its.mean()):.1f} iterations on average"))


def progress(msg):
    """one line per finished leg on stderr: a profiler run of this script is silent on stdout until the JSON line, and a watchdog takes minutes of
    silence for a hang"""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def roof_bytes(nbytes, ms):
    """algorithmic bytes over kernel milliseconds against the HBM peak"""
    if not ms:
        return None
    return {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / HBM_PEAK}


def roof_flops(nflops, ms):
    if not ms:
        return None
    return {"bound": "fp64", "achieved": nflops / (ms * 1e-3) / 1e12, "peak": FP64_PEAK / 1e12, "unit": "TFLOP/s", "frac": nflops / (ms * 1e-3) / FP64_PEAK}


def bench_cfg3(torch, _lib, abi, skeleton, synth, dev, local, d3, N, cpu=True):
    """config 3: monocular -- camera 3 of the rig --, Gaussian-mixture pose prior + window-4 autoregressive motion prior, 24 markers,
    B unique sequences (seed 1234 + b).  Solves/s of cpe_solve, per-kernel milliseconds from the handle's HIP-event profile of the untimed
    warm-up solve, bytes / fp64 rooflines per kernel, CPU oracle beside it.  Returns (dict, solved q for the config-4 leg)."""
    from cheetah_pose_estimation_amd import priors
    sk = skeleton.build_skeleton("phantom", 24)
    cams6 = synth.make_cameras(6)
    cam1 = (abi.Camera * 1)(cams6[2])
    opts = abi.default_options(120.0)
    pr = priors.load_priors()
    h = _lib.Handle(sk, cam1, opts, pr, device=local)
    B = d3["q_init"].shape[0]
    T = upload(torch, d3, dev, keys=("q_init", "meas", "weight"))
    q = torch.empty_like(T["q_init"]); dq = torch.empty_like(q); ddq = torch.empty_like(q)
    pos = torch.empty((B, N, 24, 3), dtype=torch.float64, device=dev); me = torch.empty((B, N, 1, 24, 2), dtype=torch.float64, device=dev)
    h.profile(True)
    _, wstats = h.solve(T["q_init"], T["meas"], T["weight"], q, dq, ddq, pos, me)
    prof = h.profile_totals()
    h.profile(False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    _, stats = h.solve(T["q_init"], T["meas"], T["weight"], q, dq, ddq, pos, me)
    h.synchronize()
    el = time.perf_counter() - t0
    its = np.array([s_.iterations for s_ in stats]); stt = np.array([s_.status for s_ in stats])
    wfi = float((np.array([s_.iterations for s_ in wstats]) + 1).sum()) * N        # frame-iterations of the profiled run
    nu, nq, nrev, pb = 28, sk.nq, 12, 4
    fn_b, lm_b = solve_bytes_per_frame_iteration(1, 24, nq, nu, nrev, pb)
    lm_b += 8 * pb * nu * nu                                  # k_lm_step<4> also reads the prior's off-diagonal blocks
    lr_b = 8 * ((nq + nrev) + pb * nu * nu + 2 * nu * nu + 2 * nu + 1)   # k_lr_band: state in; 4 blocks out; diagonal block and gradient read-modify-write; cost
    ms = {k: v[0] for k, v in prof.items()}; nl = {k: v[1] for k, v in prof.items()}
    lm_ms = ms.get("k_lm_step", 0.0) + ms.get("k_lm_back", 0.0)
    kern = {"k_lm_step<4> + k_lm_back<4>": dict(ms_total=lm_ms, launches=nl.get("k_lm_step", 0), bytes_per_frame_iteration=lm_b,
                                                 roofline=roof_bytes(lm_b * wfi, lm_ms), fp64=roof_flops(lm_flops_per_frame_iteration(nu, pb) * wfi, lm_ms)),
            "k_lr_band": dict(ms_total=ms.get("k_lr_band", 0.0), launches=nl.get("k_lr_band", 0), bytes_per_frame_iteration=lr_b, roofline=roof_bytes(lr_b * wfi, ms.get("k_lr_band", 0.0))),
            "k_frame_normal": dict(ms_total=ms.get("k_frame_normal", 0.0), launches=nl.get("k_frame_normal", 0), bytes_per_frame_iteration=fn_b,
                                   roofline=roof_bytes(fn_b * wfi, ms.get("k_frame_normal", 0.0)))}
    out = dict(value=B / el, unit="solves/s", workload="cfg3: 200 frames x 1 camera x 24 markers, GMM pose prior + window-4 motion prior, seed 1234 + b",
               batch=B, seconds=el, iterations_mean=float(its.mean()), iterations_max=int(its.max()), converged_frac=float((stt == 0).mean()),
               kernels=kern, roofline=kern["k_lm_step<4> + k_lm_back<4>"]["roofline"])
    if cpu:
        from oracle import oracle as O
        O.lib()
        t1 = time.perf_counter()
        r1 = O.solve(sk, cam1, opts, pr, d3["q_init"][0], d3["meas"][0], d3["weight"][0])
        ts = time.perf_counter() - t1
        cores = min(16, usable_cores()[0])
        nb = min(B, cores)
        t2 = time.perf_counter()
        us, _, itc = O.solve_batch(sk, cam1, opts, d3["q_init"][:nb], d3["meas"][:nb], d3["weight"][:nb], threads=cores, priors=pr)
        tm = time.perf_counter() - t2
        out["cpu_baseline"] = dict(value=1.0 / ts, unit="solves/s", cores=1, kind="port", sample=f"sequence 0 of the same batch, oracle/cpe_oracle.c, {ts:.1f} s, {int(r1['stats'].iterations)} iterations",
                                   multi_thread=dict(value=nb / tm, unit="solves/s", cores=us, sample=f"{nb} sequences, OpenMP over sequences, {tm:.1f} s, {float(itc.mean()):.1f} iterations on average"))
    qh = q.cpu().numpy()
    h.close()
    return out, qh


def kinetic_flops_per_node(nq=54, nrow=138, nlat=60, nc3=84, nx=48):
    """fp64 flops of the dense algebra of one node of the physics-based model per LM iteration (the derivative evaluations are counted by the
    kernel's own note, not here): Gram matrix A^T A, its Cholesky, J^T W J (lower triangle), A^T J_e, the Schur complement"""
    gram = nq * nlat * nlat
    chol = nx ** 3 / 3.0 + nlat ** 3 / 3.0
    jtj = nrow * nc3 * nc3
    afj = 2 * nq * nlat * nc3
    schur = nlat * nlat * nc3 + nlat * nc3 * nc3
    # closed-form Jacobian: three block matrices over the 89 related link pairs (~60 flops an entry), then 84 columns x (54 rows x <= 9 touched
    # links x 27 + 88 marker rows x ~40) multiply-adds
    jac = 2.0 * (3 * 89 * 9 * 60 + nc3 * (nq * 5 * 27 + (nrow - nq) * 40))
    return dict(k_dyn_eval=gram + chol, k_dyn_jac=jac, k_dyn_assemble=jtj + afj, k_dyn_schur=schur + nlat ** 3 / 3.0)


def bench_cfg4(torch, _lib, abi, skeleton, synth, dev, local, d4, N, n_cams, cpu=True):
    """config 4: physics-based model, phantom skeleton, N = 200, rotary gallop at 3 Hz with 12-frame stance, B unique sequences
    (seed 4321 + b), warm-started from the kinematic solve of the same measurements as the reference does (acinoset_opt.py:739-777).
    n_cams = 1: monocular with the pose prior, the way run_dataset.py:1198-1229 runs it; 6: the multi-view variant."""
    from cheetah_pose_estimation_amd import priors
    sk = skeleton.build_skeleton("phantom", 24)
    skk = skeleton.without_motion_model(sk)
    cams6 = synth.make_cameras(6)
    cams = (abi.Camera * 1)(cams6[2]) if n_cams == 1 else cams6
    pr_kin = priors.load_priors() if n_cams == 1 else None
    pr_dyn = priors.load_priors(pose=True, motion=False) if n_cams == 1 else None
    B = d4["q_init"].shape[0]
    T = upload(torch, d4, dev, keys=("q_init", "meas", "weight"))
    stance = torch.tensor(np.ascontiguousarray(d4["stance"], dtype=np.int32), device=dev)
    E = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
    q, dq, ddq, pos, me = E(B, N, sk.nq), E(B, N, sk.nq), E(B, N, sk.nq), E(B, N, 24, 3), E(B, N, n_cams, 24, 2)
    hk = _lib.Handle(sk, cams, abi.default_options(120.0), pr_kin, device=local)
    _, kst = hk.solve(T["q_init"], T["meas"], T["weight"], q, dq, ddq, pos, me)          # the kinematic estimate = the warm start
    hk.synchronize(); hk.close()
    q0 = q.clone()
    opts = abi.default_options(120.0); opts.tol_cost, opts.max_iter = 1e-6, 400      # (the slowest CONVERGING sequence of the batch needs 198 iterations; `converged_frac` states the rest)
    ko = abi.default_kinetic_options(skeleton.dyn_options("phantom"), 120.0)
    h = _lib.Handle(skk, cams, opts, pr_dyn, device=local)
    nm, nf, nc = ko.dyn.n_motors, ko.dyn.n_feet, h.n_constraint_rows()
    tau, lam, grf, slack = E(B, N, nm), E(B, N, nc), E(B, N, nf, 5), E(B, N, sk.nq)
    h.profile(True)
    _, wstats, _ = h.solve_kinetic(ko, q0, T["meas"], T["weight"], stance, q, dq, ddq, pos, me, tau, lam, grf, slack)
    prof = h.profile_totals()
    h.profile(False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    _, stats, ks = h.solve_kinetic(ko, q0, T["meas"], T["weight"], stance, q, dq, ddq, pos, me, tau, lam, grf, slack)
    h.synchronize()
    el = time.perf_counter() - t0
    its = np.array([s_.iterations for s_ in stats]); stt = np.array([s_.status for s_ in stats])
    wni = float((np.array([s_.iterations for s_ in wstats]) + 1).sum()) * (N - 2)    # node-iterations of the profiled run
    nrow, nlat, nc3, nq = sk.nq + 4 * nf + 3 * 24, nm + nc + 3 * nf, 84, sk.nq
    KP = nc3 * nc3 + 64 * nc3 + 64 * 64
    byts = dict(k_dyn_eval=8 * (3 * 66 + 64 + 76 + 2 * nq + 2 * nrow + nq * 64 + nq * nlat + 64 * 64 + 64 + 8 + nq),      # states, warm start, multipliers in; row gradient / weight, A (out, and read back once for e = r0 - A f), H_ff, forces, record, slack out
                k_dyn_jac=8 * (3 * 66 + 64 + nrow * nc3),                                                       # states, forces in; J out
                k_dyn_assemble=8 * (nrow * nc3 + 2 * nrow + nq * 64 + nc3 * nc3 + 64 * nc3 + nc3),                 # J, A in; H_uu, H_fu, gradient out
                k_dyn_schur=8 * (KP + 64 + 68 + 6 * 28 * 28),                                                       # the three pieces in; six blocks out
                k_dyn_gather=8 * (11 * 28 * 28 + 5 * 28))            # measurement block + the six blocks of three nodes that touch the frame in; B, three band blocks, gradient out
    fl = kinetic_flops_per_node(nq, nrow, nlat, nc3, nm + nc)
    ms = {k: v[0] for k, v in prof.items()}; nl = {k: v[1] for k, v in prof.items()}
    kern = {}
    for k in ("k_dyn_eval", "k_dyn_jac", "k_dyn_assemble", "k_dyn_schur", "k_dyn_gather"):
        kern[k] = dict(ms_total=ms.get(k, 0.0), launches=nl.get(k, 0), bytes_per_node_iteration=byts[k], roofline=roof_bytes(byts[k] * wni, ms.get(k, 0.0)),
                       fp64=roof_flops(fl[k] * wni, ms.get(k, 0.0)) if k in fl else None)
    for k in ("k_frame_normal", "k_lm_step", "k_lm_back"):
        kern[k] = dict(ms_total=ms.get(k, 0.0), launches=nl.get(k, 0))
    tot = sum(v["ms_total"] for v in kern.values())
    dom = max(("k_dyn_eval", "k_dyn_jac", "k_dyn_assemble", "k_dyn_schur", "k_dyn_gather"), key=lambda k: kern[k]["ms_total"])
    out = dict(value=B / el, unit="solves/s", workload=f"cfg4: physics-based model, 200 frames x {n_cams} camera(s) x 24 markers, rotary gallop 3 Hz, 12-frame stance, seed 4321 + b, at most 400 iterations"
               + (", pose prior, monocular warm start" if n_cams == 1 else ""),
               batch=B, seconds=el, iterations_mean=float(its.mean()), iterations_max=int(its.max()), converged_frac=float((stt == 0).mean()),
               warm_start_converged_frac=float(np.mean([s_.status == 0 for s_ in kst])), max_slack=float(max(k_.max_slack for k_ in ks)),
               kernels=kern, kernel_ms_sum=tot, dominant_kernel=dom, roofline=kern[dom]["roofline"])
    if cpu:
        from oracle import oracle as O
        O.lib()
        q0h = q0[0].cpu().numpy()
        t1 = time.perf_counter()
        ro = O.solve_kinetic(skk, cams, opts, pr_dyn, ko, q0h, d4["meas"][0], d4["weight"][0], d4["stance"][0])
        tm = time.perf_counter() - t1
        cores = min(16, usable_cores()[0])
        # one thread on a bounded sample: the first 40 frames of the same sequence (a 200-frame solve on one thread takes minutes)
        n1 = 40
        os.environ["CPO_THREADS"] = "1"
        t2 = time.perf_counter()
        r1 = O.solve_kinetic(skk, cams, opts, pr_dyn, ko, q0h[:n1], d4["meas"][0][:n1], d4["weight"][0][:n1], d4["stance"][0][:n1])
        t1s = time.perf_counter() - t2
        del os.environ["CPO_THREADS"]
        rm = float(np.sqrt(((pos[0].cpu().numpy() - ro["positions"]) ** 2).sum(-1).mean()))
        out["cpu_baseline"] = dict(value=1.0 / tm, unit="solves/s", cores=cores, kind="port",
                                   sample=f"sequence 0 of the same batch, oracle/cpe_oracle_kinetic.inc, OpenMP over nodes, {tm:.1f} s, {int(ro['stats'].iterations)} iterations",
                                   single_thread=dict(frames_per_s=n1 / t1s, unit="frames/s", cores=1, sample=f"first {n1} frames of that sequence, {t1s:.1f} s, {int(r1['stats'].iterations)} iterations"),
                                   rmse_gpu_vs_oracle_m=rm)
    h.close()
    return out


def pmc_traffic(B, N, C, L, kernel="k_resjac"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_pmc*.json, tools/summarise_pmc.py: separate
    FETCH_SIZE / WRITE_SIZE runs, read side doubled per MI355X_MICROARCH.md), scaled per frame.  The newest round's file wins.
    None if no profile for this marker / camera count (or kernel) is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc*.json"))):
        try:
            with open(f) as fh:
                j = json.load(fh)
            if "kernels" in j:                                   # round 2 format: several kernels, frames per launch stated
                if j.get("config", {}).get("C") != C or j.get("config", {}).get("L") != L:
                    continue
                for name, k in j["kernels"].items():
                    if k and kernel in name:
                        best = k["total_corrected"] / j["frames_per_launch"] * (B * N)
            else:                                                # round 1 format: k_resjac only
                c = j["config"]
                if kernel == "k_resjac" and c["C"] == C and c["L"] == L:
                    best = j["hbm_bytes_per_launch"]["total_corrected"] / (c["B"] * c["N"]) * (B * N)
        except Exception:
            pass
    return best


def pmc_valu_active(kernel="k_resjac<false"):
    """SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of `kernel` from the newest committed SQ pass (profiles/rNN_pmc_sq.csv, tools/profile_round.sh): the share
    of a wave's cycles with a vector instruction in flight (SURVEY 8d asks for it beside the achieved GB/s).  None if no such record."""
    import csv
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_sq.csv"))):
        try:
            with open(f, newline="") as fh:
                rows = [r_ for r_ in csv.DictReader(fh) if kernel in r_["kernel"]]
            v = {r_["counter"]: float(r_["mean_per_dispatch"]) for r_ in rows}
            if v.get("SQ_WAVE_CYCLES"):
                best = {"valu_active_frac": v.get("SQ_ACTIVE_INST_VALU", 0.0) / v["SQ_WAVE_CYCLES"], "wait_any_frac": v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"],
                        "source": os.path.basename(f)}
        except Exception:
            pass
    return best


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this parent never imports torch
    or touches the GPU), one per GPU, rendezvous on 127.0.0.1; rank 0 prints the JSON line.  Returns the worst exit status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10, help="untimed launches; the clocks of an idle MI355X need ~10 launches (30 ms) to ramp")
    ap.add_argument("--preheat", type=float, default=1.5, help="seconds of sustained untimed launches before the warm-up: an idle MI355X needs ~1 s of load to reach its steady memory rate")
    ap.add_argument("--batch", type=int, default=2048, help="sequences per GPU for the residual+Jacobian pass")
    ap.add_argument("--solve-batch", type=int, default=8192, help="sequences per GPU for the solve timing (8 MB of solver workspace each: 64 GB of the 288 GB)")
    ap.add_argument("--markers", type=int, default=25)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-solve", action="store_true")
    ap.add_argument("--no-l24", action="store_true", help="skip the extra 24-marker residual+Jacobian measurement")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-3 (monocular + learned priors) and config-4 (physics-based) solve timings")
    ap.add_argument("--cfg3-batch", type=int, default=2048, help="sequences of the config-3 timing: 8 launch windows of 256 (one workgroup per CU), a finished sequence hands its slot to the next one")
    ap.add_argument("--cfg4-batch", type=int, default=512, help="sequences of the config-4 timing (one launch window; 44 GB of node workspace)")
    ap.add_argument("--gen-workers", type=int, default=0, help="processes that generate the synthetic sequences (0 = as many as the CPU share allows; 1 under a profiler, whose preloaded tool does not survive fork)")
    ap.add_argument("--cfg4-cams", type=int, default=6, choices=(1, 6), help="cameras of the physics-based timing (1 = monocular + pose prior, as the reference runs it)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))                    # nothing above this line has initialised the GPU

    import torch
    import torch.distributed as dist
    from cheetah_pose_estimation_amd import _lib, abi, sharding, skeleton, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("CPE_BENCH_DRYRUN") == "1":
        # CPU rehearsal of the launch plumbing only (tests/test_bench_contract.py): rendezvous over gloo, the dealing of the
        # sequences, the barrier, the MAX over ranks and rank 0's single JSON line -- no GPU, no kernels, no numbers
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
        mine = sharding.shard_indices(world * 4, rank, world)
        owned = sharding.gather_by_index([int(i) for i in mine], world * 4, rank, world)
        tmax = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "owned": owned, "max_over_ranks": float(tmax.item())}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    # CPE_BENCH_REHEARSAL=1: rehearse the multi-rank path on a box with fewer GPUs -- every rank uses cuda:0 and the two collectives
    # (timing barrier, MAX over ranks) go over gloo.  The numbers of such a run mean nothing; it checks the plumbing.
    rehearsal = os.environ.get("CPE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    L, N, C = args.markers, args.frames, 6
    sk = skeleton.build_skeleton("phantom", L)
    cams = synth.make_cameras(C)
    opts = abi.default_options(120.0)
    # ---- synthetic inputs, generated BEFORE this process touches the GPU (the generator forks numpy-only workers).  SURVEY 8d: the sequence with
    # global index b has seed 1234 + b; the global list is dealt to the ranks round-robin: no two ranks (and no two slots of a rank) share one
    B = args.batch
    Bgen = max(B, 0 if args.no_solve else args.solve_batch)
    workers = args.gen_workers if args.gen_workers > 0 else max(1, min(16, usable_cores()[0] // max(1, min(world, 8))))
    mine = sharding.shard_indices(world * Bgen, rank, world)
    d = make_sequences("run", L, None, N, [1234 + int(i) for i in mine], workers)
    d24 = make_sequences("run", 24, None, N, [1234 + int(i) for i in mine[:B]], workers) if (L == 25 and world == 1 and not args.no_l24) else None
    extra = world == 1 and not args.no_solve and not args.no_extra
    d3 = make_sequences("run", 24, [2], N, [1234 + i for i in range(args.cfg3_batch)], workers) if extra else None
    d4 = make_sequences("gallop", 24, [2] if args.cfg4_cams == 1 else None, N, [4321 + i for i in range(args.cfg4_batch)], workers) if extra else None
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

    progress(f"rank {rank}: inputs generated ({Bgen} sequences)")
    h = _lib.Handle(sk, cams, opts, device=local)
    S = h.S
    t = upload(torch, {k: v[:B] for k, v in d.items()}, dev)
    r = torch.empty((B, N, C, L, 2), dtype=torch.float64, device=dev)
    J = torch.empty((B, N, C, S, 2), dtype=torch.float64, device=dev)
    eps = torch.empty((B, N, sk.nq), dtype=torch.float64, device=dev)
    cost = torch.empty((B, N), dtype=torch.float64, device=dev)
    stream = torch.cuda.ExternalStream(h.stream, device=dev)               # events on the stream the kernels run on

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def time_resjac(hh, tt, rr, JJ, ee, cc, n, st):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs:
            a.record(st)
            hh.eval_resjac(tt["q_true"], tt["meas"], tt["weight"], rr, JJ, ee, cc)
            b.record(st)
        hh.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs]))

    # Steady state first.  In a fresh process this store-bound kernel can run at 2.73 ms per launch for about its first second and at 2.43 ms from then
    # on, while the build that computes J without storing it takes 1.85 ms either way (tools/ab_resjac_box.py on three boxes: 20 launches after 3
    # warm-up launches 2.72 - 2.74 ms, 400 launches in a row 2.75 ms on average, 20 launches after 400 warm-up launches 2.44 ms; a second handle in the
    # same process 2.43 - 2.47 ms).  It is the memory side that settles -- whether clocks or the driver's background work on freshly allocated memory
    # was not established.  (Boxes differ as well: one ran 2.64 ms with and without this.)  `--preheat` seconds of the same launches, untimed, come before the
    # W warm-up launches; the line states them.
    n_heat, t_heat = 0, time.perf_counter()
    while time.perf_counter() - t_heat < args.preheat:
        for _ in range(25):
            h.eval_resjac(t["q_true"], t["meas"], t["weight"], r, J, eps, None)
        h.synchronize()
        n_heat += 25
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
    progress(f"rank {rank}: residual + Jacobian pass timed ({kern_ms:.3f} ms per launch)")
    # the variant that also reads the weights and writes the robust cost (k_resjac<true>), for the record
    ms_cost = time_resjac(h, t, r, J, eps, cost, 5, stream) if world == 1 else None

    # "also report L=24" -- the reference's own 24 markers, same cameras and sequence shape, same kernel
    l24 = None
    if d24 is not None:
        del r, J, eps
        sk24 = skeleton.build_skeleton("phantom", 24)
        h24 = _lib.Handle(sk24, cams, opts, device=local)
        t24 = upload(torch, d24, dev)
        r = torch.empty((B, N, C, 24, 2), dtype=torch.float64, device=dev)
        J = torch.empty((B, N, C, h24.S, 2), dtype=torch.float64, device=dev)
        eps = torch.empty((B, N, sk24.nq), dtype=torch.float64, device=dev)
        s24 = torch.cuda.ExternalStream(h24.stream, device=dev)
        for _ in range(max(2, args.warmup)):
            h24.eval_resjac(t24["q_true"], t24["meas"], t24["weight"], r, J, eps, None)
        h24.synchronize()
        ms24 = time_resjac(h24, t24, r, J, eps, None, 5, s24)
        b24 = resjac_bytes_per_frame(C, 24, h24.S, sk24.nq, False)
        l24 = dict(value=B * N / (ms24 * 1e-3), unit="frames/s", kernel_ms=ms24, bytes_per_frame=b24, frac=b24 * B * N / (ms24 * 1e-3) / HBM_PEAK)
        h24.close()
        del t24

    solves = None
    if not args.no_solve:
        Bs = args.solve_batch
        del t
        ts_ = upload(torch, {k: v[:Bs] for k, v in d.items()}, dev, keys=("q_init", "meas", "weight"))
        r = J = eps = cost = None
        q = torch.empty((Bs, N, sk.nq), dtype=torch.float64, device=dev); dq = torch.empty_like(q); ddq = torch.empty_like(q)
        pos = torch.empty((Bs, N, L, 3), dtype=torch.float64, device=dev); me = torch.empty((Bs, N, C, L, 2), dtype=torch.float64, device=dev)
        # warm-up at full size (the solver workspace, 16 GB for 2048 sequences, is allocated here), with the per-kernel HIP-event
        # profile switched on: this untimed run supplies the kernel times of the roofline object
        h.profile(True)
        _, wstats = h.solve(ts_["q_init"], ts_["meas"], ts_["weight"], q, dq, ddq, pos, me)
        prof = h.profile_totals()
        h.profile(False)
        barrier()
        t1 = time.perf_counter()
        st, stats = h.solve(ts_["q_init"], ts_["meas"], ts_["weight"], q, dq, ddq, pos, me)
        barrier()
        el = time.perf_counter() - t1
        el = max_over_ranks(el)
        its = np.array([s.iterations for s in stats])
        stt = np.array([s.status for s in stats])
        # every LM iteration of a sequence evaluates and factors all its N frames (+1: the first evaluation)
        frame_its = float((its + 1).sum()) * N
        fn_b, lm_b = solve_bytes_per_frame_iteration(C, L, sk.nq, h.nu, 12, 3)
        lm_ms, lm_n = prof.get("k_lm_step", (0.0, 0))
        bk_ms, _ = prof.get("k_lm_back", (0.0, 0))
        fwd_ms = lm_ms
        lm_ms += bk_ms                                    # the step is two kernels since round 2: factor (k_lm_step) + solve (k_lm_back)
        fn_ms, fn_n = prof.get("k_frame_normal", (0.0, 0))
        wframe_its = float((np.array([s.iterations for s in wstats]) + 1).sum()) * N
        roof = {"bound": "hbm", "kernel": "k_lm_step<3> + k_lm_back<3>", "unit": "GB/s", "peak": HBM_PEAK / 1e9,
                "achieved": (lm_b * wframe_its / (lm_ms * 1e-3) / 1e9) if lm_ms else None,
                "frac": (lm_b * wframe_its / (lm_ms * 1e-3) / HBM_PEAK) if lm_ms else None,
                "traffic": (lambda a_, b_: a_ + b_ if a_ is not None and b_ is not None else a_)(pmc_traffic(min(Bs, 512), N, C, L, "k_lm_step"), pmc_traffic(min(Bs, 512), N, C, L, "k_lm_back")),   # both kernels, per full launch window (512 sequences)
                "bytes_per_frame_iteration": lm_b, "kernel_ms_total": lm_ms, "launches": lm_n,
                "ms_per_launch": {"k_lm_step": fwd_ms / lm_n if lm_n else None, "k_lm_back": bk_ms / lm_n if lm_n else None,
                                  "k_frame_normal": fn_ms / fn_n if fn_n else None},
                "fp64": {"achieved_tflops": (lm_flops_per_frame_iteration() * wframe_its / (lm_ms * 1e-3) / 1e12) if lm_ms else None,
                         "peak_tflops": FP64_PEAK / 1e12,
                         "frac": (lm_flops_per_frame_iteration() * wframe_its / (lm_ms * 1e-3) / FP64_PEAK) if lm_ms else None},
                "k_frame_normal": {"bytes_per_frame_iteration": fn_b, "kernel_ms_total": fn_ms, "launches": fn_n,
                                   "frac": (fn_b * wframe_its / (fn_ms * 1e-3) / HBM_PEAK) if fn_ms else None},
                "whole_solve": {"bytes_per_frame_iteration": fn_b + lm_b, "achieved": (fn_b + lm_b) * frame_its / el / 1e9,
                                "frac": (fn_b + lm_b) * frame_its / el / HBM_PEAK},
                "note": "latency-bound (200 sequential block columns x 28 pivots per sequence, two 80 KB workgroups per CU): far from both rooflines, see DESIGN.md 6"}
        solves = dict(value=world * Bs / el, unit="solves/s", batch_per_gpu=Bs, seconds=el, iterations_mean=float(its.mean()),
                      iterations_max=int(its.max()), iterations_p99=float(np.percentile(its, 99)), converged_frac=float((stt == 0).mean()), roofline=roof)
        if world == 1 and Bs > 2048:
            # the batch size of rounds 1-2, for continuity: with one seed per sequence the slowest of 2048 (93 iterations against 22 on average) leaves
            # the launches half empty for a third of the run; the larger batch above amortises that drain
            t2_ = time.perf_counter()
            _, st2 = h.solve(ts_["q_init"][:2048], ts_["meas"][:2048], ts_["weight"][:2048], q[:2048], dq[:2048], ddq[:2048], pos[:2048], me[:2048])
            h.synchronize()
            e2 = time.perf_counter() - t2_
            i2 = np.array([s_.iterations for s_ in st2])
            solves["batch_2048"] = dict(value=2048 / e2, seconds=e2, iterations_mean=float(i2.mean()), iterations_max=int(i2.max()))
        if world == 1:
            # latency of ONE sequence through the drop-in path's solver (B = 1: one workgroup on one CU), N = 200 and N = 57
            lat = {}
            for n1 in (N, 57):
                d1 = synth.make_batch(sk, cams, B=1, N=n1, seed=1234)
                t1_ = {k: torch.tensor(v, device=dev) for k, v in d1.items()}
                o = [torch.empty((1, n1, sk.nq), dtype=torch.float64, device=dev) for _ in range(3)]
                p1 = torch.empty((1, n1, L, 3), dtype=torch.float64, device=dev); m1 = torch.empty((1, n1, C, L, 2), dtype=torch.float64, device=dev)
                torch.cuda.synchronize(dev)
                times = []
                for _ in range(4):
                    ta = time.perf_counter()
                    _, s1 = h.solve(t1_["q_init"], t1_["meas"], t1_["weight"], o[0], o[1], o[2], p1, m1)
                    h.synchronize()
                    times.append(time.perf_counter() - ta)
                lat[f"N{n1}"] = dict(ms=1e3 * float(np.median(times[1:])), iterations=int(s1[0].iterations), status=int(s1[0].status))
            solves["latency_b1"] = lat

    if solves is not None:
        progress(f"rank {rank}: solves timed ({solves['value']:.0f} solves/s)")
    cfg3 = cfg4 = None
    if extra:
        h.close()                                          # its 16 GB workspace goes back before the next handles are made
        ts_ = q = dq = ddq = pos = me = None
        torch.cuda.empty_cache()
        cfg3, q3 = bench_cfg3(torch, _lib, abi, skeleton, synth, dev, local, d3, N, cpu=not args.no_cpu)
        progress(f"config 3 timed ({cfg3['value']:.1f} solves/s)")
        cfg4 = bench_cfg4(torch, _lib, abi, skeleton, synth, dev, local, d4, N, args.cfg4_cams, cpu=not args.no_cpu)
        progress(f"config 4 timed ({cfg4['value']:.2f} solves/s)")
        h = None

    if rank == 0:
        bpf = resjac_bytes_per_frame(C, L, S, sk.nq, False)
        ach = bpf * B * N / (kern_ms * 1e-3)
        out = {
            "metric": "frames/sec residual+Jacobian eval + full-traj solves/sec, 200-frame 6-cam seq",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "preheat": {"seconds": args.preheat, "launches": n_heat, "note": "untimed launches of the same kernel before the warm-up: steady memory rate of the device (tools/ab_resjac_box.py)"},
            "config": {"workload": "cfg2: synthetic 200-frame x 6-cam x 25-marker sequences, phantom skeleton, const-accel model",
                       "frames": N, "cams": C, "markers": L, "sequences_per_gpu": B, "parallelism": f"shard{world} (independent sequences, no collective)"},
            "solves": solves, "solves_cfg3": cfg3, "solves_cfg4": cfg4, "markers24": l24,
            "roofline": {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                         "traffic": pmc_traffic(B, N, C, L), "kernel": "k_resjac<false>", "kernel_ms": kern_ms, "bytes_per_frame": bpf, "sq": pmc_valu_active()},
        }
        if ms_cost is not None:
            bc = resjac_bytes_per_frame(C, L, S, sk.nq, True)
            out["with_cost"] = {"kernel": "k_resjac<true>", "kernel_ms": ms_cost, "bytes_per_frame": bc, "value": B * N / (ms_cost * 1e-3),
                                "frac": bc * B * N / (ms_cost * 1e-3) / HBM_PEAK}
        if not args.no_cpu and world == 1:          # the CPU leg is timed on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(sk, cams, opts, {k: v[:32] for k, v in d.items()})
        print(json.dumps(out), flush=True)
    if h is not None:
        h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
