"""N > 1 path on CPU: world_size 2, gloo.  Sequences are sharded round-robin with no data-path collective;
the host-side gather returns results in sequence order on every rank."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cheetah_pose_estimation_amd import sharding, skeleton, synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.shard_indices(n_items, rank, world)
    # each rank 'solves' its own sequences: here, the host-side FK of the seeded truth trajectory
    sk = skeleton.build_skeleton("phantom", 25)
    local = []
    for b in mine:
        qt = synth.truth_trajectory(sk, 6, 120.0, np.random.default_rng(1234 + b))
        local.append(float(synth.fk_numpy(sk, qt)[0].sum()))
    full = sharding.gather_by_index(local, n_items, rank, world)
    q.put((rank, mine, full))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    world, n_items = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    owned = sorted(sum((g[1] for g in got), []))
    assert owned == list(range(n_items))                           # every sequence exactly once
    sk = skeleton.build_skeleton("phantom", 25)
    want = [float(synth.fk_numpy(sk, synth.truth_trajectory(sk, 6, 120.0, np.random.default_rng(1234 + b)))[0].sum()) for b in range(n_items)]
    for g in got:
        assert np.allclose(g[2], want, rtol=0, atol=1e-12)         # same, ordered result on every rank


def test_shard_indices_partition():
    for world in (1, 2, 3, 8):
        allb = sorted(sum((sharding.shard_indices(29, r, world) for r in range(world)), []))
        assert allb == list(range(29))
