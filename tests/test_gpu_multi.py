"""config 5 (BASELINE.json: "full data/test_set x all cheetah skeletons, independent sequences sharded across the GPUs") at test size: two
processes, one rank each (both on cuda:0 here -- a one-GPU box; the driver's scaling run uses one GPU per rank), the (skeleton, sequence) items
dealt round-robin by sharding.shard_indices, every rank solving ITS items on the GPU through the C ABI, results gathered on the host over gloo.
No collective on the solve path; the gathered trajectories equal those of a single process bit for bit (solves are reproducible)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ITEMS = [(animal, seed) for animal in ("phantom", "jules", "arabia-02", "shiraz-02") for seed in (31, 32)]        # the four skeletons of the reference x sequences


def _solve_item(animal, seed):
    from cheetah_pose_estimation_amd import _lib, abi, skeleton, synth
    kin = animal.endswith("-02")
    sk = skeleton.build_skeleton(animal, 24, kinetic_dataset=kin)
    cams = synth.make_cameras(4 if kin else 6)
    d = synth.make_batch(sk, cams, B=1, N=24, seed=seed, kinetic_dataset=kin)
    h = _lib.Handle(sk, cams, abi.default_options(200.0 if kin else 120.0))
    try:
        out = h.solve_host(d["q_init"], d["meas"], d["weight"])
    finally:
        h.close()
    return out["q"][0].tobytes(), int(out["stats"][0].status), int(out["stats"][0].iterations)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from cheetah_pose_estimation_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.shard_indices(len(ITEMS), rank, world)
    local = [_solve_item(*ITEMS[i]) for i in mine]                 # GPU work of this rank: its own sequences only
    full = sharding.gather_by_index(local, len(ITEMS), rank, world)
    if rank == 0:
        q.put((mine, full))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sequences_over_two_ranks_equal_one_process():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    mine0, full = q.get(timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert mine0 == [0, 2, 4, 6] and len(full) == len(ITEMS)
    for i, item in enumerate(ITEMS):
        qb, st, it = _solve_item(*item)                            # the same item in THIS process
        assert full[i][1] == st == 0 and full[i][2] == it
        assert full[i][0] == qb                                    # bit for bit
        assert np.isfinite(np.frombuffer(qb)).all()


def test_dataset_entry_batches_and_shards_sequences(tmp_path):
    """run_kinematics_dataset -- the loop of run_dataset.py:1143-1196 as a product function.  Five synthetic sequences on disk
    (two lengths: two solver groups), (a) one at a time through estimate_kinematics, (b) batched in this process (estimate_kinematics_batch groups
    them by skeleton / rig / length into cpe_solve calls with B > 1), (c) dealt to two fresh processes, one rank each (both on cuda:0 on this
    one-GPU box), by sharding.shard_indices.  Every sequence's fte.pickle is the same in all three, bit for bit."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from cheetah_pose_estimation_amd import estimator as E
    from dataset_util import write_dataset
    roots = {k: str(tmp_path / k) for k in ("single", "batch", "multi")}
    specs = [("2019_03_07/synth/run1", 24, 5), ("2019_03_07/synth/run2", 24, 6), ("2019_03_07/synth/run3", 30, 7), ("2019_03_07/synth/run4", 24, 8), ("2019_03_07/synth/run5", 30, 9)]
    for root in roots.values():
        for path, N, seed in specs:
            write_dataset(root, data_path=path, N=N, seed=seed)
    jobs = lambda root: [dict(root_dir=root, data_path=path, cheetah_name="phantom", kinetic_dataset=False, solver_path="/unused/ipopt", kinematic_model=True)
                         for path, _, _ in specs]
    oks_single = [E.estimate_kinematics(E.init_trajectory(**j), solver_output=False) for j in jobs(roots["single"])]
    oks_batch = E.run_kinematics_dataset(jobs(roots["batch"]), devices=(0,))
    oks_multi = E.run_kinematics_dataset(jobs(roots["multi"]), devices=(0, 0))
    assert oks_single == oks_batch == oks_multi == [True] * 5
    for path, N, _ in specs:
        ref = E.load_result_pickle(os.path.join(roots["single"], path, "fte_kinematic", "fte.pickle"))
        assert ref["q"].shape == (N, 54)
        for k in ("batch", "multi"):
            got = E.load_result_pickle(os.path.join(roots[k], path, "fte_kinematic", "fte.pickle"))
            for key in ("q", "dq", "ddq", "positions", "x", "com_pos", "meas_err"):
                assert np.array_equal(ref[key], got[key]), (k, path, key)
            assert got["obj_cost"] == ref["obj_cost"] and os.path.exists(os.path.join(roots[k], path, "fte_kinematic", "cam6_fte.csv"))
