"""Multi-GPU: independent video sequences shard embarrassingly across the GPUs of one node (SURVEY 8e;
the reference loops over sequences one at a time, run_dataset.py:1145).  One process per GPU, sequence
b -> rank b mod G, NO collective on the solve path; the only communication is a host-side gather of the
per-sequence results at the end."""
from typing import Any, List, Sequence


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """round-robin ownership: item b belongs to rank b mod world"""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_items, world))


def gather_by_index(local: Sequence[Any], n_items: int, rank: int, world: int) -> List[Any]:
    """host-side gather (torch.distributed, any backend) of per-sequence python objects back into item order.
    Every rank returns the full list.  Not on the timed path."""
    if world == 1:
        return list(local)
    import torch.distributed as dist
    parts: List[Any] = [None] * world
    dist.all_gather_object(parts, list(local))
    out: List[Any] = [None] * n_items
    for r in range(world):
        for k, b in enumerate(shard_indices(n_items, r, world)):
            out[b] = parts[r][k]
    return out
