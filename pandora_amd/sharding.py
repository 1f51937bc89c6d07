"""Read sharding for N GPUs of one node: reads are independent at every stage of the path, so a batch
is cut into contiguous read-id ranges balanced by bases, every rank maps its range against its own
replica of the index, and the per-rank cluster tables are concatenated in read-id order on rank 0.
No collective touches the data path; torch.distributed only carries the final tables."""
import numpy as np


def shard_bounds(offsets, world):
    """Contiguous read ranges [(begin, end), ...], one per rank, balanced by bases."""
    offsets = np.asarray(offsets, np.uint64)
    n = len(offsets) - 1
    total = int(offsets[-1] - offsets[0])
    bounds = [0]
    for r in range(1, world):
        target = int(offsets[0]) + total * r // world
        cut = int(np.searchsorted(offsets, target, side="left"))
        bounds.append(min(max(cut, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[i], bounds[i + 1]) for i in range(world)]


def shard(bases, offsets, rank, world):
    """(bases, offsets, first_read_id) of this rank's shard."""
    b, e = shard_bounds(offsets, world)[rank]
    offsets = np.asarray(offsets, np.uint64)
    lo, hi = int(offsets[b]), int(offsets[e])
    return bases[lo:hi], offsets[b:e + 1] - offsets[b], b


def gather_tables(local, dst=0):
    """Gathers per-rank dicts of equally keyed numpy arrays on rank `dst` and concatenates them in rank
    order (= read-id order, since shards are contiguous ranges).  Returns None on other ranks."""
    import torch.distributed as dist
    world = dist.get_world_size()
    parts = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(local, parts, dst=dst)
    if parts is None:
        return None
    return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
