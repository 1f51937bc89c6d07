"""Read sharding for N GPUs of one node: reads are independent at every stage of the path, so a batch
is cut into contiguous read-id ranges balanced by bases, every rank maps its range against its own
replica of the index, and the per-rank cluster tables are concatenated in read-id order on rank 0.
No collective touches the data path; torch.distributed only carries the final tables.

What is sequential in pangraph_from_read_file (read ids, the coverage stop, the 10M-read stop, the Illumina
k-mer count taken from the first sketched read; src/utils.cpp:397-440) is decided before the reads are
sharded: a shard is a range of reads that are all going to be mapped, with their final ids."""
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


def merge_tables(parts, n_hits=None):
    """Concatenates per-shard dicts of equally keyed numpy arrays in shard order (= read-id order, since
    shards are contiguous ranges).  A `begin` column (first hit of a cluster in its shard's sorted hits,
    MapResult.clusters()) is moved to the numbering of the concatenated hits: n_hits[i] = hits of shard i."""
    out = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    if "begin" in out:
        if n_hits is None:
            raise ValueError("merging cluster tables with a `begin` column needs the hit count of every shard")
        base = np.concatenate([[0], np.cumsum(n_hits)[:-1]]).astype(out["begin"].dtype)
        out["begin"] = np.concatenate([p["begin"] + base[i] for i, p in enumerate(parts)])
    return out


def gather_tables(local, dst=0, n_hits=None):
    """Gathers per-rank dicts on rank `dst` and merges them (merge_tables); n_hits = this rank's hit count
    when the tables have a `begin` column.  Returns None on other ranks."""
    import torch.distributed as dist
    world = dist.get_world_size()
    parts = [None] * world if dist.get_rank() == dst else None
    dist.gather_object((local, n_hits), parts, dst=dst)
    if parts is None:
        return None
    counts = [c for _, c in parts]
    return merge_tables([t for t, _ in parts], None if any(c is None for c in counts) else counts)
