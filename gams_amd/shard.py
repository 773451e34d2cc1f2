"""Contig sharding across ranks/devices.

Every ctg is an independent unit in the reference (src/cmd_gams/wave.rs:294-297: one ctg per
worker, windows and z-score state never cross a ctg), so N GPUs take disjoint ctg subsets and
nothing is exchanged on the data path; per-ctg TSV rows gather on the host.  Assignment is
longest-processing-time-first by window count.
"""
import heapq


def lpt_assign(weights, n_ranks):
    """Greedy LPT: returns a list (rank per item).  Deterministic: ties by index."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    heap = [(0, r) for r in range(n_ranks)]
    heapq.heapify(heap)
    out = [0] * len(weights)
    for i in order:
        load, r = heapq.heappop(heap)
        out[i] = r
        heapq.heappush(heap, (load + weights[i], r))
    return out


def shard_ctgs(ctgs, n_ranks, rank, size=100, step=10):
    """Indices of the ctgs `rank` owns (weights = windows per ctg)."""
    w = [max(0, (len(c["seq"]) - size) // step + 1) for c in ctgs]
    owner = lpt_assign(w, n_ranks)
    return [i for i, r in enumerate(owner) if r == rank]


def gather_rows(per_ctg_rows, owned, n_ctgs, dist=None):
    """Rank 0 receives every ctg's rows in ctg order; other ranks get None.
    per_ctg_rows: list aligned with `owned` (this rank's ctg indices)."""
    mine = dict(zip(owned, per_ctg_rows))
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [mine[i] for i in range(n_ctgs)]
    bucket = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(mine, bucket, dst=0)
    if dist.get_rank() != 0:
        return None
    merged = {}
    for b in bucket:
        merged.update(b)
    return [merged[i] for i in range(n_ctgs)]
