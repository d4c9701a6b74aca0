"""Static sharding of windows over the GPUs of one node.

Windows are independent (docs/guides/architecture.md:124 in the reference): window i goes to rank
i mod G, results return to the host store, NO collective on the data path.  The only cross-rank
communication a run needs is the barrier + max-over-ranks of the elapsed time that bench.py reports.
"""


def shard_indices(n_windows, rank, world):
    """Interleaved static assignment (balances the 14x slower complex windows, graph_complexity.h:99)."""
    return list(range(rank, n_windows, world))


def merge_shards(per_rank_results, n_windows, world):
    """Inverse of shard_indices: per_rank_results[r][j] is the result of window shard_indices(n, r, world)[j]."""
    out = [None] * n_windows
    for r in range(world):
        for j, w in enumerate(shard_indices(n_windows, r, world)):
            out[w] = per_rank_results[r][j]
    return out


def max_over_ranks(value, dist=None):
    """Whole-job time = max over ranks (all ranks hold the result)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
