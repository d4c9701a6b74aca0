"""Static sharding of windows over the GPUs of one node.

Windows are independent (docs/guides/architecture.md:124 in the reference): window i goes to rank
(i // block) mod G, results return to the host store, NO collective on the data path.  The only cross-rank
communication a run needs is the barrier + max-over-ranks of the elapsed time that bench.py reports.
"""


def shard_indices(n_windows, rank, world, block=1):
    """Interleaved static assignment (balances the 14x slower complex windows, graph_complexity.h:99): window i goes to
    rank (i // block) mod world -- `block` consecutive windows at a time ("i mod G in batches").  A window list
    whose difficult windows recur with a PERIOD (bench.py: every 8th / 16th / 32nd) has to be dealt out in blocks of that
    period, or some rank gets all of one kind and none of another."""
    block = max(1, int(block))
    return [i for b0 in range(rank * block, n_windows, world * block) for i in range(b0, min(b0 + block, n_windows))]


def merge_shards(per_rank_results, n_windows, world, block=1):
    """Inverse of shard_indices: per_rank_results[r][j] is the result of window shard_indices(n, r, world, block)[j]."""
    out = [None] * n_windows
    for r in range(world):
        for j, w in enumerate(shard_indices(n_windows, r, world, block)):
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
