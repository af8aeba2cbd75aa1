"""Multi-GPU plumbing: one process per GPU, the batch sharded with NO data-path collective.

Instances are independent QPs, so each rank owns a contiguous (or, for
divergent iteration counts, round-robin) slice of the batch for the whole episode; the only
exchange is ONE collective on the 10-double statistics vector produced by tiny_batch_reduce_stats
(an all-gather of 80 B per rank, reduced locally: SUM over {sum_iter, sum_solved, batch, accumulated
iters, accumulated solved}, MAX over the four residual maxima -- a mixed SUM / MAX reduction would take
two all-reduces).  With backend "nccl" this is RCCL over xGMI (latency bound);
the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

COUNT_IDX = (0, 1, 2, 7, 8)
MAX_IDX = (3, 4, 5, 6)


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous block of rank `rank` when `total` instances are split over `world` GPUs."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_indices(total: int, rank: int, world: int, interleaved: bool = False):
    """Instance ids owned by `rank`; interleaved (round-robin) balances divergent iteration counts."""
    if interleaved:
        return list(range(rank, total, world))
    lo, hi = shard_bounds(total, rank, world)
    return list(range(lo, hi))


def allreduce_stats(stats, dist=None, group=None):
    """stats: 1-D float64 torch tensor of 10 entries (device or CPU) as written by
    TinyBatchSolver.reduce_stats(_async).  Returns the job-wide vector (all ranks get it)."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return stats.clone()
    # one collective: every rank gathers the world's 10-double vectors and reduces them itself
    world = dist.get_world_size(group=group)
    parts = [torch.empty_like(stats) for _ in range(world)]
    dist.all_gather(parts, stats.contiguous(), group=group)
    stacked = torch.stack(parts)                                   # [world, 10]
    total = stacked.sum(dim=0)                                     # counts
    total[MAX_IDX[0]:MAX_IDX[-1] + 1] = stacked[:, MAX_IDX[0]:MAX_IDX[-1] + 1].max(dim=0).values     # residual maxima
    return total
