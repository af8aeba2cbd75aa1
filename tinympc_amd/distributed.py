"""Multi-GPU plumbing: one process per GPU, the batch sharded with NO data-path collective.

Instances are independent QPs, so each rank owns a contiguous (or, for
divergent iteration counts, round-robin) slice of the batch for the whole episode; the only
exchange is the reduction of the 10-double statistics vector produced by tiny_batch_reduce_stats:
SUM over {sum_iter, sum_solved, batch, accumulated iters, accumulated solved}, MAX over the four
residual maxima -- two 80-byte all-reduces.  With backend "nccl" this is RCCL over xGMI (latency bound);
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
    # Measured on one MI355X (torchrun, one rank, the bench workload): packing both reductions into ONE collective -- an
    # all-gather reduced locally, or a MAX all-reduce over one-hot count slots -- needs a handful of extra small device ops
    # and came out 5-6 % slower end to end than these two plain all-reduces, so they stay.
    total = stats.clone()
    peak = stats.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)     # counts
    dist.all_reduce(peak, op=dist.ReduceOp.MAX, group=group)      # residual maxima
    total[MAX_IDX[0]:MAX_IDX[-1] + 1] = peak[MAX_IDX[0]:MAX_IDX[-1] + 1]
    return total
