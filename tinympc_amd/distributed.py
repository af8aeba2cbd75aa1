"""Multi-GPU plumbing: one process per GPU, the batch sharded with NO data-path collective.

Instances are independent QPs, so each rank owns a contiguous (or, for
divergent iteration counts, round-robin) slice of the batch for the whole episode; the only
exchange is ONE collective over a 64-byte message per rank: the eight statistics a shard cannot know
about the others -- {sum iter, sum solved, accumulated iterations, accumulated solves} and the four
residual maxima -- are all-gathered and every rank reduces the world x 8 table itself (SUM over the
counts, MAX over the residuals).  With backend "nccl" this is RCCL over xGMI (latency bound); the
same code runs on "gloo" for the CPU tests.  The native form of the same exchange is
tiny_group_allreduce_stats / tiny_batch_allreduce_stats (include/tinympc_amd.h).
"""
from __future__ import annotations

# positions in the 10-double vector of tiny_batch_reduce_stats
COUNT_IDX = (0, 1, 2, 7, 8)
MAX_IDX = (3, 4, 5, 6)
# the 64-byte wire message: four counts, then the four residual maxima (the shard sizes are known by construction)
WIRE_IDX = (0, 1, 7, 8, 3, 4, 5, 6)


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


def shard_size(total: int, rank: int, world: int, interleaved: bool = False) -> int:
    if interleaved:
        return len(range(rank, total, world))
    lo, hi = shard_bounds(total, rank, world)
    return hi - lo


_wire_index = {}


def reduce_table(table, total_batch=None):
    """Host-side reduction of the gathered table of wire messages -> the 10-entry statistics vector.  Rows are the 8-double
    wire message, optionally followed by the shard's own size (a 9th column): total_batch = None takes the job size from
    that column (exact for uneven and interleaved shards).  A NaN residual propagates (torch.max), as in the native
    reduce_wire_table."""
    import torch
    t = torch.as_tensor(table, dtype=torch.float64)
    if t.dim() == 1:                                                # one message: its length says whether the size column is there
        if t.numel() not in (len(WIRE_IDX), len(WIRE_IDX) + 1):
            raise ValueError("reduce_table: a flat table must be ONE wire message (8 doubles, or 9 with the shard size); "
                             "pass a 2-D [ranks][8 | 9] table otherwise")
        t = t.reshape(1, -1)
    if t.dim() != 2 or t.shape[1] not in (len(WIRE_IDX), len(WIRE_IDX) + 1):
        raise ValueError("reduce_table: expected a [ranks][8 | 9] table, got shape %s" % (tuple(t.shape),))
    cols = t.shape[1]
    out = torch.zeros(10, dtype=torch.float64)
    sums = t[:, :4].sum(dim=0)
    out[0], out[1], out[7], out[8] = sums[0], sums[1], sums[2], sums[3]
    out[3:7] = t[:, 4:8].max(dim=0).values
    if total_batch is None:
        if cols <= len(WIRE_IDX):
            raise ValueError("reduce_table: total_batch is needed when the table carries no shard-size column")
        total_batch = float(t[:, 8].sum())
    out[2] = float(total_batch)
    return out


def _world_rank(dist, group):
    return dist.get_world_size(group), dist.get_rank(group)


def allreduce_stats(stats, dist=None, group=None, total_batch=None):
    """stats: 1-D float64 torch tensor of 10 entries (device or CPU) as written by
    TinyBatchSolver.reduce_stats(_async).  Returns the job-wide vector as a **CPU** tensor whatever device `stats` lives
    on (every rank gets the same one; the reduction of the gathered table happens on the host).  total_batch: the
    unsharded batch size; None = the sum of the shards' own sizes (stats[2] of every rank), which travels as a 9th
    column of the same table -- exact for uneven and interleaved shards, no extra collective, no device read.

    The gather of the 64-byte messages is issued as ONE all-reduce(SUM) over a world x 8 table in which a
    rank fills only its own row (adding the other ranks' zeros is exact): on this stack a small all-reduce
    costs 20 us, all_gather_into_tensor of the same bytes 0.9 ms (measured, tools/dist_exchange_cost.py)."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return stats.detach().to("cpu").clone()
    world, rank = _world_rank(dist, group)
    idx = _wire_index.get(stats.device)
    if idx is None:                                                 # built once per device: no host->device copy per call
        idx = _wire_index[stats.device] = torch.tensor(WIRE_IDX, dtype=torch.long, device=stats.device)
    table = torch.zeros(world, len(WIRE_IDX) + 1, dtype=stats.dtype, device=stats.device)
    table[rank, :len(WIRE_IDX)] = stats.index_select(0, idx)        # 8 doubles = 64 bytes
    table[rank, len(WIRE_IDX)] = stats[2]                           # + this shard's size (device-side copy, no sync)
    dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)       # the one collective of the path
    return reduce_table(table.to("cpu"), total_batch)               # device -> host (synchronises), reduced here


class StatsExchange:
    """The exchange for a device-resident TinyBatchSolver on the library's NATIVE path: an RCCL communicator of our own
    (the ncclUniqueId of rank 0 travels over torch.distributed once, at construction) and tiny_batch_allreduce_stats --
    device reduction, 64-byte message, ONE ncclAllGather on the solver's stream, pinned device->host copy, host
    reduction.  No torch op sits between the last solve launch and the result."""

    def __init__(self, solver, dist, device_index, total_batch, group=None):
        import tinympc_amd as tm
        self.s, self.total = solver, int(total_batch)
        self.world, self.rank = _world_rank(dist, group)
        self.dist, self.group, self.comm, self.kind, self.comm_ranks = dist, group, None, "native", 0
        import os
        import torch
        # ncclCommInitRank is itself a collective: a rank that cannot take part (librccl missing on that rank only, a bad
        # device) must say so BEFORE any rank enters it, or the others would block inside init.  So the ranks first agree,
        # over the process group they already share, that every one of them can load RCCL and reach its device.
        can = os.environ.get("TINYMPC_EXCHANGE", "native") != "torch"                    # (=torch: force the fallback)
        try:
            can = can and tm.rccl_available() and 0 <= device_index < tm.device_count()
        except Exception:                            # noqa: BLE001
            can = False
        ok = self._agree(can)
        if ok:
            try:
                box = [tm.rccl_unique_id() if self.rank == 0 else None]
            except Exception as e:                   # noqa: BLE001  (rank 0 could not draw an id: every rank must learn it)
                box = [repr(e)]
            dist.broadcast_object_list(box, src=0, group=group)
            ok = isinstance(box[0], (bytes, bytearray))
            if ok:
                try:
                    self.comm = tm.rccl_comm_init_rank(self.world, bytes(box[0]), self.rank, device_index)
                    self.comm_ranks = tm.rccl_comm_count(self.comm)
                    ok = self.comm_ranks == self.world          # a communicator of another size is not the one the job agreed on
                except Exception:                    # noqa: BLE001
                    ok = False
        # all ranks or none: a rank whose join failed must not leave the others waiting inside the all-gather
        if not self._agree(ok):
            # Still RCCL, through torch.distributed: the same 64-byte messages as a one-hot all-reduce (allreduce_stats)
            if self.comm:
                tm.rccl_comm_destroy(self.comm)
            self.comm, self.kind = None, "torch.distributed"
            self._stats = torch.zeros(10, dtype=torch.float64, device=f"cuda:{device_index}")
            import sys
            print("tinympc_amd: native RCCL communicator unavailable, statistics exchange through torch.distributed", file=sys.stderr)

    def _agree(self, mine):
        """MIN over the ranks of a 0/1 flag.  The flag lives where the process group can certainly reach it -- the device the
        group was initialised on (torch.cuda.current_device(): a rank whose OWN device_index is bad still enters the collective
        and votes 0 instead of raising on its own and leaving the others inside it), host memory for gloo."""
        import torch
        backend = str(self.dist.get_backend(self.group)) if hasattr(self.dist, "get_backend") else "nccl"
        try:
            dev = torch.device("cpu") if "gloo" in backend else torch.device("cuda", torch.cuda.current_device())
            flag = torch.tensor([1.0 if mine else 0.0], device=dev)
        except Exception:                            # noqa: BLE001  (no usable device at all: vote 0 from the host if the backend lets us)
            flag = torch.tensor([0.0])
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        return float(flag.item()) >= 1.0

    def __call__(self):
        import torch
        if self.comm is None:
            self.s.reduce_stats_async(self._stats.data_ptr())
            torch.cuda.current_stream().synchronize()          # (the solver's stream is the current one in bench.py; harmless otherwise)
            self.s.synchronize()
            return allreduce_stats(self._stats, self.dist, self.group, total_batch=self.total)
        return torch.from_numpy(self.s.allreduce_stats(self.comm, self.world, self.rank, self.total))

    def close(self):
        if self.comm:
            import tinympc_amd as tm
            tm.rccl_comm_destroy(self.comm)
            self.comm = None


def init_process_group(local_rank):
    """One rank per GPU over RCCL ("nccl" is RCCL on ROCm).  Returns (torch.distributed, device index of this rank).
    TINYMPC_BENCH_SHARE_GPU=1 is the smoke mode of the drivers: the ranks share this box's devices round-robin and talk over
    gloo (RCCL refuses two ranks on one device), the statistics exchange takes its torch.distributed form."""
    import os
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("TINYMPC_BENCH_SHARE_GPU"):
        device = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(device)
        os.environ["TINYMPC_EXCHANGE"] = "torch"
        dist.init_process_group("gloo")
    else:
        device = local_rank
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    return dist, device
