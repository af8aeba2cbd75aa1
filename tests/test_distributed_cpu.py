"""World-size-2 CPU (gloo) test of the multi-GPU path: batch sharding and the one statistics all-reduce
(the RCCL collective of bench.py).  The per-rank "solve" is played by the plain-C
oracle here (test infrastructure only); on GPUs each rank runs libtinympc_amd.so on its shard."""
import os
import socket

import numpy as np
import pytest

from tinympc_amd.distributed import allreduce_stats, shard_bounds, shard_indices


def test_sharding_covers_everything_once():
    for total in (1, 7, 64, 65536, 1000003):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
            if total < 100:
                inter = sorted(i for r in range(world) for i in shard_indices(total, r, world, interleaved=True))
                assert inter == list(range(total))


def _suite(kind):
    import scenarios as sc
    # the recipes of BASELINE configs[2] (tracking, per-instance random references) and configs[4] (a sweep cell)
    return sc.tracking_random_suite(B=23, seed=4100) if kind == "config3" else sc.sweep_suite(4, 2, 10, B=17, max_iter=200)


def _shard_worker(rank, world, port, q, kind):
    """one rank of a sharded config driver (tools/config_bench.py / sweep_bench.py --gpus N): round-robin shard of the same
    seeded inputs, the solver played by the oracle, the one statistics exchange over gloo"""
    import torch
    import torch.distributed as dist
    import scenarios as sc
    from cpu_solvers import OracleSolver
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    suite = _suite(kind)
    B = suite["cases"]["x0"].shape[0]
    idx = np.array(shard_indices(B, rank, world, interleaved=True))
    out = sc.run_cases(OracleSolver, dict(suite, cases={k: v[idx] for k, v in suite["cases"].items()}))
    stats = torch.tensor([out["iter"].sum(), out["sol_solved"].sum(), len(idx), out["primal_residual_state"].max(),
                          out["primal_residual_input"].max(), out["dual_residual_state"].max(), out["dual_residual_input"].max(),
                          out["iter"].sum(), out["sol_solved"].sum(), 0.0], dtype=torch.float64)
    total = allreduce_stats(stats, dist, total_batch=B)
    q.put((rank, idx.tolist(), out["iter"].tolist(), out["x"].tolist(), total.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["config3", "config5"])
def test_round_robin_shards_of_the_config_drivers_equal_the_single_process_run(kind):
    import torch.multiprocessing as mp
    import scenarios as sc
    from cpu_solvers import OracleSolver
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = sc.run_cases(OracleSolver, _suite(kind))
    B = len(ref["iter"])
    seen = []
    for rank, idx, its, xs, total in res:
        seen += idx
        assert np.array_equal(np.array(its), ref["iter"][idx])                  # per instance: identical iteration counts
        assert np.array_equal(np.array(xs), ref["x"][idx])                      # ... and bit-identical trajectories
        expect = [ref["iter"].sum(), ref["sol_solved"].sum(), B, ref["primal_residual_state"].max(), ref["primal_residual_input"].max(),
                  ref["dual_residual_state"].max(), ref["dual_residual_input"].max(), ref["iter"].sum(), ref["sol_solved"].sum(), 0.0]
        assert np.array_equal(total, expect)                                    # the reduced 64-byte messages = the unsharded statistics
    assert sorted(seen) == list(range(B)) and len(np.unique(ref["iter"])) > 2    # every instance once; the batch does diverge


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import scenarios as sc
    from cpu_solvers import OracleSolver
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    suite = sc.tracking_random_suite(B=11, seed=77)                 # uneven shards (6 + 5): the job size is the sum of the shard sizes
    lo, hi = shard_bounds(11, rank, world)
    mine = dict(problem=suite["problem"], config=suite["config"], cases={k: v[lo:hi] for k, v in suite["cases"].items()})
    out = sc.run_cases(OracleSolver, mine)
    n = hi - lo
    stats = torch.tensor([out["iter"].sum(), out["sol_solved"].sum(), n, out["primal_residual_state"].max(),
                          out["primal_residual_input"].max(), out["dual_residual_state"].max(),
                          out["dual_residual_input"].max(), out["iter"].sum(), out["sol_solved"].sum(), 0.0],
                         dtype=torch.float64)
    total = allreduce_stats(stats, dist)
    q.put((rank, total.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_stats_allreduce_matches_single_process():
    import torch.multiprocessing as mp
    import scenarios as sc
    from cpu_solvers import OracleSolver
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = sc.run_cases(OracleSolver, sc.tracking_random_suite(B=11, seed=77))
    expect = [ref["iter"].sum(), ref["sol_solved"].sum(), 11, ref["primal_residual_state"].max(),
              ref["primal_residual_input"].max(), ref["dual_residual_state"].max(), ref["dual_residual_input"].max(),
              ref["iter"].sum(), ref["sol_solved"].sum(), 0.0]
    assert np.allclose(res[0], expect) and res[0] == res[1]


def test_reduce_table_takes_the_job_size_from_the_shard_column_and_keeps_a_nan_residual():
    """(1) the default job size must be exact for uneven shards (a 9th column, not shard size x world);
    (2) a NaN residual of a diverged shard survives the reduction (the native reduce_wire_table does the same)."""
    import torch
    from tinympc_amd.distributed import reduce_table
    rows = [[10, 2, 10, 2, 0.1, 0.2, 0.3, 0.4, 6], [7, 1, 7, 1, 0.5, float("nan"), 0.1, 0.2, 5]]
    out = reduce_table(torch.tensor(rows, dtype=torch.float64))
    assert out[0] == 17 and out[1] == 3 and out[2] == 11 and out[7] == 17 and out[8] == 3
    assert out[3] == 0.5 and torch.isnan(out[4]) and out[5] == 0.3 and out[6] == 0.4
    out = reduce_table(torch.tensor(rows, dtype=torch.float64)[:, :8], total_batch=11)
    assert out[2] == 11 and torch.isnan(out[4])
    with pytest.raises(ValueError):
        reduce_table(torch.tensor(rows, dtype=torch.float64)[:, :8])


def test_native_reduction_of_the_gathered_messages_agrees_with_reduce_table_nan_included():
    """the native group / RCCL exchange reduces the gathered messages on the host with the same result as
    distributed.reduce_table -- including a NaN residual of a diverged shard, which std::max(out, NaN) used to drop."""
    import torch
    import tinympc_amd as tm
    from tinympc_amd.distributed import reduce_table
    rng = np.random.default_rng(3)
    table = np.concatenate([rng.integers(0, 1000, (5, 4)).astype(float), rng.uniform(0, 1, (5, 4))], axis=1)
    for nan_at in (None, (3, 5), (0, 7), (4, 4)):
        t = table.copy()
        if nan_at:
            t[nan_at] = np.nan
        a = tm.reduce_stats_messages(t, 4242)
        b = reduce_table(torch.tensor(t), 4242).numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), (nan_at, a, b)
        assert a[2] == 4242 and (nan_at is None) == (not np.isnan(a).any())
