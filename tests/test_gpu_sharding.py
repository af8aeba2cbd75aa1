"""GPU: the multi-GPU split of SURVEY.md 8(e) on ONE device -- a divergent batch sharded round-robin
(shard_indices(interleaved=True)) over two handles must give, instance by instance, bit-identical results to the
unsharded batch, and the reduction of the shards' 64-byte statistics messages must equal the unsharded statistics."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scenarios as sc  # noqa: E402
from hip_runner import make_batch, IN_FIELDS, OUT_FIELDS  # noqa: E402
from tinympc_amd.distributed import WIRE_IDX, reduce_table, shard_indices  # noqa: E402

pytestmark = pytest.mark.gpu


def _solve(suite, idx):
    cases = {k: v[idx] for k, v in suite["cases"].items()}
    s = make_batch(dict(suite, cases=cases))
    s.set_x0(cases["x0"])
    for f in IN_FIELDS:
        if f in cases:
            s.set(f, cases[f])
    s.solve()
    out = {f: s.get(f) for f in OUT_FIELDS}
    st = s.status()
    out.update(iter=st["iter"], solved=st["solved"], status=st["status"],
               resid=np.stack([st[k] for k in ("primal_residual_state", "primal_residual_input", "dual_residual_state", "dual_residual_input")], axis=1))
    stats = s.reduce_stats()
    s.close()
    return out, stats


@pytest.mark.parametrize("world,interleaved", [(2, True), (2, False), (3, True)])
def test_sharded_batch_equals_unsharded(world, interleaved):
    B = 1021                                                     # not a multiple of the shard count or of a wave's 4 instances
    suite = sc.tracking_random_suite(B=B, seed=4242)             # config-3 recipe: iteration counts diverge
    full, full_stats = _solve(suite, np.arange(B))
    assert len(np.unique(full["iter"])) > 3
    table = []
    for r in range(world):
        idx = np.array(shard_indices(B, r, world, interleaved=interleaved))
        part, st = _solve(suite, idx)
        for k, v in part.items():
            assert np.array_equal(v, full[k][idx]), (k, r)       # bit-identical, instance by instance
        table.append(st[list(WIRE_IDX)])
    total = reduce_table(np.array(table), B).numpy()
    assert np.array_equal(total, full_stats), (total, full_stats)


def test_bench_control_flow_with_two_ranks_on_this_gpu():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank) with both ranks on this
    box's GPU over gloo (TINYMPC_BENCH_SHARE_GPU: RCCL refuses two ranks on one device): barriers, the max over ranks, the
    statistics exchange (its torch.distributed form) and rank 0's one JSON line -- whole-job counts, not rank 0's."""
    import json
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, TINYMPC_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--no-regimes", "--min-seconds", "0.05", "--batch", "8192"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                              # rank 0 speaks for the job
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak" and "SMOKE RUN" in d["data"]
    assert abs(d["admm_iters_per_solve"] - 34.55) < 1e-9               # 691 iterations over the first 20 steps, on BOTH ranks' instances
    assert d["config"]["batch_per_gpu"] == 8192 and d["config"]["stats_exchange"] == "torch.distributed"
    assert abs(d["value"] - 2 * 8192 * 20 / (d["ms_per_step"] * 20 * 1e-3)) < 1e-6 * d["value"]


def test_sharded_sweep_driver_counts_what_the_single_process_counts(tmp_path):
    """tools/sweep_bench.py (BASELINE configs[4]) under torch.distributed.run with two ranks sharing this GPU: the cell's
    batch is dealt round-robin to the ranks, every rank draws the same seeded inputs and keeps its own instances -- the
    job-wide iteration count and solved fraction must be the single-process ones."""
    import json
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tool = os.path.join(ROOT, "tools", "sweep_bench.py")
    common = ["--batch", "4099", "--cells", "12,4,10;20,4,10", "--reps", "0"]
    one, two = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    p = subprocess.run([sys.executable, tool, *common, "--out", one], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    env = dict(os.environ, TINYMPC_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), tool, *common, "--out", two], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    a, b = json.load(open(one)), json.load(open(two))
    assert [r["kernel"] for r in a] == ["regs", "tile"] and [r["n_gpus"] for r in b] == [2, 2]
    for ra, rb in zip(a, b):
        assert ra["iters_per_solve"] == rb["iters_per_solve"] and ra["solved_fraction"] == rb["solved_fraction"], (ra, rb)
