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
    assert abs(d["value"] - 2 * 8192 * 20 / (d["ms_per_step"] * 20 * 1e-3)) < 1e-5 * d["value"]      # (the line rounds to 6-7 digits)
    assert len(lines[0]) <= 4096


def test_bench_plain_process_spawns_its_own_ranks_weak_and_strong():
    """`python3 bench.py --gpus 2 ...` as a PLAIN process (no launcher in front of it) starts its own two
    ranks, emits exactly one JSON line, and carries the strong-scaling form of the metric (65 536-style total batch sharded
    over the ranks) beside the weak one; --scaling strong makes that form the headline."""
    import json
    import subprocess
    env = dict(os.environ, TINYMPC_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-regimes",
            "--min-seconds", "0.05", "--batch", "8192"]
    p = subprocess.run(base, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["launcher"] == "self-spawned torch.distributed.run"
    assert d["config"]["batch_per_gpu"] == 8192 and d["config"]["total_batch"] == 16384 and d["rccl_ranks"] == 2
    assert abs(d["admm_iters_per_solve"] - 34.55) < 1e-9
    ss = d["strong_scaling"]
    assert ss["total_batch"] == 8192 and ss["batch_this_rank"] == 4096
    assert abs(ss["value"] - 8192 * 20 / (ss["ms_per_step"] * 20 * 1e-3)) < 1e-5 * ss["value"]
    p = subprocess.run(base + ["--scaling", "strong"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["batch_per_gpu"] == 4096 and d["config"]["total_batch"] == 8192 and "strong_scaling" not in d
    assert abs(d["value"] - 8192 * 20 / (d["ms_per_step"] * 20 * 1e-3)) < 1e-5 * d["value"]
    assert abs(d["admm_iters_per_solve"] - 34.55) < 1e-9


@pytest.mark.parametrize("fault", ["1:exit", "1:hang", "0:exit"])
def test_bench_with_a_lost_rank_still_prints_one_json_line(fault):
    """the first real N > 1 run must not be able to end in silence.  One of two self-spawned ranks dies (or
    stalls for ever) in front of the rendezvous: within the deadline the plain process prints exactly ONE JSON line -- `error`,
    `value` null, the preflight of the node -- and returns non-zero; nothing is left running."""
    import json
    import subprocess
    import time
    env = dict(os.environ, TINYMPC_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", TINYMPC_BENCH_FAULT=fault)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-regimes",
                        "--min-seconds", "0.05", "--batch", "8192", "--dist-timeout", "20", "--run-timeout", "60"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert time.time() - t0 < 120
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["value"] is None and d["error"] and d["n_gpus"] == 2 and d["preflight"]["gpus_visible"] >= 1
    assert d["metric"].startswith("QP solves/sec") and d["steps"] == 20


def test_bench_line_carries_the_other_configs_and_honest_hbm_fields(tmp_path):
    """the 1-GPU run prints ONE compact line (<= 4 KiB of strict JSON: the contract's keys,
    `roofline`, `cpu_baseline`, a few numbers per regime and `configs` entry, the warm regime beyond the Infinity Cache as
    `roofline_hbm`) and writes the full record -- `configs` (BASELINE configs 3, 4 and six sweep cells, each with a roofline that
    recomputes from its own fields), `regimes` with hbm_frac from the bytes really moved next to the formula figure -- to the details
    file the line names."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    details = str(tmp_path / "bench_details.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--min-seconds", "0.3", "--cpu-seconds", "0.5",
                        "--details", details], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.count("\n") == 1 and p.stdout.startswith("{")      # stdout IS the one line

    def reject(name):
        raise ValueError(name)
    line = json.loads(p.stdout, parse_constant=reject)
    assert len(p.stdout) <= 4096 and "truncated" not in line
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["config"]["launcher"] == "plain process" and line["dtype"] == "f64"
    assert line["roofline"]["bound"] == "fp64-valu" and 0.5 < line["roofline"]["frac"] < 1.0 and line["roofline"]["avg_launch_ms"] > 0
    assert line["cpu_baseline"]["kind"] in ("reference", "port") and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    assert line["roofline_hbm"]["beyond_L3"] is True and line["roofline_hbm"]["batch"] >= 262144
    assert 0.2 < line["roofline_hbm"]["own_refs"]["hbm_frac"] < 1.0 and 0.2 < line["warm_regime"]["own_refs"]["hbm_frac"] < 1.0
    assert line["parity"] == {"entries_checked": 12, "mismatches": 0}
    for name, c in line["configs"].items():            # with the shipped plans the bench line shows first_call_ms next to ms; on a quiet box every entry is within 1.1x, the test allows a noisy one 1.3x
        assert c["first_call_ms"] <= 1.3 * c["ms"] + 0.2, (name, c)     # (+ 0.2 ms: the sub-millisecond entries on a shared box)
    assert all(line["configs"][k].get("plan") == "shipped" for k in ("config3", "config4", "sweep_4_2_10", "sweep_12_4_30"))
    c3 = line["configs"]["config3"]                               # an imported plan: the settled form on the first call, not a probe
    assert c3["planned_first_call_ms"] <= 1.25 * c3["ms"] or c3["planned_first_call_ms"] < 0.95 * c3["first_call_ms"], c3
    assert os.path.samefile(os.path.join(ROOT, line["details"]), details)
    d = json.load(open(details))
    assert abs(d["value"] / line["value"] - 1) < 1e-6
    assert d["roofline"]["traffic_source"] is None or "traffic.json" in d["roofline"]["traffic_source"]
    cf = d["configs"]
    assert set(cf) == set(line["configs"])
    assert set(cf) == {"config3", "config4", "config4_state_cone", "config4_both_cones", "sweep_4_2_10", "sweep_12_4_30", "sweep_4_2_50",
                       "sweep_12_8_30", "sweep_20_8_10", "sweep_20_8_50", "hetero_20_8_10", "tracking_12_8_30"}
    for name, e in cf.items():
        assert "error" not in e and "skipped" not in e, (name, e)
        r = e["roofline"]
        # every entry is as checkable as the headline: the fraction recomputes from the entry's own fields, `ms` is
        # the median of its repetitions, a sample of its own records agrees with the oracle, the reference was timed on the same records
        assert abs(r["frac"] - e["iters"] * r["flops_per_iter"] / (e["ms"] * 1e-3) / (r["peak"] * 1e12)) < 1e-12 + 1e-9 * r["frac"], name
        assert abs(e["solves_per_s"] - e["solves"] / (e["ms"] * 1e-3)) < 1e-6 * e["solves_per_s"]
        assert e["ms_min"] <= e["ms"] <= e["ms_max"] and e["timed_repetitions"] >= 3, name
        ps = e["parity_sample"]
        assert ps["iteration_count_mismatches"] == 0 and ps["iter_sum_gpu"] == ps["iter_sum_oracle"] and ps["instances"] >= 200, (name, ps)
        assert ps["max_rel_err_u0"] < 1e-5, (name, ps)                     # BASELINE's tolerance; the suite's own bar (1e-9) is in test_gpu_parity.py
        cb = e["cpu_baseline"]
        assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "QP solves/s", (name, cb)
    assert cf["config3"]["solves"] == 262144 and cf["config4"]["solves"] == 65536 * 90 and cf["sweep_20_8_50"]["kernel"] == "tile"
    assert cf["config4_both_cones"]["en_state_soc"] == 1 and cf["config4_both_cones"]["en_input_soc"] == 1
    rg = d["regimes"]
    big = rg["beyond_l3"]
    assert big["batch"] == 262144 and big["working_set_bytes"] > 3 * 256 * 2 ** 20          # several times the 256 MiB Infinity Cache
    for k in ("steady_state", "steady_state_per_instance_refs", "steady_state_no_primal_store", "steady_state_first_knot_store"):
        assert rg[k]["bytes_moved_per_solve"] <= rg[k]["algorithmic_bytes_per_solve"] and rg[k]["hbm_frac"] <= rg[k]["hbm_frac_formula"] + 1e-12
    assert rg["steady_state_per_instance_refs"]["bytes_moved_per_solve"] > rg["steady_state"]["bytes_moved_per_solve"]
    assert sum(rg["iters_per_step"]) == 882                            # the reference's 100-step hover episode, SURVEY.md 8(c)
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["value"] > 0
    assert d["gpu_phase_seconds"] > 0


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
