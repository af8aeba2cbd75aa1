"""CPU: bench.py's launch-shape arithmetic (the timed region must be EXACTLY --steps MPC steps in whole launches) and its
refusal to run without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_steps_per_launch_divides_the_timed_steps():
    import bench
    assert bench.steps_per_launch(100, 100) == 100          # the default: the whole reference episode in one launch
    assert bench.steps_per_launch(100, 10) == 100           # (the warm-up is cut into launches of its own)
    assert bench.steps_per_launch(20, 5) == 20              # the driver's flags: the 20 timed steps in ONE launch
    assert bench.steps_per_launch(7, 3) == 7
    assert bench.steps_per_launch(50, 0) == 50
    assert bench.steps_per_launch(1000, 1000) == 100        # never more than 100 steps per launch
    assert bench.steps_per_launch(300, 200) == 100
    assert bench.steps_per_launch(100, 100, 1) == 1
    assert bench.steps_per_launch(100, 10, 20) == 20
    assert bench.steps_per_launch(100, 10, 30) is None      # 30 does not divide the timed steps
    assert bench.steps_per_launch(5) == 5 and bench.steps_per_launch(101) == 1 and bench.steps_per_launch(202) == 2
    for steps in range(1, 230):
        T = bench.steps_per_launch(steps, 7)
        assert T and 1 <= T <= 100 and steps % T == 0


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no CPU fallback" in (p.stderr + p.stdout)
    assert p.stdout.strip() == ""                             # stdout is reserved for the one result line


def _reject_constant(name):
    raise ValueError("non-standard JSON constant %s in the bench line" % name)


def _canned_record():
    """a full record of a real run (round 4's own copy of the driver command: 33 KB as one line, which the driver dropped)"""
    import json
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench_driver_flags.json")))


def test_compact_line_is_short_strict_json_with_the_contract_keys():
    """the ONE stdout line stays below 4 KiB whatever the run measured, is strict JSON, and carries the
    contract's keys + roofline + cpu_baseline; the rest goes to the details file it names."""
    import json
    import bench
    out = _canned_record()
    out["regimes"]["beyond_l3"] = {"batch": 262144, "steady_state": {"hbm_frac": 0.58, "hbm_gbs": 4690.0, "ms_per_launch": 0.21},
                                   "steady_state_per_instance_refs": {"hbm_frac": 0.61, "hbm_gbs": 4930.0, "ms_per_launch": 0.25}}
    out["wall_seconds"] = 41.0
    line = bench.compact_line(out, "gpurun_out/bench_details.json")
    assert "\n" not in line and len(line.encode()) <= bench.COMPACT_LINE_LIMIT < 8192
    d = json.loads(line, parse_constant=_reject_constant)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert abs(d["value"] / out["value"] - 1) < 1e-6 and abs(d["ms_per_step"] / out["ms_per_step"] - 1) < 1e-5
    assert d["config"]["workload"].startswith("quadrotor_hovering") and "model" not in d["config"]
    rf = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf)
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["cores"] == 256 and d["cpu_baseline"]["value"] > 0
    assert d["roofline_hbm"]["beyond_L3"] is True and d["roofline_hbm"]["batch"] == 262144 and d["roofline_hbm"]["own_refs"]["hbm_frac"] == 0.61
    assert d["warm_regime"]["beyond_L3"] is False
    assert set(d["configs"]) == set(out["configs"]) and d["configs"]["config4"]["frac"] > 0.3
    assert d["parity"] == {"entries_checked": 10, "mismatches": 0}
    assert d["details"] == "gpurun_out/bench_details.json" and "truncated" not in d


def test_compact_line_survives_hostile_records():
    """non-finite numbers become null (strict JSON), a `configs` leg of any size cannot push the line over the limit, an error
    record keeps the envelope"""
    import json
    import bench
    out = _canned_record()
    out["value"] = float("nan")
    out["roofline"]["achieved"] = float("inf")
    out["configs"] = {"entry_%03d" % i: dict(out["configs"]["config3"]) for i in range(200)}
    out["configs"]["broken"] = {"error": "x" * 5000}
    line = bench.compact_line(out, None)
    assert len(line.encode()) <= bench.COMPACT_LINE_LIMIT
    d = json.loads(line, parse_constant=_reject_constant)
    assert d["value"] is None and d["roofline"]["achieved"] is None and d["truncated"] is True and "configs" not in d
    assert "roofline" in d and "cpu_baseline" in d and d["metric"].startswith("QP solves/sec")
    args = bench.parse_args([])
    e = json.loads(bench.error_line(args, 8, "boom " * 2000, partial_lines=["{" + "y" * 40000]), parse_constant=_reject_constant)
    assert e["value"] is None and e["n_gpus"] == 8 and len(json.dumps(e)) <= bench.COMPACT_LINE_LIMIT


def test_write_details_round_trips(tmp_path):
    import json
    import bench
    out = _canned_record()
    p = bench.write_details(out, str(tmp_path / "sub" / "bench_details.json"))
    assert p is not None
    back = json.load(open(tmp_path / "sub" / "bench_details.json"))
    assert back["configs"]["sweep_20_8_50"]["roofline"]["frac"] == out["configs"]["sweep_20_8_50"]["roofline"]["frac"]
