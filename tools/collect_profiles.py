#!/usr/bin/env python3
"""gpurun_out/<stage>/ (tools/gpu_stage.sh) -> profiles/<tag>_* and profiles/traffic.json.
    python tools/collect_profiles.py [tag = r02]

Stages read: tests (pytest + smoke), bench (driver flags / default / per-step / torchrun-1), prof (rocprofv3 kernel stats
and the FETCH_SIZE / WRITE_SIZE PMC passes of the driver-flags, default and per-step commands), configs, sweep."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"


def find(stage, pattern):
    hits = glob.glob(os.path.join(OUT, stage, pattern), recursive=True)
    return hits[0] if hits else None


def pmc_per_launch(dirname, counter, kernel="admm_solve_kernel"):
    f = find("prof", f"{dirname}/**/*counter_collection.csv")
    if not f:
        return None, 0
    tot, launches = 0.0, set()
    for r in csv.DictReader(open(f)):
        if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
            tot += float(r["Counter_Value"])
            launches.add(r.get("Dispatch_Id"))
    return (tot / max(len(launches), 1)), len(launches)


copies = [("tests", "pytest_gpu.txt", "pytest_gpu.txt"), ("tests", "smoke.txt", "smoke.txt"),
          ("bench", "bench_driver_flags.json", "bench_driver_flags.json"), ("bench", "bench_default.json", "bench_default.json"),
          ("bench", "bench_driver_flags_details.json", "bench_driver_flags_details.json"), ("bench", "bench_default_details.json", "bench_default_details.json"),
          ("bench", "bench_per_step_details.json", "bench_per_step_details.json"), ("bench", "bench_driver_flags.time", "bench_driver_flags.time"),
          ("hetero", "hetero_bench.txt", "hetero_bench.txt"), ("hetero", "dropin_latency.txt", "dropin_latency.txt"),
          ("warm5", "warm_beyond_l3.md", "warm_beyond_l3.md"), ("warm5", "warm_traffic.json", "warm_traffic_beyond_l3.json"),
          ("bench", "bench_per_step.json", "bench_per_step.json"), ("bench", "bench_torchrun1.json", "bench_torchrun1.json"),
          ("sweep", "sweep_config5.json", "sweep_config5.json"),
          ("sweep", "sweep_config5.md", "sweep_config5.md"), ("sweep", "sweep_parity.md", "sweep_parity.md"), ("adaptive", "pytest_adaptive.txt", "pytest_adaptive.txt"),
          ("adaptive", "adaptive_bench_run.txt", "adaptive_bench_run.txt"),
          ("counters", "kernel_counters.md", "kernel_counters_table.md"),      # (rNN_kernel_counters.md = this table + its reading)
          ("probes", "warm_order.md", "warm_launch_order.md"), ("probes", "warm_order_batches.md", "warm_launch_order_batches.md"),
          ("probes", "soc_iter_cost.txt", "soc_iter_cost.txt"), ("probes", "half_rows_bench.md", "half_rows_bench.md"),
          ("cfgtraffic", "configs_traffic.json", "configs_traffic.json"),
          ("probes4", "regroup_input_s1.md", "regroup_config4_one_stream.md"), ("probes4", "regroup_input_s2.md", "regroup_config4_two_streams.md"),
          ("probes4", "tile_variants_bench.md", "tile_variants_bench.md"), ("probes4", "second_stream_probe.md", "second_stream_probe.md")]
for stage, name, dst in copies:
    src = os.path.join(OUT, stage, name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(DST, f"{TAG}_{dst}"))
for mode in ("driver", "default", "step"):
    f = find("prof", f"prof_{mode}_trace/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(DST, f"{TAG}_{mode}_kernel_stats.csv"))
    b = os.path.join(OUT, "prof", f"rocprof_{mode}_bench.json")
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(DST, f"{TAG}_{mode}_bench_under_rocprof.json"))
# SQ counters of the fused launches: per-launch averages
f = find("prof", "prof_default_sq/**/*counter_collection.csv")
if f:
    acc, n = {}, {}
    for r in csv.DictReader(open(f)):
        if "admm_solve_kernel" in r.get("Kernel_Name", ""):
            k = r["Counter_Name"]
            acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
            n[k] = n.get(k, 0) + 1
    if acc:
        avg = {k: acc[k] / n[k] for k in acc}
        if avg.get("SQ_WAVE_CYCLES"):
            avg["valu_insts_per_wave_cycle"] = avg.get("SQ_INSTS_VALU", 0.0) / avg["SQ_WAVE_CYCLES"]
        json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES around `python bench.py` "
                             "(100 MPC steps fused per launch), per launch of admm_solve_kernel<12,4,10>", "per_launch": avg},
                  open(os.path.join(DST, f"{TAG}_sq_counters.json"), "w"), indent=1)
traffic = {}
for mode, key, what in (("step", "per_step_launch", "one launch per MPC step, the 100-step episode from cold"),
                        ("default", "fused_launch", "100 MPC steps fused per launch"),
                        ("driver", "fused_launch_driver_flags", "the launches of --steps 20 --warmup 5: one 5-step warm-up launch, then the 20 timed steps fused per launch")):
    fetch, n1 = pmc_per_launch(f"prof_{mode}_fetch", "FETCH_SIZE")
    write, n2 = pmc_per_launch(f"prof_{mode}_write", "WRITE_SIZE")
    if fetch is None or write is None:
        continue
    alg = 10124 * 65536
    hbm = fetch * 1024 * 2 + write * 1024
    traffic[key] = {
        "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around bench.py ({what}), tools/gpu_stage.sh prof",
        "launches": n1, "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio_traffic_over_algorithmic": hbm / alg,
        "round": f"{TAG} PMC run",
        "moved_bytes_model_per_launch": 65536 * (10124 - 2 * 1248),
        "note": "FETCH_SIZE x 2: the guide documents that correction for 16 B / lane streams; these loads are 8 B / lane -- the doubled "
                "figure lands within a few % of the bytes the launch form must read (7 x 1248 + 96 B per instance less the shared "
                "reference record), which is the evidence for using it here, not the guide.  algorithmic = "
                "bytes_warm = 10124 B per instance and LAUNCH: a launch loads and stores the records once however many MPC steps it "
                "fuses.  Below 1.0: the hover references are one shared record (share_ref) and a solve that converges at its first "
                "check does not store v|z again"}
if traffic:
    tp = os.path.join(DST, "traffic.json")
    try:
        old = json.load(open(tp))
    except Exception:                                    # noqa: BLE001
        old = {}
    if "configs" in old:                                 # (written by tools/configs_traffic.py)
        traffic["configs"] = old["configs"]
    json.dump(traffic, open(tp, "w"), indent=1)
print("collected into", DST, "traffic keys:", list(traffic))
