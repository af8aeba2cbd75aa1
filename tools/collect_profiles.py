#!/usr/bin/env python3
"""gpurun_out/final (tools/gpu_final.sh) -> profiles/<tag>_* and profiles/traffic.json.
    python tools/collect_profiles.py [tag = r01_final]"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01_final"


def find(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return hits[0] if hits else None


def pmc_per_launch(dirname, counter, kernel="admm_solve_kernel"):
    f = find(f"{dirname}/**/*counter_collection.csv")
    if not f:
        return None, 0
    tot, launches = 0.0, set()
    for r in csv.DictReader(open(f)):
        if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
            tot += float(r["Counter_Value"])
            launches.add(r.get("Dispatch_Id"))
    return (tot / max(len(launches), 1)), len(launches)


for name in ("pytest_gpu.txt", "smoke.txt", "bench_default.json", "bench_per_step.json", "bench_regimes.json", "bench_torchrun1.json",
             "configs_3_4.json", "phase_clocks.txt", "sweep_config5.json", "sweep_config5.md", "sweep_config5_split_solve.json",
             "sweep_config5_split_solve.md"):
    if os.path.exists(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, f"{TAG}_{name}"))
for mode in ("fused", "step"):
    f = find(f"prof_{mode}_trace/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(DST, f"{TAG}_{mode}_kernel_stats.csv"))
traffic = {}
for mode, key, steps in (("step", "per_step_launch", 1), ("fused", "fused_100_steps_launch", 100)):
    fetch, n1 = pmc_per_launch(f"prof_{mode}_fetch", "FETCH_SIZE")
    write, n2 = pmc_per_launch(f"prof_{mode}_write", "WRITE_SIZE")
    if fetch is None or write is None:
        continue
    alg = 10124 * 65536 * steps
    hbm = fetch * 1024 * 2 + write * 1024
    traffic[key] = {
        "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around bench.py ({mode} mode), tools/gpu_final.sh",
        "launches": n1, "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio_traffic_over_algorithmic": hbm / alg,
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of a coalesced stream)"
                + ("; the ADMM state stays in registers between the fused MPC steps: real traffic = one load + one store of the records" if steps > 1 else "")}
if traffic:
    old = json.load(open(os.path.join(DST, "traffic.json"))) if os.path.exists(os.path.join(DST, "traffic.json")) else {}
    old.update(traffic)
    json.dump(old, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print("collected into", DST, "traffic keys:", list(traffic))
