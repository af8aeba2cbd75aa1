#!/usr/bin/env python3
"""HBM-side bytes per solve of the `configs` entries of bench.py's line, from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
runs, kernel trace only) of `python tools/bench_configs.py <entry>`:

    python tools/configs_traffic.py <dir>      # <dir>/<entry>_{fetch,write}/ (counter csv) + <dir>/<entry>_{fetch,write}.json (the entry's own output)
    -> merges {"configs": {entry: {...}}} into profiles/traffic.json

Per entry: the counter summed over every launch of the solver kernels (admm_solve_kernel / admm_tile_kernel) of the run, divided by the
solves the run launched (all of them cold solves / episodes of the same batch; config 3's first solves take other launch forms than its
settled ones -- an average).  FETCH_SIZE is doubled as profiles/traffic.json argues for the headline (TCC_EA0_RDREQ counts 128-B requests
at 64 B); WRITE_SIZE is taken as reported.  Infinity-Cache hits are inside both (the counters sit on the L2's fabric side)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
out = {}
for f in sorted(glob.glob(os.path.join(d, "*_fetch.json"))):
    name = os.path.basename(f)[:-len("_fetch.json")]
    try:
        line = [ln for ln in open(f).read().splitlines() if ln.startswith("@@CFG@@")][-1]
        e = json.loads(line[7:])[name]
        if "error" in e or "skipped" in e:
            continue
    except Exception:                                    # noqa: BLE001
        continue
    tot = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        t = 0.0
        for c in glob.glob(os.path.join(d, f"{name}_{kind}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(c)):
                if ("admm_solve_kernel" in r.get("Kernel_Name", "") or "admm_tile_kernel" in r.get("Kernel_Name", "")) and r.get("Counter_Name") == counter:
                    t += float(r["Counter_Value"])
        tot[kind] = t * 1024.0
    n = e.get("solves_launched") or e["solves"]
    n = n / e.get("mpc_steps_per_launch", 1)             # a fused episode moves its records once per LAUNCH: count instance-launches
    hbm = (2 * tot["fetch"] + tot["write"]) / n
    out[name] = {"hbm_bytes_per_instance_launch": hbm, "fetch_bytes_x2": 2 * tot["fetch"] / n, "write_bytes": tot["write"] / n,
                 "algorithmic_bytes_per_solve": e["hbm"]["algorithmic_bytes_per_solve"], "ratio_traffic_over_algorithmic": hbm / e["hbm"]["algorithmic_bytes_per_solve"],
                 "instance_launches_in_the_profiled_run": n, "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE around `python tools/bench_configs.py %s` (tools/gpu_stage.sh cfgtraffic)" % name,
                 "note": "per instance and launch (one cold solve; config 4: one fused 90-step EPISODE -- the records move once per launch, i.e. once per stretch where step_regroup cuts the episode); L2-fabric-side counters: Infinity-Cache hits included"}
tp = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tp)) if os.path.exists(tp) else {}
t.setdefault("configs", {}).update(out)
json.dump(t, open(tp, "w"), indent=1)
print(json.dumps(out, indent=1))
