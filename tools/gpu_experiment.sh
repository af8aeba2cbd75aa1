#!/bin/bash
O=$1; mkdir -p $O; export O
R=$PWD
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench rc=$?"; tail -3 $O/bench_driver_flags.err
python - <<'PY'
import json, os
d = json.load(open(os.environ["O"] + "/bench_driver_flags.json"))
print("value %.4g  ms/step %.4f  roofline %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for k, v in d["regimes"].items():
    if isinstance(v, dict) and "hbm_frac" in v: print(k, "ms %.4f hbm_frac %.3f formula %.3f" % (v["ms_per_launch"], v["hbm_frac"], v["hbm_frac_formula"]))
for k, e in d["configs"].items():
    if "error" in e or "skipped" in e: print(k, e); continue
    ps, cb = e.get("parity_sample", {}), e.get("cpu_baseline", {})
    print("%-20s ms %.3f (min %.3f) frac %.3f | parity mism %s relerr %s | cpu %s %.3g/s on %s cores" % (k, e["ms"], e["ms_min"], e["roofline"]["frac"], ps.get("iteration_count_mismatches"), ps.get("max_rel_err_u0"), cb.get("kind"), cb.get("value", 0), cb.get("cores")))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "gpu_phase_s", d["gpu_phase_seconds"])
PY
timeout 900 python -m pytest tests/test_gpu_sharding.py -m gpu -q -x > $O/pytest_sharding.txt 2>&1; tail -5 $O/pytest_sharding.txt
