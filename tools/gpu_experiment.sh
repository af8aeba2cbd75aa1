#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py tests/test_gpu_repack.py tests/test_gpu_fuzz.py tests/test_gpu_hetero.py tests/test_gpu_adaptive.py -m gpu -q -x > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | tail -5
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench rc=$?"
python - <<'PY'
import json, os
d = json.load(open(os.environ["O"] + "/bench_driver_flags.json"))
print("value %.4g  ms/step %.4f  roofline %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for k, v in d["regimes"].items():
    if isinstance(v, dict) and "hbm_frac" in v: print(k, "ms %.4f hbm_frac %.3f formula %.3f" % (v["ms_per_launch"], v["hbm_frac"], v["hbm_frac_formula"]))
    elif isinstance(v, dict): print(k, v)
for k, e in d["configs"].items():
    if "error" in e or "skipped" in e: print(k, e); continue
    print("%-20s ms %.3f (min %.3f) frac %.3f" % (k, e["ms"], e["ms_min"], e["roofline"]["frac"]))
PY
