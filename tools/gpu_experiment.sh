#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py tests/test_gpu_repack.py tests/test_gpu_phases.py -m gpu -q -x > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | tail -5
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --min-seconds 2 > $O/b.json 2>$O/b.err
python - <<'PY'
import json, os
d = json.load(open(os.environ["O"] + "/b.json"))
print("value %.4g roofline %.3f" % (d["value"], d["roofline"]["frac"]))
for k, v in d["regimes"].items():
    if isinstance(v, dict) and "hbm_frac" in v: print(k, "ms %.4f min %.4f hbm_frac %.3f" % (v["ms_per_launch"], v["ms_per_launch_min"], v["hbm_frac"]))
PY
