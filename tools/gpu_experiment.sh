#!/bin/bash
O=$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_repack.py tests/test_gpu_fused_variants.py -m gpu -q -x -k "rocket or soc or cone" > $O/pytest_soc.txt 2>&1; tail -3 $O/pytest_soc.txt
python - <<'PY' 2>&1 | tee $O/config4.txt
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_configs
e = bench_configs.config4()
print("config4 ms %.3f  it/s %.3e  frac %.3f" % (e["ms"], e["iters_per_s"], e["roofline"]["frac"]))
PY
