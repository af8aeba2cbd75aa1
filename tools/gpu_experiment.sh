#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python - <<'PY' 2>&1 | tee $O/config3_autocold.txt
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_configs
e = bench_configs.config3()
print("config3 ms %.3f plain %.3f auto %.3f frac %.3f" % (e["ms"], e["plain_launch_ms"], e["automatic_split_ms"], e["roofline"]["frac"]))
PY
