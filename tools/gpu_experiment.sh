#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests -m gpu -q -x -k "cone or soc or rocket or closed or fused or steps" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python - > $O/cfg4.txt 2>&1 <<'PY'
import sys, json
sys.path.insert(0, "tools")
import bench_configs as bc
e = bc.config4()
print(json.dumps({k: e[k] for k in ("ms", "iters", "iters_per_s", "solved_fraction")}), e["roofline"]["frac"])
PY
cat $O/cfg4.txt
timeout 300 python tools/soc_iter_cost.py > $O/soc_iter_cost.txt 2>&1; cat $O/soc_iter_cost.txt
