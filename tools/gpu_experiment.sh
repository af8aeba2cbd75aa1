#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py tests/test_gpu_repack.py tests/test_gpu_phases.py tests/test_gpu_regroup.py -m gpu -x -q --durations=12 > $O/pytest_subset.txt 2>&1; tail -22 $O/pytest_subset.txt
timeout 300 python tools/bench_configs.py config4 > $O/config4.json 2> $O/config4.err; python -c "
import json,sys
d=json.loads(open('$O/config4.json').read().split('@@CFG@@')[1])
e=d.get('config4',d)
print({k:e.get(k) for k in ('ms','ms_min','ms_max','plain_launch_ms','iters_per_s','step_regroup')}, e.get('roofline',{}).get('frac'))
"; tail -3 $O/config4.err
