#!/bin/bash
O=$1; mkdir -p $O
timeout 600 python tools/config_bench.py $O/cfg.json config4 > /dev/null 2>&1
python -c "
import json; c=json.load(open('$O/cfg.json'))['config4']
print({k:(round(v['seconds'],4), '%.3e'%v['admm_iters_per_s'], '%.3e'%v['solves_per_s']) for k,v in c.items() if k.startswith('steps')})"
timeout 300 python tools/soc_iter_cost.py 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_repack.py -m gpu -q -x 2>&1 | tail -2
