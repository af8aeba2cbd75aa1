#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp")
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_closed_loop" 2>&1 | tail -2
python bench.py --no-cpu-baseline --min-seconds 0.3 --steps 20 --warmup 5 > $O/b.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b.json'))['regimes']; print({k:(round(v['ms_per_launch'],4), round(v.get('hbm_frac',0),3)) for k,v in d.items()}, d['steady_state_lean']['first_knot_only'])"
