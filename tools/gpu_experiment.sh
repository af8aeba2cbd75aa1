#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py tests/test_gpu_repack.py tests/test_gpu_phases.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | tail -5
timeout 300 python tools/soc_iter_cost.py
timeout 300 python tools/bench_configs.py config4 config4_state_cone config4_both_cones 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('@@CFG@@'):
        d = json.loads(ln[7:])
        for k, e in d.items(): print(k, round(e['ms'], 3), round(e['roofline']['frac'], 3))
"
