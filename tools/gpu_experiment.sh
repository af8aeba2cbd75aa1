#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py tests/test_gpu_repack.py tests/test_gpu_phases.py -m gpu -q -x > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | tail -5
timeout 300 python tools/soc_iter_cost.py > $O/soc_iter_cost.txt 2>&1; cat $O/soc_iter_cost.txt
timeout 300 python tools/config_bench.py $O/configs_4.json config4 > $O/config4.out 2>&1; tail -c 600 $O/config4.out
timeout 600 python tools/warm_order_probe.py > $O/warm_order.txt 2>&1; cat $O/warm_order.txt
