#!/bin/bash
O=$1; mkdir -p $O; export O
R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py tests/test_gpu_repack.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | tail -5
timeout 300 python tools/soc_iter_cost.py > $O/soc_iter_cost.txt 2>&1; cat $O/soc_iter_cost.txt
timeout 300 python tools/config_bench.py $O/configs_4.json config4 > $O/config4.out 2>&1; grep -A6 "steps_per_launch=90" $O/config4.out | head -8
for v in "" _prim0 _refnt; do
  echo "== lib$v"
  TINYMPC_AMD_LIB=$R/tinympc_amd/libtinympc_amd$v.so BATCHES=65536 QUICK=1 timeout 300 python tools/warm_order_probe.py > $O/warm_order$v.txt 2>&1; grep -v "per step" $O/warm_order$v.txt | tail -4; grep "per step" $O/warm_order$v.txt | sed -n 2p;  grep "per step" $O/warm_order$v.txt | sed -n 4p
done
