#!/bin/bash
O=$1; mkdir -p $O; export O
R=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py -m gpu -q -x > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | tail -5
timeout 300 python tools/soc_iter_cost.py > $O/soc_iter_cost.txt 2>&1; cat $O/soc_iter_cost.txt
TINYMPC_AMD_LIB=$R/tinympc_amd/libtinympc_amd_socclk.so timeout 300 python tools/soc_phase_clocks.py > $O/soc_phase_clocks.txt 2>&1; cat $O/soc_phase_clocks.txt
CHECK=1 TINYMPC_AMD_LIB=$R/tinympc_amd/libtinympc_amd_socclk.so timeout 300 python tools/soc_phase_clocks.py > $O/soc_phase_clocks_check1.txt 2>&1; cat $O/soc_phase_clocks_check1.txt
for v in "" _prim1 _prim2 _prim3; do
  echo "== lib$v"
  TINYMPC_AMD_LIB=$R/tinympc_amd/libtinympc_amd$v.so BATCHES=65536 QUICK=1 timeout 300 python tools/warm_order_probe.py > $O/warm_order$v.txt 2>&1; grep -v "per step" $O/warm_order$v.txt | tail -4; grep "per step" $O/warm_order$v.txt | sed -n 2p
done
