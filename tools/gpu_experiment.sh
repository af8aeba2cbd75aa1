#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py tests/test_gpu_fused_variants.py -m gpu -q -x > $O/pytest.txt 2>&1; grep -n "passed\|failed" $O/pytest.txt | tail -3
