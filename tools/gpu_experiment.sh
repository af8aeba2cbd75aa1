#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_repack.py -m gpu -q > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | tail -5
