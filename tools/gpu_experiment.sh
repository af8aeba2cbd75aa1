#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tile_kernel_matches_oracle" > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | tail -5
