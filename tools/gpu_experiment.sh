#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_steps_on_a_tile" > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error\|assert" $O/pytest.txt | tail -8
