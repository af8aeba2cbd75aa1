#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | tail -5
timeout 600 python tools/half_rows_bench.py > $O/half_rows_bench.md 2>&1; cat $O/half_rows_bench.md
