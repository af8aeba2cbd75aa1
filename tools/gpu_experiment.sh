#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | tail -5
timeout 600 python tools/sweep_bench.py --reps 8 --cells "12,2,50;20,4,50;12,8,50;20,8,50;4,8,50" > $O/sweep_cells.md 2>&1; grep "^| [0-9]" $O/sweep_cells.md
