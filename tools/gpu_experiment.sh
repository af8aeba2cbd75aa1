#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_repack.py -m gpu -q > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | tail -5
timeout 600 python tools/sweep_bench.py --reps 8 --cells "4,2,50;4,4,50" > $O/sweep_hr50.md 2> $O/sweep_hr50.err; tail -3 $O/sweep_hr50.md
