#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp")
O=$1; mkdir -p $O
timeout 300 python tools/adaptive_bench.py 2>&1 | tail -1 | tee $O/adaptive_bench.txt
timeout 900 python -m pytest tests/test_gpu_adaptive.py tests/test_gpu_fuzz.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -3
