#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp")
O=$1; mkdir -p $O
timeout 900 python tools/fuzz_compat.py 400 5001 > $O/fuzz_compat.txt 2>&1; tail -2 $O/fuzz_compat.txt | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_dropin.py -m gpu -q 2>&1 | tail -2
