#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): the kernel experiment of the moment goes here
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_jit.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/fuzz_parity.py 300 71001 2>&1 | tail -2
timeout 600 python tools/sweep_bench.py --reps 2 --cells "6,3,40;10,4,36" 2>&1 | tail -3
