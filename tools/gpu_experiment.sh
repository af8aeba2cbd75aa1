#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): the kernel experiment of the moment goes here
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_jit.py -x -q -m gpu 2>&1 | tail -4
C="4,2,50;4,4,50;4,8,50;8,2,50;8,4,50;8,8,50;12,2,50;12,4,50;12,8,10;20,8,10;12,8,30;20,8,30;12,8,50;20,8,50"
timeout 900 python tools/sweep_bench.py --reps 3 --cells "$C" --out $O/tile_new.json 2>&1 | tail -14
