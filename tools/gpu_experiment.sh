#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py tests/test_gpu_repack.py tests/test_gpu_jit.py tests/test_gpu_dropin.py tests/test_gpu_hetero.py -m gpu -q -x > $O/pytest_lin.txt 2>&1; tail -6 $O/pytest_lin.txt
ONLY="half" SHAPES="12,4,10;6,3,10;8,4,30" BATCH=65536 timeout 600 python tools/tile_variants_bench.py > $O/onerow_lin.md 2> $O/onerow_lin.err; grep half $O/onerow_lin.md; tail -4 $O/onerow_lin.err
