#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 600 python -m pytest tests/test_gpu_jit.py -m gpu -q -x > $O/pytest_jit.txt 2>&1; tail -15 $O/pytest_jit.txt
ONLY="half" timeout 600 python tools/tile_variants_bench.py > $O/tile_variants_lin.md 2> $O/tile_variants_lin.err; grep "half" $O/tile_variants_lin.md; tail -12 $O/tile_variants_lin.err
