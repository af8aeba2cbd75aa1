#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 800 python -m pytest tests/test_gpu_jit.py -m gpu -q --durations=5 > $O/pytest_jit.txt 2>&1; tail -30 $O/pytest_jit.txt
timeout 600 python tools/tile_variants_bench.py > $O/tile_variants_bench.md 2> $O/tile_variants_bench.err; cat $O/tile_variants_bench.md; tail -3 $O/tile_variants_bench.err
