#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 300 python -m pytest tests/test_gpu_regroup.py tests/test_gpu_repack.py -m gpu -q -x > $O/pytest_regroup.txt 2>&1; tail -3 $O/pytest_regroup.txt
SHAPES="12,4,10;6,3,10;8,4,30;4,2,30" BATCH=65536 timeout 300 python tools/tile_variants_bench.py > $O/onerow_variants_after.md 2> $O/onerow_variants_after.err; cat $O/onerow_variants_after.md
