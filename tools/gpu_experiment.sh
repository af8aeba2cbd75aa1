#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 600 python -m pytest tests/test_gpu_hetero.py tests/test_gpu_fused_variants.py tests/test_gpu_repack.py -m gpu -q > $O/pytest_hetero.txt 2>&1; tail -3 $O/pytest_hetero.txt
timeout 600 python tools/fuzz_closed_loop.py 300 63000 > $O/fuzz_closed_loop.txt 2>&1; tail -2 $O/fuzz_closed_loop.txt
timeout 600 python tools/hetero_bench.py 2> $O/hetero_bench.err | tee $O/hetero_bench.txt | head -3
