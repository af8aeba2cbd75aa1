#!/bin/bash
# round 5: reference windows / reset_duals on the shape's fast box form (EXT bit 0 with an LDS-offload set / dynamic slots)
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_hetero.py tests/test_gpu_jit.py -m gpu -q > $O/pytest_hetero.txt 2>&1; tail -3 $O/pytest_hetero.txt
timeout 900 python tools/fuzz_closed_loop.py 400 62000 > $O/fuzz_closed_loop.txt 2>&1; tail -3 $O/fuzz_closed_loop.txt
timeout 600 python tools/hetero_bench.py > $O/hetero_bench.txt 2> $O/hetero_bench.err; tail -8 $O/hetero_bench.txt
