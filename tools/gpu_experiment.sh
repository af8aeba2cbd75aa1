#!/bin/bash
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_jit.py -m gpu -q -x > $O/pytest_jit.txt 2>&1; tail -3 $O/pytest_jit.txt
