#!/bin/bash
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "affine or half_row or clock" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
