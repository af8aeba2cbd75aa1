#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 600 python -m pytest tests/test_gpu_repack.py -m gpu -q -x -k "reset" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
