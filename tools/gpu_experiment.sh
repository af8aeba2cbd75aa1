#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp")
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_group.py -m gpu -q -x 2>&1 | tail -8 | cut -c1-300
