#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): the kernel experiment of the moment goes here
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_variants.py -x -q -m gpu 2>&1 | tail -8
