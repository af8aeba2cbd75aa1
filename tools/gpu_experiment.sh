#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): the kernel experiment of the moment goes here
O=$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -15
tests/dropin/_build/rho_driver | head -5 | cut -c1-200
