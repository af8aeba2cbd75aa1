#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): the kernel experiment of the moment goes here
O=$1; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 400 31001 adaptive 2>&1 | tail -3 | tee $O/fuzz_adaptive.txt
timeout 600 python tools/fuzz_parity.py 400 61001 2>&1 | tail -3 | tee $O/fuzz_parity.txt
