#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): the kernel experiment of the moment goes here
O=$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_group.py tests/test_gpu_sharding.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for e in multi_gpu_group multi_gpu_rccl; do [ -x examples/$e ] && (timeout 120 examples/$e | tail -2); done
