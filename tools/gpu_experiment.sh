#!/bin/bash
# stage "exp" of tools/gpu_stage.sh: whatever kernel experiment is being measured at the moment (this default: the cone step's cost)
O=$1; mkdir -p $O; export O
timeout 300 python tools/soc_iter_cost.py > $O/soc_iter_cost.txt 2>&1; cat $O/soc_iter_cost.txt
