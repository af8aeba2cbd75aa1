#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): whatever is being debugged / measured at the moment
O=$1; mkdir -p $O
NCCL_DEBUG=WARN timeout 600 python -m pytest tests/test_gpu_group.py -m gpu -q -x > $O/group_pytest.txt 2>&1; grep -v "alt_rsmi\|^$" $O/group_pytest.txt | tail -30
# where does the time of the torch.distributed exchange go (one rank)?
TINYMPC_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 tools/dist_exchange_cost.py > $O/dist_cost.txt 2>&1
tail -12 $O/dist_cost.txt
