#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): whatever is being debugged / measured at the moment
O=$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_group.py -m gpu -q -x > $O/group_pytest.txt 2>&1; grep -v "alt_rsmi\|^$" $O/group_pytest.txt | grep -i "warn\|error\|fail\|passed" | head -30
TINYMPC_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
tail -c 400 $O/bench_torchrun1.json; grep -i "librccl path" $O/bench_torchrun1.err
