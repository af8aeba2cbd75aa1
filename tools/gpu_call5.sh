#!/bin/bash
set +e
O=gpurun_out/call5; mkdir -p $O
for i in 1 2; do
TINYMPC_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 1 --steps 100 --warmup 100 --no-cpu-baseline 2> $O/bench_torchrun$i.err | tail -1 > $O/bench_torchrun$i.json
python -c "
import json; d=json.load(open('$O/bench_torchrun$i.json')); print('torchrun single-rank RCCL', d['value'], d['ms_per_step'], d['kernel_ms'])"
done
timeout 300 python bench.py --no-cpu-baseline | tail -1 > $O/bench_plain.json; python -c "
import json; d=json.load(open('$O/bench_plain.json')); print('plain', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['traffic'])"
