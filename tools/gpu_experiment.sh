#!/bin/bash
O=$1; mkdir -p $O
for ex in native torch; do
TINYMPC_EXCHANGE=$ex TINYMPC_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_$ex.json 2> $O/bench_$ex.err
python -c "
import json; d=json.load(open('$O/bench_$ex.json')); print('$ex', d['config']['stats_exchange'], '%.4g'%d['value'], d['timed_region']['ms'], d['admm_iters_per_solve'], d['solved_fraction'])"
done
TINYMPC_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 tools/config_bench.py $O/cfg_dist.json config3,config4 > /dev/null 2> $O/cfg_dist.err; python -c "
import json; c=json.load(open('$O/cfg_dist.json')); print({k:(v.get('seconds'), v.get('admm_iters_per_s')) for k,v in c.items()}); print(c['config4'])" | cut -c1-600
TINYMPC_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 tools/sweep_bench.py --batch 131072 --cells "12,4,10;8,4,30" --out $O/sweep_dist.json 2> $O/sweep_dist.err | tail -3
