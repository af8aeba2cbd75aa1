#!/bin/bash
set +e
O=gpurun_out/call23; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
for i in 1 2; do python bench.py --no-cpu-baseline > $O/bench_$i.json 2>$O/bench_$i.err; python -c "
import json; d=json.load(open('$O/bench_$i.json')); print('fused', d['value'], d['roofline_fp64']['frac'], d['ms_per_step'])"; done
python bench.py --no-cpu-baseline --steps-per-launch 1 > $O/bench_step.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_step.json')); print('step', d['value'], d['kernel_ms'])"
python tools/config_bench.py $O/cfg.json config3,config4 | grep -E "admm_iters_per_s|kernel_ms"
