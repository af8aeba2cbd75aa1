#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): the kernel experiment of the moment goes here
O=$1; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_t20.json 2> $O/bench_t20.err; tail -2 $O/bench_t20.err
python -c "
import json; d=json.loads(open('$O/bench_t20.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['mpc_steps_per_launch'], d['admm_iters_per_solve'], d['timed_region']['repeats'], d['regimes']['steady_state']['hbm_frac'])"
