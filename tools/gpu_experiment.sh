#!/bin/bash
# step_regroup: parity of the cut launches, K sweep on config 4, the automatic switch on the other cone settings
O=$1; mkdir -p $O; export O
timeout 600 python -m pytest tests/test_gpu_regroup.py tests/test_gpu_fused_variants.py -m gpu -x -q > $O/pytest_regroup.txt 2>&1; tail -5 $O/pytest_regroup.txt
timeout 300 python tools/regroup_bench.py --cones input --ks 0,8,10,15,23,30,45,-1,0 > $O/regroup_input.md 2> $O/regroup_input.err; cat $O/regroup_input.md; tail -3 $O/regroup_input.err
timeout 300 python tools/regroup_bench.py --cones state --ks 0,-1,23 --reps 3 > $O/regroup_state.md 2> $O/regroup_state.err; cat $O/regroup_state.md; tail -3 $O/regroup_state.err
timeout 300 python tools/regroup_bench.py --cones both --ks 0,-1 --reps 3 > $O/regroup_both.md 2> $O/regroup_both.err; cat $O/regroup_both.md; tail -3 $O/regroup_both.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-regimes --min-seconds 2 > $O/bench_headline.json 2> $O/bench_headline.err; tail -c 700 $O/bench_headline.json
