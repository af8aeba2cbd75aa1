#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 800 python -m pytest tests/test_gpu_repack.py tests/test_gpu_regroup.py -m gpu -x -q > $O/pytest_repack.txt 2>&1; tail -3 $O/pytest_repack.txt
CELLS="4,4,10;4,2,10;12,4,10;12,2,30"
for opt in "repack_sort=0" "repack_sort=-1" "repack_sort=1"; do
echo "== $opt"
TINYMPC_OPTS="$opt" timeout 600 python tools/sweep_bench.py --reps 10 --cells "$CELLS" > $O/sweep_$opt.md 2> $O/sweep_$opt.err; grep "^| [0-9]" $O/sweep_$opt.md | cut -d'|' -f2-8,11; tail -2 $O/sweep_$opt.err
done
timeout 300 python tools/regroup_bench.py --cones input --ks 0,-1 --reps 5 | tail -3
