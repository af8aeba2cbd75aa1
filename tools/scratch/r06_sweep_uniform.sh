#!/bin/bash
O=gpurun_out/r06k
mkdir -p $O/sq
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/sq -o sweep -- python tools/sweep_bench.py --uniform 100 --reps 0 --out $O/sweep_pmc.json > $O/sq_sweep.md 2> $O/sq_sweep.err
echo "pmc rc=$?"
cp $O/sweep_pmc.json $O/sq/sweep_pmc.json
python tools/sweep_counters.py $O/sq 1 > $O/sweep_counters_uniform.json 2> $O/sweep_counters.err
rm -f $O/sq/*kernel_trace.csv $O/sq/*counter_collection.csv
tail -5 $O/sq_sweep.md; tail -3 $O/sweep_counters.err
