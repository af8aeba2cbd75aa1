#!/bin/bash
O=gpurun_out/r06j
mkdir -p $O/sq
C="12,2,10;12,4,10;4,8,10;8,4,10;20,2,10;12,8,10;4,2,30"
for o in "" "prefetch=0" "plan=0" "plan=0,prefetch=0"; do
  echo "== TINYMPC_OPTS=$o"
  TINYMPC_OPTS=$o timeout 300 python tools/sweep_bench.py --cells "$C" 2>&1 | grep "^| [0-9]" | cut -d'|' -f2-6,9
done
echo "== reps 5, default"
timeout 300 python tools/sweep_bench.py --cells "$C" --reps 5 2>&1 | grep "^| [0-9]" | cut -d'|' -f2-6,9
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/sq -o sweep -- python tools/sweep_bench.py --out $O/sweep_pmc.json > $O/sq_sweep.md 2> $O/sq_sweep.err
echo "pmc rc=$?"
cp $O/sweep_pmc.json $O/sq/sweep_pmc.json
python tools/sweep_counters.py $O/sq > $O/sweep_counters.json 2> $O/sweep_counters.err
rm -f $O/sq/*kernel_trace.csv $O/sq/*counter_collection.csv
head -c 600 $O/sweep_counters.json
