#!/bin/bash
# GPU: the config-5 sweep of round 6 -- clean run + parity, per-instance-data parity on the tile cells, one SQ-counter pass
O=gpurun_out/r06i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python tools/sweep_bench.py --out $O/sweep.json --parity $O/sweep_parity.md > $O/sweep.md 2> $O/sweep.err
echo "sweep rc=$?"
timeout 900 python tools/sweep_bench.py --hetero $O/sweep_parity_hetero.md > $O/sweep_hetero.md 2> $O/sweep_hetero.err
echo "hetero rc=$?"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/sq -o sweep -- python tools/sweep_bench.py --out $O/sq/sweep_pmc.json > $O/sq_sweep.md 2> $O/sq_sweep.err
echo "pmc rc=$?"
python tools/sweep_counters.py $O/sq > $O/sweep_counters.json 2> $O/sweep_counters.err
rm -f $O/sq/*kernel_trace.csv
ls -la $O $O/sq | head -30
du -sh $O
tail -3 $O/sweep.md; tail -3 $O/sweep_hetero.md; tail -2 $O/sweep_parity.md; tail -2 $O/sweep_parity_hetero.md
