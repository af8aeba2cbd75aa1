#!/bin/bash
# GPU: the config-5 sweep evidence of a round in one call (round 6):  gpurun --timeout 2700 -- 'bash tools/sweep_evidence.sh [out dir]'
#   sweep.md / sweep.json / sweep_parity.md      the 36 cells, 256 instances of every cell's full batch against the oracle
#   sweep_parity_hetero.md                       per-instance problem data on the 20 tile cells, every sampled instance against its own oracle
#   sweep_counters.json, sweep_counters_uniform.json   SQ_INSTS_VALU / kernel time per cell of the real run and of a run in which every
#                                                instance takes exactly 100 iterations (loop length, issue-slot use) -> tools/sweep_ceiling.py
O=${1:-gpurun_out/sweep_evidence}
mkdir -p $O/sq $O/squ
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python tools/sweep_bench.py --out $O/sweep.json --parity $O/sweep_parity.md > $O/sweep.md 2> $O/sweep.err; echo "sweep rc=$?"
timeout 900 python tools/sweep_bench.py --hetero $O/sweep_parity_hetero.md > $O/sweep_hetero.md 2> $O/sweep_hetero.err; echo "hetero rc=$?"
PMC="SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
timeout 900 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/sq -o sweep -- python tools/sweep_bench.py --out $O/sweep_pmc.json > $O/sq_sweep.md 2> $O/sq_sweep.err; echo "pmc rc=$?"
cp $O/sweep_pmc.json $O/sq/sweep_pmc.json
timeout 900 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/squ -o sweep -- python tools/sweep_bench.py --uniform 100 --reps 0 --out $O/sweep_pmc_uniform.json > $O/squ_sweep.md 2> $O/squ_sweep.err; echo "pmc uniform rc=$?"
cp $O/sweep_pmc_uniform.json $O/squ/sweep_pmc.json
python tools/sweep_counters.py $O/squ 1 > $O/sweep_counters_uniform.json 2> $O/sweep_counters.err
python tools/sweep_counters.py $O/sq 2 > $O/sweep_counters.json 2>> $O/sweep_counters.err
rm -f $O/sq/*kernel_trace.csv $O/sq/*counter_collection.csv $O/squ/*kernel_trace.csv $O/squ/*counter_collection.csv
tail -2 $O/sweep.md; tail -1 $O/sweep_parity.md; tail -1 $O/sweep_parity_hetero.md; tail -2 $O/sweep_counters.err
