#!/bin/bash
# Round-end evidence run on one MI355X: GPU tests, smoke, the default bench (+ per-step mode, + the RCCL path on one
# rank), rocprofv3 kernel statistics of the same commands, PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately),
# the other BASELINE configs and the phase clocks.  tools/collect_profiles.py turns gpurun_out/final into profiles/.
set +e
O=gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json; echo
timeout 300 python bench.py --steps-per-launch 1 --no-cpu-baseline > $O/bench_per_step.json 2> $O/bench_per_step.err
timeout 300 python bench.py --regimes --no-cpu-baseline > $O/bench_regimes.json 2> $O/bench_regimes.err
TINYMPC_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 100 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
cd /tmp
for mode in fused step; do
  extra=""; [ $mode = step ] && extra="--steps-per-launch 1 --warmup 0"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_${mode}_trace -o hover -- python $R/bench.py --no-cpu-baseline $extra > $R/$O/rocprof_${mode}_bench.json 2> $R/$O/rocprof_${mode}_trace.err
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_${mode}_fetch -o hover -- python $R/bench.py --no-cpu-baseline $extra > /dev/null 2> $R/$O/rocprof_${mode}_fetch.err
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_${mode}_write -o hover -- python $R/bench.py --no-cpu-baseline $extra > /dev/null 2> $R/$O/rocprof_${mode}_write.err
done
cd $R
timeout 600 python tools/config_bench.py $O/configs_3_4.json > /dev/null 2> $O/configs.err
# BASELINE config 5 (36-cell sweep), plain and as split solves; skipped with FINAL_QUICK=1 (about 25 s of GPU time each)
if [ -z "$FINAL_QUICK" ]; then
  timeout 300 python tools/sweep_bench.py --out $O/sweep_config5.json > $O/sweep_config5.md 2> $O/sweep.err
  TINYMPC_OPTS=repack_after=32 timeout 300 python tools/sweep_bench.py --out $O/sweep_config5_split_solve.json > $O/sweep_config5_split_solve.md 2>> $O/sweep.err
fi
TINYMPC_AMD_LIB=$R/tinympc_amd/libtinympc_amd_clk.so timeout 300 python tools/phase_clocks.py > $O/phase_clocks.txt 2>&1
find $O -name "*.csv" | head -20
