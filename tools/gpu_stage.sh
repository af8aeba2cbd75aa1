#!/bin/bash
# One script for every intermediate GPU call of a round (the round-end evidence run is tools/gpu_final.sh):
#     gpurun --timeout 900 -- 'bash tools/gpu_stage.sh <stage> [<stage> ...]'
# Every stage writes under gpurun_out/<stage>/ and prints a short tail.  Stages:
#   tests        pytest -m gpu (whole suite) + smoke
#   tests_fast   the parity / sharding / native-group tests only
#   bench        bench.py with the driver's flags (--steps 20 --warmup 5) and the defaults
#   prof         rocprofv3 kernel stats + PMC traffic passes of both bench commands
#   configs      tools/config_bench.py (configs 3, 4, linear examples)
#   sweep        tools/sweep_bench.py (config 5)
#   adaptive     the adaptive-rho parity tests + a timing of the adaptive kernel variant
#   counters     rocprofv3 SQ / HBM counters of the sweep and cone kernels -> kernel_counters.md
#   cfgtraffic   FETCH_SIZE / WRITE_SIZE per solve of every `configs` entry of the bench line -> profiles/traffic.json[configs]
#   probes       warm-regime launch order table, cone iteration cost, half-row forms (round 4)
#   probes4      config 4 in stretches (step_regroup) per K and stream count, the tile kernel's cone / half-space variants, the second-stream probe
#   hetero       round 5: tools/hetero_bench.py (one-row HET variant + the tile kernel's per-instance form) and tools/dropin_latency.py
#   warm5        round 5: the warm regime beyond the Infinity Cache (batch 262 144), times + FETCH_SIZE / WRITE_SIZE of those launches
#   fuzz         the four differential fuzzers against the oracle
#   exp          whatever tools/gpu_experiment.sh holds (kernel experiments of the moment)
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for stage in "$@"; do
  O=gpurun_out/$stage; mkdir -p $O
  echo "=== stage $stage"
  case $stage in
    tests)
      timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
      python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt ;;
    tests_fast)
      timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharding.py tests/test_gpu_group.py tests/test_gpu_repack.py -m gpu -q > $O/pytest_gpu.txt 2>&1
      echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt ;;
    bench)
      # (round 5: stdout is ONE compact line; the full record of each run goes to its own details file)
      ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --details $R/$O/bench_driver_flags_details.json > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err ) 2> $O/bench_driver_flags.time
      wc -c $O/bench_driver_flags.json; cat $O/bench_driver_flags.json; grep real $O/bench_driver_flags.time
      timeout 600 python bench.py --no-cpu-baseline --no-configs --min-seconds 2 --details $R/$O/bench_default_details.json > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
      timeout 300 python bench.py --steps-per-launch 1 --no-cpu-baseline --no-regimes --no-configs --min-seconds 2 --details $R/$O/bench_per_step_details.json > $O/bench_per_step.json 2> $O/bench_per_step.err
      TINYMPC_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-regimes --min-seconds 2 > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
      tail -c 300 $O/bench_torchrun1.json; echo ;;
    prof)
      cd /tmp
      for mode in driver default step; do
        case $mode in driver) extra="--steps 20 --warmup 5" ;; default) extra="" ;; step) extra="--steps-per-launch 1 --warmup 0" ;; esac
        timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_${mode}_trace -o hover -- python $R/bench.py --no-cpu-baseline --no-regimes --no-configs --min-seconds 0.3 $extra > $R/$O/rocprof_${mode}_bench.json 2> $R/$O/rocprof_${mode}_trace.err
        timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_${mode}_fetch -o hover -- python $R/bench.py --no-cpu-baseline --no-regimes --no-configs --min-seconds 0.05 $extra > /dev/null 2> $R/$O/rocprof_${mode}_fetch.err
        timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_${mode}_write -o hover -- python $R/bench.py --no-cpu-baseline --no-regimes --no-configs --min-seconds 0.05 $extra > /dev/null 2> $R/$O/rocprof_${mode}_write.err
      done
      # SQ counters of the fused launches (issue utilisation of the headline kernel): their own pass, kernel-trace only
      timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/$O/prof_default_sq -o hover -- python $R/bench.py --no-cpu-baseline --no-regimes --no-configs --min-seconds 0.05 > /dev/null 2> $R/$O/rocprof_default_sq.err
      cd $R; find $O -name "*kernel_stats.csv" | head ;;
    configs)
      timeout 900 python tools/config_bench.py $O/configs_3_4.json > $O/configs.out 2> $O/configs.err; tail -c 800 $O/configs.out ;;
    sweep)
      timeout 1500 python tools/sweep_bench.py --reps 8 --out $O/sweep_config5.json --parity $O/sweep_parity.md > $O/sweep_config5.md 2> $O/sweep.err; tail -40 $O/sweep_config5.md; tail -3 $O/sweep_parity.md ;;
    adaptive)
      timeout 600 python -m pytest tests/test_gpu_adaptive.py -m gpu -q > $O/pytest_adaptive.txt 2>&1; tail -5 $O/pytest_adaptive.txt
      timeout 300 python tools/adaptive_bench.py > $O/adaptive_bench_run.txt 2>&1; tail -2 $O/adaptive_bench_run.txt ;;
    counters)
      # SQ / HBM counters of the kernels behind the other BASELINE configs (sweep cells on the one-row and tile kernels, the cone
      # kernel of config 4): one rocprofv3 pass per counter set, kernel trace only
      CELLS="12,4,10;4,2,10;12,4,30;4,2,30;12,4,50;4,2,50;20,8,10;20,8,50"
      cd /tmp
      for pass in "trace:" "sq:--pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "fetch:--pmc FETCH_SIZE" "write:--pmc WRITE_SIZE"; do
        name=${pass%%:*}; pmc=${pass#*:}
        timeout 600 rocprofv3 --kernel-trace $pmc --output-format csv -d $R/$O/$name -o sweep -- python $R/tools/sweep_bench.py --batch 65536 --reps 0 --cells "$CELLS" > $R/$O/${name}_sweep.out 2> $R/$O/${name}_sweep.err
        timeout 600 rocprofv3 --kernel-trace $pmc --output-format csv -d $R/$O/$name -o cfg4 -- python $R/tools/config_bench.py /dev/null config4 > $R/$O/${name}_cfg4.out 2> $R/$O/${name}_cfg4.err
      done
      cd $R; python tools/kernel_counters.py $O > $O/kernel_counters.md; cat $O/kernel_counters.md ;;
    bench3)
      # round 3: the driver's command (plain process, defaults of the new line: regimes, configs, CPU baseline last)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "rc=$?"; tail -c 600 $O/bench_driver_flags.json; echo; tail -5 $O/bench_driver_flags.err ;;
    cfgtraffic)
      # HBM-side bytes per solve of every `configs` entry: FETCH_SIZE / WRITE_SIZE passes (kernel trace only) of one entry per run
      cd /tmp
      for e in ${CFG_ENTRIES:-config3 config4 config4_state_cone config4_both_cones sweep_4_2_10 sweep_12_4_30 sweep_4_2_50 sweep_12_8_30 sweep_20_8_10 sweep_20_8_50 hetero_20_8_10 tracking_12_8_30}; do
        timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/${e}_fetch -o c -- python $R/tools/bench_configs.py $e > $R/$O/${e}_fetch.json 2> $R/$O/${e}_fetch.err
        timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/${e}_write -o c -- python $R/tools/bench_configs.py $e > $R/$O/${e}_write.json 2> $R/$O/${e}_write.err
      done
      cd $R; python tools/configs_traffic.py $O > $O/configs_traffic.json 2> $O/configs_traffic.err; tail -c 400 $O/configs_traffic.json ;;
    probes)
      # round 4: warm-regime launch order / store policy table, cone iteration cost, half-row forms
      timeout 600 python tools/warm_order_probe.py > $O/warm_order.md 2>&1; tail -20 $O/warm_order.md
      BATCHES=16384,32768,65536,131072 QUICK=1 timeout 600 python tools/warm_order_probe.py > $O/warm_order_batches.md 2>&1
      timeout 300 python tools/soc_iter_cost.py > $O/soc_iter_cost.txt 2>&1; cat $O/soc_iter_cost.txt
      timeout 600 python tools/half_rows_bench.py > $O/half_rows_bench.md 2>&1; cat $O/half_rows_bench.md ;;
    probes4)
      # round 4, second half: config 4 uncut / in stretches (one and two streams), the tile kernel's cone / half-space variants, other streams
      for st in 1 2; do timeout 300 python tools/regroup_bench.py --streams $st --cones input --ks 0,15,23,30,45,-1 > $O/regroup_input_s$st.md 2> $O/regroup_input_s$st.err; cat $O/regroup_input_s$st.md; done
      timeout 600 python tools/tile_variants_bench.py > $O/tile_variants_bench.md 2> $O/tile_variants_bench.err; cat $O/tile_variants_bench.md
      timeout 300 python tools/second_stream_probe.py > $O/second_stream_probe.md 2>&1 ;;
    hetero)
      # round 5: per-instance problem data on the one-row kernel and on the tile kernel's EXT form (wide / long shapes), single-solver drop-in latency
      timeout 600 python tools/hetero_bench.py > $O/hetero_bench.txt 2> $O/hetero_bench.err; cat $O/hetero_bench.txt
      timeout 300 python tools/dropin_latency.py > $O/dropin_latency.txt 2> $O/dropin_latency.err; cat $O/dropin_latency.txt ;;
    warm5)
      # round 5: the warm regime beyond the Infinity Cache (batch 262 144): time table + the PMC bytes of the very launches
      BATCHES=65536,262144 QUICK=1 timeout 600 python tools/warm_order_probe.py 2>&1 | grep -v "per step us" > $O/warm_beyond_l3.md; cat $O/warm_beyond_l3.md
      cd /tmp
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o warm -- python $R/tools/warm_traffic.py > $R/$O/warm_traffic_$c.out 2> $R/$O/warm_traffic_$c.err
      done
      cd $R; python tools/warm_traffic.py --collect $O > $O/warm_traffic.json 2> $O/warm_traffic.err; cat $O/warm_traffic.json ;;
    fuzz)
      # differential fuzzers against the oracle (new seeds every round); the closed-loop one draws the tile kernel's EXT forms since round 5
      timeout 1200 python tools/fuzz_parity.py ${FUZZ_N:-800} ${FUZZ_SEED:-50000} > $O/fuzz_parity.txt 2>&1; tail -2 $O/fuzz_parity.txt
      timeout 1200 python tools/fuzz_closed_loop.py $(( ${FUZZ_N:-800} / 2 )) $(( ${FUZZ_SEED:-50000} + 1000 )) > $O/fuzz_closed_loop.txt 2>&1; tail -2 $O/fuzz_closed_loop.txt
      timeout 600 python tools/fuzz_api_sequence.py 150 $(( ${FUZZ_SEED:-50000} + 2000 )) > $O/fuzz_api_sequence.txt 2>&1; tail -2 $O/fuzz_api_sequence.txt
      timeout 600 python tools/fuzz_parity.py 200 $(( ${FUZZ_SEED:-50000} + 3000 )) phases > $O/fuzz_phases.txt 2>&1; tail -2 $O/fuzz_phases.txt
      # round 6: the closed-loop trials once more on the coverage kernel (overlapping cones where the draw has a state cone, else force_general)
      timeout 900 python tools/fuzz_closed_loop.py $(( ${FUZZ_N:-800} / 2 )) $(( ${FUZZ_SEED:-50000} + 4000 )) cover > $O/fuzz_closed_loop_cover.txt 2>&1; tail -2 $O/fuzz_closed_loop_cover.txt ;;
    exp)
      bash tools/gpu_experiment.sh $O ;;
    *) echo "unknown stage $stage" ;;
  esac
done
