#!/bin/bash
# GPU call 2: v2 kernel -- parity tests, bench matrix (dpp_mode x steps_per_launch x grid), rocprof trace + PMC
set +e
O=gpurun_out/call2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -8 $O/pytest.txt
timeout 300 python bench.py --steps 100 --warmup 10 > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
run() { n=$(echo "$*" | tr -d ' -'); timeout 200 python bench.py --steps 100 --warmup 0 --no-cpu-baseline $* > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$n.json")); print("%-50s"%"$*", "%.3e solves/s"%d["value"], "%.3e it/s"%d["admm_iters_per_s"], "hbm %.3f fp64 %.3f"%(d["roofline"]["frac"], d["roofline_fp64"]["frac"]), {k:round(v,3) for k,v in d["kernel_ms"].items()})
except Exception as e: print("$* FAILED", e)
PY
}
run --dpp-mode 0
run --dpp-mode 2
run --dpp-mode 0 --grid-waves-per-cu 8
run --dpp-mode 2 --grid-waves-per-cu 8
run --dpp-mode 0 --steps-per-launch 10
run --dpp-mode 0 --steps-per-launch 100
run --dpp-mode 2 --steps-per-launch 100
run --dpp-mode 0 --steps-per-launch 100 --grid-waves-per-cu 8
run --dpp-mode 2 --steps-per-launch 100 --grid-waves-per-cu 8
run --dpp-mode 0 --steps-per-launch 100 --batch 262144
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace -o hover -- python $R/bench.py --steps 100 --warmup 0 --no-cpu-baseline > $R/$O/rocprof_trace_bench.json 2> $R/$O/rocprof_trace.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_fetch -o hover -- python $R/bench.py --steps 100 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/$O/rocprof_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_write -o hover -- python $R/bench.py --steps 100 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/$O/rocprof_write.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$O/prof_sq -o hover -- python $R/bench.py --steps 100 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/$O/rocprof_sq.err
cd $R; find $O -name "*.csv" | head -20; du -sh $O
