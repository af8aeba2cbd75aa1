#!/bin/bash
# GPU call 3: kernel v3 (fused slot update + pipelined bound reads)
set +e
O=gpurun_out/call3; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
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
run --dpp-mode 0 --steps-per-launch 100
run --dpp-mode 2 --steps-per-launch 100
run --dpp-mode 0
run --dpp-mode 2
run --dpp-mode 0 --steps-per-launch 100
run --dpp-mode 2 --steps-per-launch 100
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$O/prof_sq -o hover -- python $R/bench.py --steps 100 --warmup 0 --no-cpu-baseline --dpp-mode 2 > /dev/null 2> $R/$O/rocprof_sq.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --output-format csv -d $R/$O/prof_sq2 -o hover -- python $R/bench.py --steps 100 --warmup 0 --no-cpu-baseline --dpp-mode 2 > /dev/null 2> $R/$O/rocprof_sq2.err
cd $R; du -sh $O
