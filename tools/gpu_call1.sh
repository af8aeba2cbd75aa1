#!/bin/bash
# First GPU call: probe, parity tests, bench variants, rocprofv3 kernel trace. Logs -> gpurun_out/
set +e
O=gpurun_out/call1; mkdir -p $O
export TMPDIR=/tmp
{ nproc; lscpu | grep -E "Model name|Socket|Thread|Core|MHz" ; rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8; ls /root/reference 2>&1 | head -3; } > $O/probe.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
timeout 300 python bench.py --steps 100 --warmup 10 > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
for v in "--dpp-mode 1" "--grid-waves-per-cu 8" "--grid-waves-per-cu 12" "--grid-waves-per-cu 16"; do
  n=$(echo $v | tr -d ' -'); timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline $v > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$n.json")); print("$v", "%.3e solves/s"%d["value"], "%.3e it/s"%d["admm_iters_per_s"], "hbm %.3f fp64 %.3f"%(d["roofline"]["frac"], d["roofline_fp64"]["frac"]), d["kernel_ms"])
except Exception as e: print("$v FAILED", e)
PY
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o hover -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err
cd $GRAFT_REPO_ROOT; find $O/prof -name "*stats*" | head; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
