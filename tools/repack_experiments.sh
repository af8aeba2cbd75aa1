#!/bin/bash
# Split solves on BASELINE config 3: per-stage kernel durations (rocprofv3 kernel trace) for a few stage schedules / grids.
#   gpurun -- bash tools/repack_experiments.sh
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for o in "repack_waves_per_cu=8" "repack_waves_per_cu=4" "repack_waves_per_cu=16" "repack_growth=3" "repack_growth=4"; do
  echo "== $o"
  rm -rf $R/gpurun_out/c3trace
  TINYMPC_OPTS=$o timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c3trace -o c3 -- python $R/tools/repack_trace.py 10 > $R/gpurun_out/c3trace.log 2>&1
  python $R/tools/trace_summary.py $R/gpurun_out/c3trace | tail -5
done
