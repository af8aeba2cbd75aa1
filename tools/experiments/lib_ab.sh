#!/bin/bash
# GPU: A/B of two builds of the library on sweep cells, alternating (TINYMPC_AMD_LIB):  bash tools/experiments/lib_ab.sh <other .so> "<cells>"
C=${2:-"12,8,10;20,2,10;20,4,10;20,8,10;12,8,30;20,2,30;20,4,30;20,8,30;4,2,50;4,4,50;4,8,50;8,2,50;8,4,50;8,8,50;12,2,50;12,4,50;12,8,50;20,2,50;20,4,50;20,8,50"}
for rep in 1 2; do
for lib in "" "$1"; do
  echo "== lib=${lib:-default} rep $rep"
  TINYMPC_AMD_LIB=$lib TINYMPC_AMD_JIT_PREBUILT=0 timeout 900 python tools/sweep_bench.py --cells "$C" --reps 3 2>&1 | grep "^| [0-9]" | cut -d'|' -f2-6,9,11
done
done
