#!/bin/bash
C="12,8,10;20,2,10;20,4,10;20,8,10;12,8,30;20,2,30;20,4,30;20,8,30;12,8,50;20,2,50;20,4,50;20,8,50"
for rep in 1 2; do
for lib in "" "tinympc_amd/libtinympc_amd_w2c1.so"; do
  echo "== lib=${lib:-default(2 chains)} rep $rep"
  TINYMPC_AMD_LIB=$lib timeout 600 python tools/sweep_bench.py --cells "$C" --reps 3 2>&1 | grep "^| [0-9]" | cut -d'|' -f2-6,9,11
done
done
