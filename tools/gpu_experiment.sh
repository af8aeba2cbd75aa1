#!/bin/bash
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sweep or 50" 2>&1 | tail -2
for opt in "prefer_tile=0" "prefer_tile=1"; do
  echo "== $opt"
  TINYMPC_OPTS=$opt timeout 600 python tools/sweep_bench.py --reps 5 --cells "12,4,50;4,2,50;8,4,50" 2>/dev/null | tail -3
done
