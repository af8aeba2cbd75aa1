#!/bin/bash
O=$1; mkdir -p $O
R=$PWD
for lib in libtinympc_amd.so libtinympc_amd_w1.so; do
  for rep in 1 2; do
  TINYMPC_AMD_LIB=$R/tinympc_amd/$lib python bench.py --no-cpu-baseline --min-seconds 0.5 > $O/b_$lib.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_$lib.json'))
print('$lib', 'fused100 %.4g solves/s  fp64 %.4f | cold step %.4f ms fp64 %.4f | steady %.4f ms' % (d['value'], d['roofline_fp64']['frac'], d['regimes']['cold']['ms_per_launch'], d['regimes']['cold']['fp64_frac'], d['regimes']['steady_state']['ms_per_launch']))"
  done
done
