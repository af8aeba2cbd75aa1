#!/bin/bash
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py tests/test_gpu_repack.py tests/test_gpu_hetero.py tests/test_gpu_adaptive.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt | cut -c1-300
for i in 1 2; do
  python bench.py --no-cpu-baseline --min-seconds 0.5 > $O/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b.json'))
print('fused100 %.4g solves/s  fp64 %.4f it/solve %.3f | cold step %.4f ms fp64 %.4f | steady %.4f lean %.4f' % (d['value'], d['roofline_fp64']['frac'], d['admm_iters_per_solve'], d['regimes']['cold']['ms_per_launch'], d['regimes']['cold']['fp64_frac'], d['regimes']['steady_state']['ms_per_launch'], d['regimes']['steady_state_lean']['ms_per_launch']))"
done
TINYMPC_OPTS=repack_after=0 python tools/config_bench.py $O/c3.json config3 > /dev/null 2>&1; python -c "
import json; c=json.load(open('$O/c3.json'))['config3']; print('config3 plain %.3f ms' % c['kernel_ms'])"
