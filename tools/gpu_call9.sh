#!/bin/bash
set +e
O=gpurun_out/call9; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
timeout 300 python bench.py --no-cpu-baseline | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['admm_iters_per_s'], d['kernel_ms'], d['roofline_fp64']['frac'])"
