#!/bin/bash
set +e
O=gpurun_out/call22; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
python tools/config_bench.py $O/cfg.json config4 | grep -E "admm_iters_per_s|solves_per_s|iters_per_solve"
