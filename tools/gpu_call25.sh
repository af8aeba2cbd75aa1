#!/bin/bash
set +e
O=gpurun_out/call25; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 300 python tools/c4_soc_cost.py 2>&1 | tail -2
python tools/config_bench.py $O/cfg.json config4 | grep -E "admm_iters_per_s|solves_per_s"
