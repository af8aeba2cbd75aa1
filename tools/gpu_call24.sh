#!/bin/bash
set +e
O=gpurun_out/call24; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
python tools/config_bench.py $O/cfg.json config3 | head -30
python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline_fp64']['frac'], d['regimes']['steady_state']['ms_per_launch'])"
