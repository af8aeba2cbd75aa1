#!/bin/bash
set +e
O=gpurun_out/call7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 1500 python tools/sweep_bench.py --batch 131072 --out $O/sweep.json | tee $O/sweep.md
