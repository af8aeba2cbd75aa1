#!/bin/bash
set +e
O=gpurun_out/call8; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 1500 python tools/sweep_bench.py --batch 131072 --out $O/sweep.json > $O/sweep.md; grep -c regs $O/sweep.md
