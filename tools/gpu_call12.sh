#!/bin/bash
set +e
O=gpurun_out/call12; mkdir -p $O
timeout 1500 python tools/sweep_bench.py --batch 131072 --out $O/sweep.json | tee $O/sweep.md
