#!/bin/bash
O=$1; mkdir -p $O
timeout 900 python tools/sweep_bench.py --reps 8 --out $O/sweep_config5_reps8.json 2> $O/sweep.err > $O/sweep_config5_reps8.md; grep -c tile $O/sweep_config5_reps8.md
