#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python tools/tile_variants_bench.py > $O/tile_variants_bench.md 2>$O/tile_variants.err; cat $O/tile_variants_bench.md; tail -3 $O/tile_variants.err
timeout 1500 python tools/sweep_bench.py --reps 8 --out $O/sweep_config5.json --parity $O/sweep_parity.md > $O/sweep_config5.md 2> $O/sweep.err; tail -5 $O/sweep_config5.md; cat $O/sweep_parity.md | tail -42
