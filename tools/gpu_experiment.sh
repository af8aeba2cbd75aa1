#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): the kernel experiment of the moment goes here
O=$1; mkdir -p $O
TILE="4,2,50;4,4,50;4,8,50;8,2,50;8,4,50;8,8,50;12,2,50;12,4,50;12,8,10;20,2,10;20,4,10;20,8,10;12,8,30;20,2,30;20,4,30;20,8,30;12,8,50;20,2,50;20,4,50;20,8,50"
# parity of the tile kernel forms: default entries (R = 1 where tile_dims.txt lists one) and the R = 2 entries
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py -m gpu -q -x > $O/pytest_default.txt 2>&1; tail -3 $O/pytest_default.txt
TINYMPC_TEST_OPTS=tile_r=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py -m gpu -q -x -k "tile or sweep or golden or seeded or box" > $O/pytest_r2.txt 2>&1; tail -3 $O/pytest_r2.txt
timeout 600 python tools/sweep_bench.py --reps 2 --cells "$TILE" --out $O/sweep_tile_default.json > $O/sweep_tile_default.md 2> $O/sweep_default.err; cat $O/sweep_tile_default.md
TINYMPC_OPTS=tile_r=2 timeout 600 python tools/sweep_bench.py --reps 2 --cells "$TILE" --out $O/sweep_tile_r2.json > $O/sweep_tile_r2.md 2> $O/sweep_r2.err; cat $O/sweep_tile_r2.md
