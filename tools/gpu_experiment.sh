#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): the kernel experiment of the moment goes here
O=$1; mkdir -p $O
TILE="4,2,50;4,4,50;4,8,50;8,2,50;8,4,50;8,8,50;12,2,50;12,4,50;12,8,10;20,2,10;20,4,10;20,8,10;12,8,30;20,2,30;20,4,30;20,8,30;12,8,50;20,2,50;20,4,50;20,8,50"
N30="4,2,30;4,4,30;4,8,30;8,2,30;8,4,30;8,8,30;12,2,30;12,4,30"
T="tests/test_gpu_parity.py tests/test_gpu_fused_variants.py"
timeout 900 python -m pytest $T -m gpu -q -x > $O/pytest_default.txt 2>&1; tail -3 $O/pytest_default.txt
TINYMPC_TEST_OPTS=tile_dyn=1 timeout 900 python -m pytest $T -m gpu -q -x > $O/pytest_dyn.txt 2>&1; tail -3 $O/pytest_dyn.txt
TINYMPC_TEST_OPTS=tile_dyn=1,prefer_tile=1 timeout 900 python -m pytest $T -m gpu -q -x -k "sweep or golden or seeded" > $O/pytest_dyn_prefer.txt 2>&1; tail -3 $O/pytest_dyn_prefer.txt
TINYMPC_TEST_OPTS=tile_r=2,tile_dyn=1 timeout 900 python -m pytest $T -m gpu -q -x -k "tile or sweep or golden or seeded or box" > $O/pytest_r2.txt 2>&1; tail -3 $O/pytest_r2.txt
timeout 600 python tools/sweep_bench.py --reps 2 --cells "$TILE" --out $O/sweep_tile_default.json > $O/sweep_tile_default.md 2> $O/sweep_default.err; cat $O/sweep_tile_default.md
TINYMPC_OPTS=tile_dyn=0 timeout 600 python tools/sweep_bench.py --reps 2 --cells "$TILE" --out $O/sweep_tile_static.json > $O/sweep_tile_static.md 2> $O/sweep_static.err; cat $O/sweep_tile_static.md
TINYMPC_OPTS=tile_r=2 timeout 600 python tools/sweep_bench.py --reps 2 --cells "$TILE" --out $O/sweep_tile_r2.json > $O/sweep_tile_r2.md 2> $O/sweep_r2.err; cat $O/sweep_tile_r2.md
TINYMPC_OPTS=prefer_tile=1 timeout 600 python tools/sweep_bench.py --reps 2 --cells "$N30" --out $O/sweep_n30_tile.json > $O/sweep_n30_tile.md 2> $O/sweep_n30.err; cat $O/sweep_n30_tile.md
timeout 600 python tools/sweep_bench.py --reps 4 --cells "$N30" --out $O/sweep_n30_regs.json > $O/sweep_n30_regs.md 2> $O/sweep_n30r.err; cat $O/sweep_n30_regs.md
