#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 300 python -m pytest tests/test_gpu_jit.py -m gpu -q -k "cone" > $O/pytest_jit.txt 2>&1; tail -3 $O/pytest_jit.txt
for r in "" 2; do
echo "== TILE_R=$r"
TILE_R=$r ONLY="input cone" timeout 600 python tools/tile_variants_bench.py > $O/tile_variants_r$r.md 2> $O/tile_variants_r$r.err; grep "cone" $O/tile_variants_r$r.md; tail -6 $O/tile_variants_r$r.err | grep tile
done
