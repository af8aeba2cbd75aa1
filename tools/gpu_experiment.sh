#!/bin/bash
O=$1; mkdir -p $O; export O
for w in 0 2; do
echo "== TINYMPC_TILE_SOC_WAVES=$w"
TINYMPC_AMD_JIT_DEFINES="TINYMPC_TILE_SOC_WAVES=$w" timeout 600 python tools/tile_variants_bench.py > $O/tile_variants_w$w.md 2> $O/tile_variants_w$w.err; grep "cone" $O/tile_variants_w$w.md; tail -12 $O/tile_variants_w$w.err
done
