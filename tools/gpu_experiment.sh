#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 600 python tools/tile_variants_bench.py > $O/tile_variants_bench.md 2> $O/tile_variants_bench.err; cat $O/tile_variants_bench.md
