#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): the kernel experiment of the moment goes here
O=$1; mkdir -p $O
timeout 1500 python tools/tile_forms.py --reps 1 > $O/tile_forms.md 2> $O/tile_forms.err; grep -c fastest $O/tile_forms.md; tail -3 $O/tile_forms.err
