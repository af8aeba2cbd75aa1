#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python tools/tile_forms.py --reps 1 --cells "12,2,50;4,8,50;8,4,50" > $O/tile_forms_qxr3.md 2> $O/tile_forms_qxr3.err; grep "LM=2[23] dynamic\|LM=54 dynamic" $O/tile_forms_qxr3.md; tail -3 $O/tile_forms_qxr3.err
