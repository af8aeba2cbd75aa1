#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 900 python tools/tile_forms.py --reps 1 --cells "4,2,30;4,4,30;4,8,30;8,2,30;8,4,30" > $O/tile_forms_vpg6.md 2> $O/tile_forms_vpg6.err; grep "dynamic\|regs" $O/tile_forms_vpg6.md; tail -3 $O/tile_forms_vpg6.err
