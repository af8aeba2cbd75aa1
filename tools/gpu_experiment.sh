#!/bin/bash
O=$1; mkdir -p $O
./tools/ubench/ubench_test_half_matvec 2>&1 | tee $O/half_matvec.txt
./tools/ubench/ubench_dpp_bankmask2 > $O/ubench_dpp_bankmask2.txt 2>&1
TINYMPC_TEST_OPTS=prefer_tile=1,tile_w=0,tile_dyn=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sweep_4_2 or sweep_cells or 4-2-50 or dims0" > $O/pytest_hr.txt 2>&1; tail -4 $O/pytest_hr.txt
TINYMPC_TEST_OPTS=prefer_tile=1,tile_w=0,tile_dyn=0 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sweep_4_2 or 4-2-50 or dims0" > $O/pytest_hr_static.txt 2>&1; tail -3 $O/pytest_hr_static.txt
timeout 900 python tools/tile_forms.py --reps 1 --cells "4,2,10;4,4,10;4,2,30;4,4,30;4,2,50;4,4,50" > $O/tile_forms_hr.md 2> $O/tile_forms_hr.err; grep -E "half|fastest|one-row" $O/tile_forms_hr.md; tail -3 $O/tile_forms_hr.err
