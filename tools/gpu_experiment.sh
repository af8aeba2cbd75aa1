#!/bin/bash
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
TINYMPC_TEST_OPTS=tile_r=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_variants.py -m gpu -q -x -k "tile or sweep or golden or seeded or box" > $O/pytest_r2.txt 2>&1; tail -3 $O/pytest_r2.txt
timeout 900 python tools/tile_forms.py --reps 1 --cells "4,2,50;4,4,50;4,8,50;8,2,50;8,4,50;8,8,50;12,2,50;12,4,50;12,8,30;20,8,30;12,8,50;20,2,50;20,4,50;20,8,50" > $O/tile_forms_defer.md 2> $O/tile_forms_defer.err; grep -E "R=2.*dynamic|fastest" $O/tile_forms_defer.md; tail -3 $O/tile_forms_defer.err
