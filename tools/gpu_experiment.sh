#!/bin/bash
# scratch stage of tools/gpu_stage.sh ("exp"): whatever is being debugged / measured at the moment
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_adaptive.py "tests/test_gpu_parity.py::test_hip_matches_reference_golden" -m gpu -q -x > $O/adaptive_pytest.txt 2>&1; tail -25 $O/adaptive_pytest.txt | cut -c1-300
