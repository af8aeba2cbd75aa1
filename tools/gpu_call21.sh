#!/bin/bash
set +e
O=gpurun_out/call21; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_phases.py -m gpu -q -x > $O/pytest_phases.txt 2>&1; tail -25 $O/pytest_phases.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
