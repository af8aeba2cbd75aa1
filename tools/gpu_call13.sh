#!/bin/bash
set +e
O=gpurun_out/call13; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
