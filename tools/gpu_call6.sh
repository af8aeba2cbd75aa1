#!/bin/bash
set +e
O=gpurun_out/call6; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -25 $O/pytest.txt
