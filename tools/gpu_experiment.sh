#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 600 python tools/leak_check.py > $O/leak_check.txt 2>&1; tail -4 $O/leak_check.txt
timeout 600 python tools/big_batch_check.py > $O/big_batch.txt 2>&1; tail -4 $O/big_batch.txt
