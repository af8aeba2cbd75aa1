#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 1500 python tools/fuzz_parity.py 2500 50001 > $O/fuzz_parity.txt 2>&1; tail -4 $O/fuzz_parity.txt
timeout 600 python tools/fuzz_closed_loop.py 300 9001 > $O/fuzz_closed_loop.txt 2>&1; tail -3 $O/fuzz_closed_loop.txt
