#!/bin/bash
O=$1; mkdir -p $O
timeout 1200 python tools/fuzz_parity.py 400 1 adaptive > $O/fuzz_adaptive.txt 2>&1; tail -8 $O/fuzz_adaptive.txt | cut -c1-400
timeout 1200 python tools/fuzz_parity.py 600 50001 > $O/fuzz_parity.txt 2>&1; tail -4 $O/fuzz_parity.txt | cut -c1-400
timeout 900 python tools/fuzz_closed_loop.py 300 70001 > $O/fuzz_cl.txt 2>&1; tail -3 $O/fuzz_cl.txt | cut -c1-400
