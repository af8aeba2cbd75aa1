#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 300 python tools/fuzz_parity.py 400 8101 > $O/fuzz_parity_c.txt 2>&1; tail -2 $O/fuzz_parity_c.txt
timeout 200 python tools/fuzz_closed_loop.py 60 8102 > $O/fuzz_cl_c.txt 2>&1; tail -1 $O/fuzz_cl_c.txt
timeout 200 python tools/fuzz_compat.py 40 8103 > $O/fuzz_compat_c.txt 2>&1; tail -1 $O/fuzz_compat_c.txt
