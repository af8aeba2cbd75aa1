#!/bin/bash
O=$1; mkdir -p $O; export O
FUZZ_DIAG=1 timeout 300 python - <<'P' > $O/fuzz_diag_61066.txt 2>&1
import sys
sys.path.insert(0, "tools")
import fuzz_closed_loop as f
print(f.trial(61066))
P
grep -n "instance 7" $O/fuzz_diag_61066.txt | head -30; tail -2 $O/fuzz_diag_61066.txt
