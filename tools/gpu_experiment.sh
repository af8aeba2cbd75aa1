#!/bin/bash
O=$1; mkdir -p $O; export O
R=$PWD
for r in 1 2; do
for l in _gcz0 ""; do
echo "lib$l"
TINYMPC_AMD_LIB=$R/tinympc_amd/libtinympc_amd$l.so timeout 300 python tools/_c4mode.py | cut -c1-70
done
done
timeout 900 python -m pytest tests -m gpu -x -q -k "soc or cone or rocket" 2>&1 | tail -3
