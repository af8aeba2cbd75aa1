#!/bin/bash
O=$1; mkdir -p $O; export O
R=$PWD
for v in "" _socpd3 _socpd4 ""; do
  echo "== lib$v"
  TINYMPC_AMD_LIB=$R/tinympc_amd/libtinympc_amd$v.so CHECK=1 timeout 300 python tools/soc_iter_cost.py 2>&1 | grep -v "^box"
done
