#!/bin/bash
O=$1; mkdir -p $O; export O
R=$PWD
for v in "" _abl_fwd_nogc _abl_bwd_now _abl_nopass; do
  echo "== lib$v"
  TINYMPC_AMD_LIB=$R/tinympc_amd/libtinympc_amd$v.so timeout 300 python tools/soc_iter_cost.py > $O/soc_iter_cost$v.txt 2>&1; cat $O/soc_iter_cost$v.txt
done
