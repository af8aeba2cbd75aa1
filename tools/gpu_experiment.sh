#!/bin/bash
O=$1; mkdir -p $O; export O
for v in "" _abl_vp; do
  echo "--- lib$v" >> $O/soc_vp.txt
  TINYMPC_AMD_LIB=$PWD/tinympc_amd/libtinympc_amd$v.so timeout 200 python tools/soc_iter_cost.py >> $O/soc_vp.txt 2>&1
done
cat $O/soc_vp.txt
