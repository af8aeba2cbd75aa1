#!/bin/bash
O=$1; mkdir -p $O; export O
for v in abl1 abl2 abl3; do
  echo "--- $v" >> $O/soc_abl.txt
  TINYMPC_AMD_LIB=$PWD/tinympc_amd/libtinympc_amd_$v.so timeout 200 python tools/soc_iter_cost.py >> $O/soc_abl.txt 2>&1
done
cat $O/soc_abl.txt
