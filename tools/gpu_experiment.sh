#!/bin/bash
O=$1; mkdir -p $O; export O
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/regroup_stats -o rg -- python $R/tools/regroup_bench.py --cones input --ks -1 --reps 3 > $R/$O/regroup_stats.out 2> $R/$O/regroup_stats.err
cd $R; find $O/regroup_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/regroup_kernel_stats.csv; cat $O/regroup_kernel_stats.csv | cut -c1-200; cat $O/regroup_stats.out
rm -rf $O/regroup_stats
