# HBM-side bytes of the warm launches at 262 144 instances: plain form against the PREFETCH form (separate PMC passes)
R=$PWD; O=gpurun_out/r06c; mkdir -p $O
rocm-smi --showserial | grep Serial
cd /tmp; export TMPDIR=/tmp
for pf in 0 -1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/$O/pmc_$c
    WARM_OPTS=prefetch=$pf timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o warm -- python $R/tools/warm_traffic.py > $R/$O/warm_traffic_$c.out 2> $R/$O/warm_traffic_$c.err
  done
  (cd $R; echo "## prefetch=$pf"; python tools/warm_traffic.py --collect $O)
done
