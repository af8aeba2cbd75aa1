# how much does the PREFETCH form's warm time at 262 144 instances depend on the process / the allocation / the box?
mkdir -p gpurun_out/r06c
for rep in 1 2 3; do
  for env in "X=1" "TINYMPC_KPI_SEPARATE=1" "TINYMPC_KPI_SKEW=2101248"; do
    echo "## rep $rep env $env"
    env $env BATCHES=262144 timeout 600 python tools/prefetch_probe.py warm | grep "^| 262144"
  done
done
