mkdir -p gpurun_out/r06c
rocm-smi --showserial | grep Serial
for lib in "" fls2; do
  echo "## lib ${lib:-product}"
  if [ -n "$lib" ]; then export TINYMPC_AMD_LIB=$PWD/tinympc_amd/libtinympc_amd_$lib.so; else unset TINYMPC_AMD_LIB; fi
  BATCHES=262144,65536 OPTS="prefetch_static=0;prefetch_static=100" timeout 600 python tools/prefetch_probe.py $WHAT | grep "^| " | grep -v "batch"
done
