#!/bin/bash
set +e
O=gpurun_out/call4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -12 $O/pytest.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 900 $O/bench_default.json; echo; tail -3 $O/bench_default.err
timeout 300 python bench.py --steps-per-launch 1 --no-cpu-baseline > $O/bench_perstep.json 2> $O/bench_perstep.err; python -c "
import json; d=json.load(open('$O/bench_perstep.json')); print('per-step', d['value'], d['roofline']['frac'], d['roofline_fp64']['frac'], d['kernel_ms'])"
TINYMPC_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 100 > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; tail -c 300 $O/bench_torchrun1.json; echo; tail -3 $O/bench_torchrun1.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace -o hover -- python $R/bench.py --no-cpu-baseline > $R/$O/rocprof_trace_bench.json 2> $R/$O/rocprof_trace.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_fetch -o hover -- python $R/bench.py --no-cpu-baseline > /dev/null 2> $R/$O/rocprof_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_write -o hover -- python $R/bench.py --no-cpu-baseline > /dev/null 2> $R/$O/rocprof_write.err
cd $R; head -4 $O/prof_trace/hover_kernel_stats.csv | cut -c1-200
