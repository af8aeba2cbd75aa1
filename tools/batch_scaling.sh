#!/bin/bash
# throughput of the default bench workload vs batch size (one GPU): where the chip saturates
for B in 256 1024 4096 16384 65536 262144 1048576; do
  python bench.py --no-cpu-baseline --batch $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('| $B | %.3e | %.3e | %.3f | %.3f |' % (d['value'], d['admm_iters_per_s'], d['ms_per_step'], d['roofline_fp64']['frac']))"
done
