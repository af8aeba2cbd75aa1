#!/bin/bash
# per-GPU shard sizes of BASELINE's 65 536-instance batch under STRONG scaling on 1 / 2 / 4 / 8 GPUs, each measured on this one GPU:
# what a rank of the N-GPU job runs (no data-path collective: the ranks do not interact until the 64-byte exchange)
O=$1; mkdir -p $O; export O
for B in 65536 32768 16384 8192; do
  for K in 20 100; do
    timeout 300 python bench.py --gpus 1 --batch $B --steps $K --warmup 5 --no-configs --no-cpu-baseline --no-regimes --min-seconds 1 > $O/shard_${B}_${K}.json 2> $O/shard_${B}_${K}.err
  done
done
python - <<'PY' | tee $O/strong_scaling_shards.md
import json, glob, os
O = os.environ.get("O", "gpurun_out/exp")
print("| steps | shard (instances per GPU) | = strong scaling on | QP solves/s per GPU | ms per step | FP64 frac | job = N x per-GPU | efficiency vs 1 GPU |")
print("|---|---|---|---|---|---|---|---|")
for K in (20, 100):
    base = None
    for B, n in ((65536, 1), (32768, 2), (16384, 4), (8192, 8)):
        d = json.load(open(f"{O}/shard_{B}_{K}.json"))
        if base is None: base = d["value"]
        print(f"| {K} | {B} | {n} GPU(s) | {d['value']:.4e} | {d['ms_per_step']:.4f} | {d['roofline_fp64']['frac']:.3f} | {n * d['value']:.4e} | {n * d['value'] / base / n:.3f} |")
PY
