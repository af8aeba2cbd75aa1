#!/bin/bash
# round 5: the gated v|z store -- parity, then A/B/C on one box: the code before the gate (old), the gate with a scalar branch around
# each store (the tree's library), the gate that only re-aims the stores at the pad (pad); every cell the v|z-in-its-record forms serve
O=$1; mkdir -p $O; export O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "gated_v_store or keep_v_in_its_record or fused_steps_on_a_tile_form or sweep_cells_full_batch" > $O/pytest_vgate.txt 2>&1; tail -4 $O/pytest_vgate.txt
CELLS="4,2,50;4,4,50;4,8,50;8,2,50;8,4,50;8,8,50;12,2,50;12,4,50;12,8,30;20,2,30;20,4,30;20,8,30;12,8,50;20,2,50;20,4,50;20,8,50"
R=$PWD
for v in old branch pad old branch pad; do
  lib=$R/tinympc_amd/ab/libtinympc_amd_$v.so; [ $v = branch ] && lib=$R/tinympc_amd/libtinympc_amd.so
  n=1; [ -f $O/sweep_${v}_1.json ] && n=2
  TINYMPC_AMD_LIB=$lib timeout 600 python tools/sweep_bench.py --reps 3 --cells "$CELLS" --out $O/sweep_${v}_$n.json > $O/sweep_${v}_$n.md 2> $O/sweep_${v}_$n.err
done
python - <<'P'
import json, os
O = os.environ["O"]
runs = {v: [json.load(open(f"{O}/sweep_{v}_{n}.json")) for n in (1, 2)] for v in ("old", "branch", "pad")}
print("| cell | old ms (two runs) | gate + branch | gate, pad only | branch / old | pad / old |")
print("|---|---|---|---|---|---|")
for i, x in enumerate(runs["old"][0]):
    m = {v: [r[i]["ms"] for r in runs[v]] for v in runs}
    best = {v: min(m[v]) for v in m}
    print(f"| ({x['nx']},{x['nu']},{x['N']}) | {m['old'][0]:.3f} {m['old'][1]:.3f} | {m['branch'][0]:.3f} {m['branch'][1]:.3f} | {m['pad'][0]:.3f} {m['pad'][1]:.3f} | {best['branch'] / best['old']:.3f} | {best['pad'] / best['old']:.3f} |")
P
