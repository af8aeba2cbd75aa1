#!/bin/bash
# round 5: the first generation of waves out of phase (option stagger): config 3, then the hover regimes
O=$1; mkdir -p $O; export O
GRIDS=0 STAGGER=0,1,2,3,4,6,8 timeout 600 python tools/config3_probe.py 2>&1 | tee $O/config3_stagger.md
GRIDS=8 STAGGER=0,2,4 timeout 600 python tools/config3_probe.py 2>&1 | tee $O/config3_stagger_grid8.md
for s in 0 2 4; do
  timeout 300 python bench.py --no-cpu-baseline --no-configs --min-seconds 1 --opt stagger=$s --details $PWD/$O/bench_stagger$s.json > $O/bench_stagger$s.line 2> $O/bench_stagger$s.err
  python - <<P
import json
d = json.load(open("$O/bench_stagger$s.json"))
r = d["regimes"]
print("stagger $s: headline %.4g solves/s frac %.4f; cold %.4f ms; warm shared %.4f ms own %.4f ms; beyond L3 shared %.4f own %.4f ms" % (
    d["value"], d["roofline"]["frac"], r["cold"]["ms_per_launch"], r["steady_state"]["ms_per_launch"], r["steady_state_per_instance_refs"]["ms_per_launch"],
    r["beyond_l3"]["steady_state"]["ms_per_launch"], r["beyond_l3"]["steady_state_per_instance_refs"]["ms_per_launch"]))
P
done
