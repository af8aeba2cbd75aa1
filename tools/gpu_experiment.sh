#!/bin/bash
O=$1; mkdir -p $O
for g in 0 8 16; do
  python bench.py --no-cpu-baseline --no-regimes --grid-waves-per-cu $g --min-seconds 0.3 > $O/bench_g$g.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-regimes --grid-waves-per-cu $g --min-seconds 0.3 --steps 20 --warmup 5 > $O/bench20_g$g.json 2>/dev/null
  TINYMPC_OPTS=grid_waves_per_cu=$g python tools/config_bench.py $O/cfg_g$g.json config3,config4 > /dev/null 2>&1
  python - <<PY
import json
a=json.load(open("$O/bench_g$g.json")); b=json.load(open("$O/bench20_g$g.json")); c=json.load(open("$O/cfg_g$g.json"))
print("grid $g: fused100 %.4g solves/s (%.3f ms) | steps20 %.4g (%.3f ms) | config3 %.3f ms repack10 %.3f | config4 per-step %.4f s fused %.4f s" % (
  a["value"], a["timed_region"]["ms"]["median"], b["value"], b["timed_region"]["ms"]["median"], c["config3"]["kernel_ms"], c["config3"]["repack"]["repack_after=10"]["kernel_ms"],
  c["config4"]["steps_per_launch=1"]["seconds"], c["config4"]["steps_per_launch=90"]["seconds"]))
PY
done
