mkdir -p gpurun_out/r06c
for env in "TINYMPC_KPI_SEPARATE=1" "X=1"; do
for o in "prefetch=0" "prefetch=-1"; do
env $env python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --min-seconds 1 --opt $o --details gpurun_out/r06c/bd.json > gpurun_out/r06c/b.json 2>/dev/null
python - "$env $o" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r06c/bd.json'))
r=d['regimes']; b=r['beyond_l3']
print(sys.argv[1], 'inside', round(r['steady_state']['ms_per_launch'],4), round(r['steady_state_per_instance_refs']['ms_per_launch'],4), 'beyond', round(b['steady_state']['ms_per_launch'],4), round(b['steady_state_per_instance_refs']['ms_per_launch'],4))
PY
done
done
