mkdir -p gpurun_out/r06c
for o in "prefetch=0" "prefetch=-1"; do
python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --min-seconds 1 --opt $o --details gpurun_out/r06c/bd_$o.json > gpurun_out/r06c/b_$o.json 2>/dev/null
python - "$o" <<'PY'
import json,sys
o=sys.argv[1]
d=json.load(open('gpurun_out/r06c/bd_%s.json'%o))
r=d['regimes']; b=r['beyond_l3']
print(o, 'inside', r['steady_state']['ms_per_launch'], r['steady_state_per_instance_refs']['ms_per_launch'], 'beyond', b['steady_state']['ms_per_launch'], b['steady_state_per_instance_refs']['ms_per_launch'])
PY
done
BATCHES=262144 timeout 600 python tools/prefetch_probe.py warm
