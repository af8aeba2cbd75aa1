mkdir -p gpurun_out/r06c
for o in "prefetch=0" "prefetch=-1" "prefetch=0" "prefetch=-1"; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --opt $o --details gpurun_out/r06c/bd.json > gpurun_out/r06c/b.json 2>/dev/null
python - "$o" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r06c/bd.json'))
r=d['regimes']; b=r['beyond_l3']
print(sys.argv[1], 'inside', round(r['steady_state']['ms_per_launch'],4), round(r['steady_state_per_instance_refs']['ms_per_launch'],4), 'beyond', round(b['steady_state']['ms_per_launch'],4), round(b['steady_state_per_instance_refs']['ms_per_launch'],4), 'config3', d['configs']['config3']['ms'] if 'configs' in d else None)
PY
done
