import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as bc
import tinympc_amd as tm
out = bc.run_all(budget_s=600.0)
print("@@JIT@@" + json.dumps(tm.jit_used()))
print({k: v.get("ms") for k, v in out.items()})
