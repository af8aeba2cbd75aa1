"""GPU, round 6: is the per-instance-data form of (20,8,10) slower after other handles have lived and died in the process?  No -- it is slower when
torch was imported first (profiles/r06_jit_compiler_probe.md; tools/jit_compiler_probe.py is the clean form of this probe)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch; torch.zeros(1, device="cuda")
import bench_configs as bc
import tinympc_amd as tm
def het(tag, B=32768):
    e, _ = bc.hetero_cell(20, 8, 10, B=B)
    print(tag, "hetero ms %.3f (min %.3f max %.3f)" % (e["ms"], e["ms_min"], e["ms_max"]), flush=True)
het("fresh process")
e, _ = bc.sweep_cell(20, 8, 50); print("sweep_20_8_50 ms %.2f" % e["ms"], flush=True)
het("after sweep_20_8_50")
het("again")
e, _ = bc.sweep_cell(4, 2, 10); print("sweep_4_2_10 ms %.3f" % e["ms"], flush=True)
het("after sweep_4_2_10")
# hold a big allocation alive while the hetero handle is created
import ctypes
prob, rng = tm.random_problem(20, 8, 50)
big = tm.TinyBatchSolver.from_problem(prob, 131072)
het("with a 10 GB handle alive")
big.close()
het("after closing it")
