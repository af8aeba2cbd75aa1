#!/usr/bin/env python3
"""Device-memory leak check: free HBM before / after 300 create-use-destroy cycles that touch every optional buffer
(linear records, debug buffers, trajectories, step logs, heterogeneous tables, the compat contexts)."""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import tinympc_amd as tm, scenarios as sc, pod, fuzz_compat

def free():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]

prob, extra = tm.load_problem("quadrotor_20hz")
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
L = tm.lib(); fuzz_compat.proto(L)
rng = np.random.default_rng(0)
devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
def cycle(i):
    B = 2000
    if i % 3 == 0:
        s = tm.TinyBatchSolver.hetero(np.stack([prob["A"]] * 64), np.stack([prob["B"]] * 64), np.stack([prob["f"]] * 64),
                                      np.stack([prob["Q"]] * 64), np.stack([prob["R"]] * 64), np.full(64, prob["rho"]), N)
    else:
        s = tm.TinyBatchSolver.from_problem(prob, B)
        s.set_linear_constraints(np.ones((1, nx)), [3.0], np.ones((1, nu)), [6.0])
        s.set_tv_linear_constraints(np.ones((N, nx)), np.full((1, N), 3.0), np.ones((N - 1, nu)), np.full((1, N - 1), 6.0))
        s.update_settings(max_iter=5, en_state_linear=i % 2, en_tv_input_linear=1 - i % 2)
        s.set_option("debug", 1)
        s.set_option("repack_after", 2)                         # the split solve's index lists
    s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.solve()
    if i % 3 == 1:
        s.update_settings(max_iter=5)
        s.set_reference_trajectory(np.zeros((40, nx)), np.zeros(B, dtype=np.int32))
        s.set_option("steps_per_launch", 4); s.set_option("step_log", 1); s.set_option("debug", 0)
        s.solve_async(); s.step_log(4)
    s.phase("update_slack")
    s.close()
    ms = [pod.mat(prob["A"]), pod.mat(prob["B"]), pod.mat(prob["f"]), pod.mat(np.diag(prob["Q"])), pod.mat(np.diag(prob["R"]))]
    sp = C.POINTER(pod.TinySolver)()
    L.tiny_setup(C.byref(sp), *[C.byref(m[0]) for m in ms], prob["rho"], nx, nu, N, 0)
    sp.contents.settings.contents.max_iter = 3
    L.tiny_solve(sp)
    L.tiny_destroy(sp)

os.dup2(devnull, 1)
for i in range(6): cycle(i)
f0 = free()
for i in range(300): cycle(i)
f1 = free()
os.dup2(saved, 1)
print(f"free HBM before {f0 / 2**20:.1f} MiB, after 300 cycles {f1 / 2**20:.1f} MiB, delta {(f0 - f1) / 2**20:.2f} MiB")
sys.exit(0 if f0 - f1 < 8 * 2**20 else 1)
