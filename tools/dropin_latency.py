#!/usr/bin/env python3
"""Host-boundary cost of the drop-in entry points (host structs -> HBM -> solve -> host structs): per-call latency of
tiny_solve on one TinySolver and the PCIe-inclusive rate of tiny_solve_batch on n solvers (quadrotor hover, warm)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pod, tinympc_amd as tm, scenarios as sc

L = tm.lib()
prob, extra = sc.load_problem("quadrotor_20hz")
h = extra["hover"]
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
L.tiny_setup.argtypes = [C.POINTER(C.POINTER(pod.TinySolver))] + [C.POINTER(pod.Mat)] * 5 + [C.c_double] + [C.c_int] * 4
L.tiny_set_bound_constraints.argtypes = [C.POINTER(pod.TinySolver)] + [C.POINTER(pod.Mat)] * 4
L.tiny_solve.argtypes = [C.POINTER(pod.TinySolver)]
L.tiny_solve_batch.argtypes = [C.POINTER(C.POINTER(pod.TinySolver)), C.c_int]
keep = []
def make():
    ms = [pod.mat(prob["A"]), pod.mat(prob["B"]), pod.mat(prob["f"]), pod.mat(np.diag(prob["Q"])), pod.mat(np.diag(prob["R"]))]
    keep.append(ms)
    sp = C.POINTER(pod.TinySolver)()
    assert L.tiny_setup(C.byref(sp), *[C.byref(m[0]) for m in ms], prob["rho"], nx, nu, N, 0) == 0
    bs = [pod.mat(np.full((nx, N), h["x_min"])), pod.mat(np.full((nx, N), h["x_max"])), pod.mat(np.full((nu, N - 1), h["u_min"])), pod.mat(np.full((nu, N - 1), h["u_max"]))]
    keep.append(bs)
    assert L.tiny_set_bound_constraints(sp, *[C.byref(m[0]) for m in bs]) == 0
    w = sp.contents.work.contents
    pod.to_np(w.Xref)[...] = np.array(h["xref"]).reshape(-1, 1)
    pod.to_np(w.x)[:, 0] = h["x0"]
    return sp
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1)
s1 = make()
os.dup2(devnull, 1)                      # "Solver converged in N iterations" lines
for _ in range(20): L.tiny_solve(s1)
t0 = time.perf_counter()
for _ in range(200): L.tiny_solve(s1)
t1 = time.perf_counter()
os.dup2(saved, 1)
print(f"tiny_solve, one TinySolver (12,4,10): {(t1 - t0) / 200 * 1e6:.0f} us per call (upload + 1 launch + download)")
n = 4096
arr = (C.POINTER(pod.TinySolver) * n)()
for k in range(n): arr[k] = make()
os.dup2(devnull, 1)
for _ in range(3): L.tiny_solve_batch(arr, n)
t0 = time.perf_counter()
for _ in range(10): L.tiny_solve_batch(arr, n)
t1 = time.perf_counter()
os.dup2(saved, 1)
print(f"tiny_solve_batch, {n} TinySolvers: {(t1 - t0) / 10 * 1e3:.2f} ms per call = {n * 10 / (t1 - t0):.3e} solves/s host-struct to host-struct")
