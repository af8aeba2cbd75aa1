"""Where the 40 us of one tiny_solve(TinySolver*) go: the library's own trace (TINYMPC_AMD_TRACE: gather / enqueue / wait / scatter, microseconds) of eight warm
quadrotor solves through the reference's structs.  python tools/dropin_trace.py 2>&1 | tail"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import pod, tinympc_amd as tm, scenarios as sc
L = tm.lib()
prob, extra = sc.load_problem("quadrotor_20hz"); h = extra["hover"]
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
L.tiny_setup.argtypes = [C.POINTER(C.POINTER(pod.TinySolver))] + [C.POINTER(pod.Mat)] * 5 + [C.c_double] + [C.c_int] * 4
L.tiny_set_bound_constraints.argtypes = [C.POINTER(pod.TinySolver)] + [C.POINTER(pod.Mat)] * 4
L.tiny_solve.argtypes = [C.POINTER(pod.TinySolver)]
ms = [pod.mat(prob["A"]), pod.mat(prob["B"]), pod.mat(prob["f"]), pod.mat(np.diag(prob["Q"])), pod.mat(np.diag(prob["R"]))]
sp = C.POINTER(pod.TinySolver)()
assert L.tiny_setup(C.byref(sp), *[C.byref(m[0]) for m in ms], prob["rho"], nx, nu, N, 0) == 0
bs = [pod.mat(np.full((nx, N), h["x_min"])), pod.mat(np.full((nx, N), h["x_max"])), pod.mat(np.full((nu, N - 1), h["u_min"])), pod.mat(np.full((nu, N - 1), h["u_max"]))]
assert L.tiny_set_bound_constraints(sp, *[C.byref(m[0]) for m in bs]) == 0
w = sp.contents.work.contents
pod.to_np(w.Xref)[...] = np.array(h["xref"]).reshape(-1, 1)
pod.to_np(w.x)[:, 0] = h["x0"]
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
for _ in range(50): L.tiny_solve(sp)
os.environ["TINYMPC_AMD_TRACE"] = "1"
for _ in range(8): L.tiny_solve(sp)
os.dup2(saved, 1)
