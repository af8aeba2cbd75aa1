#!/usr/bin/env python3
"""Throughput of the heterogeneous-family path: 65 536 quadrotor-like instances, every one with its
own (A, B, Q, R, rho) -- the batched Riccati precompute (riccati_kernel.hip.h), then the bench workload's closed loop on the HET
variant of the one-row kernel (per-instance lane tables re-loaded for every instance), against the shared-family kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm

prob, extra = tm.load_problem("quadrotor_20hz")
h = extra["hover"]
nx, nu, N, B = prob["nx"], prob["nu"], prob["N"], 65536
rng = np.random.default_rng(1)
A = prob["A"][None] + rng.normal(0, 1e-3, (B, nx, nx))
Bm = prob["B"][None] * (1 + rng.normal(0, 0.05, (B, 1, 1)))
t0 = time.perf_counter()
s = tm.TinyBatchSolver.hetero(A, Bm, None, np.tile(prob["Q"], (B, 1)), np.tile(prob["R"], (B, 1)), rng.uniform(3.0, 7.0, B), N)
t_setup = time.perf_counter() - t0
its = np.array([s.cache_instance(i, "riccati_iters")[0, 0] for i in range(0, B, 4096)])
print(f"setup of {B} families (upload + batched Riccati recursion + lane tables): {t_setup * 1e3:.1f} ms wall clock; Riccati iterations {its.min():.0f}..{its.max():.0f}")
xref = np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N))
x0 = np.array(h["x0"], dtype=np.float64)
fl = tm.flops_per_iter(nx, nu, N)


def episode(sol, label):
    sol.set_bound_constraints(np.full((nx, 1), h["x_min"]), np.full((nx, 1), h["x_max"]), np.full((nu, 1), h["u_min"]), np.full((nu, 1), h["u_max"]))
    sol.update_settings(max_iter=h["max_iter"])
    sol.set_option("advance_x0", 1)
    sol.set_option("steps_per_launch", 100)
    best = None
    for _ in range(4):
        sol.reset()
        sol.set_x_ref(xref, broadcast=True)
        sol.set_x0(x0, broadcast=True)
        sol.set_option("timing", 1)
        sol.solve_async()
        ms = float(sol.timing_ms()[0])
        best = ms if best is None else min(best, ms)
    st = sol.reduce_stats()
    print(f"{label}: {best:.2f} ms per 100-step episode x {B}: {B * 100 / best * 1e3:.3e} solves/s, {st[7] / best * 1e3:.3e} ADMM it/s "
          f"({st[7] / B:.0f} iterations per instance on average), FP64 fraction {st[7] * fl / (best * 1e-3) / 78.6e12:.3f}")


episode(s, "heterogeneous families (HET variant)")
s.close()
episode(tm.TinyBatchSolver.from_problem(prob, B), "one shared family (plain variant)")
