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


# ---- round 5: wide and long shapes, the tile kernel's per-instance form (EXT bit 1) against the same shape's plain tile form
def wide(dims, Bw):
    wnx, wnu, wN = dims
    fam, r2 = tm.random_problem(wnx, wnu, wN)
    A0, B0 = np.asarray(fam["A"]), np.asarray(fam["B"])
    Aw = A0[None] * (1 + r2.normal(0, 1e-3, (Bw, 1, 1)))
    Bw_ = B0[None] * (1 + r2.normal(0, 0.05, (Bw, 1, 1)))
    rho = r2.uniform(0.8, 1.2, Bw) * fam["rho"]
    t0 = time.perf_counter()
    het = tm.TinyBatchSolver.hetero(Aw, Bw_, None, np.tile(fam["Q"], (Bw, 1)), np.tile(fam["R"], (Bw, 1)), rho, wN)
    t_set = time.perf_counter() - t0
    hom = tm.TinyBatchSolver.from_problem(fam, Bw)
    x0w = r2.uniform(-1, 1, (Bw, wnx))
    out = []
    for sol, label in ((het, "per-instance data"), (hom, "one shared family")):
        sol.set_bound_constraints(np.full((wnx, 1), -1e17), np.full((wnx, 1), 1e17), np.full((wnu, 1), -0.5), np.full((wnu, 1), 0.5))
        sol.update_settings(max_iter=500)
        best = None
        for _ in range(3):
            sol.reset()
            sol.set_x0(x0w)
            sol.set_option("timing", 1)
            sol.solve_async()
            ms = float(np.sum(sol.timing_ms()))
            best = ms if best is None else min(best, ms)
        st = sol.reduce_stats()
        out.append((label, sol.kernel_path(), best, st[0]))
        sol.close()
    flw = tm.flops_per_iter(wnx, wnu, wN)
    print(f"({wnx},{wnu},{wN}) x {Bw}: setup of the families {t_set * 1e3:.0f} ms; " +
          "; ".join(f"{lb} [{kp}]: {ms:.2f} ms, {it / ms * 1e3:.3e} ADMM it/s, FP64 {it * flw / (ms * 1e-3) / 78.6e12:.3f}" for lb, kp, ms, it in out))


for dims, Bw in (((12, 8, 10), 65536), ((20, 8, 10), 32768), ((20, 4, 30), 16384), ((20, 8, 50), 8192)):
    try:
        wide(dims, Bw)
    except Exception as e:                                # noqa: BLE001
        print(dims, "failed:", repr(e))


# ---- round 5: tracking-style fused episodes (reference window + reset_duals, examples/quadrotor_tracking.cpp:77-106) on wide / long
# shapes -- the tile kernel's EXT bit-0 form on the shape's fast box form -- against the same fused episode without a window
def tracking(dims, Bw, T=10):
    wnx, wnu, wN = dims
    fam, r2 = tm.random_problem(wnx, wnu, wN)
    x0w = r2.uniform(-1, 1, (Bw, wnx))
    traj = r2.normal(0, 0.2, (wN + T + 2, wnx))
    out = []
    for windowed in (False, True):
        sol = tm.TinyBatchSolver.from_problem(fam, Bw)
        sol.set_bound_constraints(np.full((wnx, 1), -1e17), np.full((wnx, 1), 1e17), np.full((wnu, 1), -0.5), np.full((wnu, 1), 0.5))
        sol.update_settings(max_iter=100)
        sol.set_option("advance_x0", 1)
        sol.set_option("steps_per_launch", T)
        if windowed:
            sol.set_reference_trajectory(traj)
            sol.set_option("reset_duals", 1)
        best = None
        for _ in range(3):
            sol.reset()
            sol.set_x0(x0w)
            if windowed:
                sol.set_option("traj_step", 0)
            sol.set_option("timing", 1)
            sol.solve_async()
            ms = float(np.sum(sol.timing_ms()))
            best = ms if best is None else min(best, ms)
        st = sol.reduce_stats()
        out.append((("window + reset_duals" if windowed else "fixed reference"), sol.kernel_path(), best, st[7]))
        sol.close()
    print(f"({wnx},{wnu},{wN}) x {Bw}, {T} fused MPC steps: " + "; ".join(f"{lb} [{kp}]: {ms:.2f} ms, {it / ms * 1e3:.3e} ADMM it/s" for lb, kp, ms, it in out))


for dims, Bw in (((20, 8, 10), 32768), ((12, 8, 30), 16384), ((20, 8, 50), 8192)):
    try:
        tracking(dims, Bw)
    except Exception as e:                                # noqa: BLE001
        print(dims, "tracking failed:", repr(e))
