#!/usr/bin/env python3
"""Steady-state (warm, 1-2 ADMM iterations per solve) launch time of the hover workload, one launch per MPC step, for
the byte-saving launch forms: x|u write-back on / off, shared / per-instance reference record, grid shape.  Per form:
median over replays of the per-step kernel time (HIP events), steps 70-99 split by their iteration count."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm

prob, extra = tm.load_problem("quadrotor_20hz")
h = extra["hover"]
nx, nu, N, B = prob["nx"], prob["nu"], prob["N"], int(os.environ.get("BATCH", "65536"))
s = tm.TinyBatchSolver.from_problem(prob, B)
s.set_bound_constraints(np.full((nx, 1), h["x_min"]), np.full((nx, 1), h["x_max"]), np.full((nu, 1), h["u_min"]), np.full((nu, 1), h["u_max"]))
s.update_settings(max_iter=h["max_iter"])
s.set_option("advance_x0", 1)
xref = np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N))
x0 = np.array(h["x0"], dtype=np.float64)
S = nx * N + nu * (N - 1)
print(f"batch {B}; bytes_warm = {s.algorithmic_bytes()} B per solve")
print("| store_primal | share_ref | grid waves/CU | 1-iteration steps: us | formula TB/s | 2-iteration steps: us | formula TB/s |")
print("|---|---|---|---|---|---|---|")
for sp in (1, 0):
    for sr in (0, 1):
        for g in (0, 8, 16, 32):
            s.set_option("store_primal", sp); s.set_option("share_ref", sr); s.set_option("grid_waves_per_cu", g)
            runs, its = [], None
            for _ in range(7):
                s.reset()
                s.set_x_ref(xref, broadcast=True)
                s.set_x0(x0, broadcast=True)
                s.set_option("step_log", 0)
                s.set_option("timing", 100)
                it = []
                for k in range(100):
                    s.solve_async()
                    if its is None and k >= 70:
                        it.append(int(s.status()["iter"][0]))
                if its is None:
                    its = np.array(it)
                runs.append(s.timing_ms()[70:])
            ms = np.median(np.array(runs[1:]), axis=0)          # (the first replay carries the status read-backs)
            one, two = ms[its == 1], ms[its == 2]
            f = s.algorithmic_bytes() * B
            print(f"| {sp} | {sr} | {g} | {one.mean()*1e3:.1f} | {f/one.mean()/1e9:.2f} | {two.mean()*1e3:.1f} | {f/two.mean()/1e9:.2f} |", flush=True)
s.close()
