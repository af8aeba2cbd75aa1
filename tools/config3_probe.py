#!/usr/bin/env python3
"""Config 3 (262 144 cold tracking solves) under launch-grid options: one wave per tile (the default) against persistent waves that
walk the tiles (option grid_waves_per_cu), plain launch and the automatic split; median kernel ms of the settled repetitions."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm

B = int(os.environ.get("B", "262144"))
prob, extra = tm.load_problem("quadrotor_20hz")
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
traj = np.array(extra["y_axis_line"])
rng = np.random.default_rng(20260923)
k = rng.integers(0, 291, B)
Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
Uref = rng.normal(0, 0.05, (B, nu, N - 1))
x0 = Xref[:, :, 0].copy()
x0[:, :3] += rng.normal(0, 0.1, (B, 3))
STAG = [int(v) for v in os.environ.get("STAGGER", "0").split(",")]
GRIDS = [int(v) for v in os.environ.get("GRIDS", "0,4,8,16,32").split(",")]
print("| grid waves/CU | stagger | repack_after | ms median (settled) | ms min | split K | verdict |")
print("|---|---|---|---|---|---|---|")
for g, stg in [(g, t) for g in GRIDS for t in STAG]:
    for ra in (0, -1):
        s = tm.TinyBatchSolver.from_problem(prob, B)
        s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
        s.update_settings(max_iter=100)
        s.set_x_ref(Xref); s.set_u_ref(Uref); s.set_x0(x0)
        s.set_option("grid_waves_per_cu", g)
        s.set_option("stagger", stg)
        s.set_option("repack_after", ra)
        ms = []
        for _ in range(5 if ra == 0 else 14):
            s.reset()
            s.set_option("timing", 1)
            s.solve_async()
            ms.append(float(np.sum(s.timing_ms())))
        settled = ms[2:] if ra == 0 else ms[7:]
        print(f"| {g} | {stg} | {ra} | {np.median(settled):.4f} | {np.min(settled):.4f} | {s.get_option('auto_split_k')} | {s.get_option('auto_split_verdict')} |", flush=True)
        s.close()
