#!/usr/bin/env python3
"""Config 3 (divergent cold solves) with a split solve, for a kernel trace: one warm-up solve, then ONE traced solve.
  rocprofv3 --kernel-trace --output-format csv -d <dir> -o c3 -- python tools/repack_trace.py <repack_after> [batch]
prints nothing but the iteration total; the per-stage kernel durations are in the trace."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm  # noqa: E402

cap = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
prob, extra = tm.load_problem("quadrotor_20hz")
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
traj = np.array(extra["y_axis_line"])
rng = np.random.default_rng(20260923)
k = rng.integers(0, 291, B)
Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
Uref = rng.normal(0, 0.05, (B, nu, N - 1))
x0 = Xref[:, :, 0].copy()
x0[:, :3] += rng.normal(0, 0.1, (B, 3))
s = tm.TinyBatchSolver.from_problem(prob, B)
s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
s.update_settings(max_iter=100)
s.set_x_ref(Xref)
s.set_u_ref(Uref)
s.set_option("repack_after", cap)
for kv in filter(None, os.environ.get("TINYMPC_OPTS", "").split(",")):
    kk, vv = kv.split("=")
    s.set_option(kk, int(vv))
for _ in range(2):
    s.reset()
    s.set_x0(x0)
    s.solve()
print("iterations", s.reduce_stats()[0])
s.close()
