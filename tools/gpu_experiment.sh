#!/bin/bash
O=$1; mkdir -p $O
python - <<'PY' 2>&1 | tee $O/config3_tile_dyn.txt
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import tinympc_amd as tm
B=262144
prob, extra = tm.load_problem("quadrotor_20hz")
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
traj = np.array(extra["y_axis_line"])
rng = np.random.default_rng(20260923)
k = rng.integers(0, 291, B)
Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
Uref = rng.normal(0, 0.05, (B, nu, N - 1))
x0 = Xref[:, :, 0].copy(); x0[:, :3] += rng.normal(0, 0.1, (B, 3))
for name, opts in (("one-row plain", {"repack_after": 0}), ("one-row auto split", {}), ("tile LM=4 static", {"prefer_tile": 1, "tile_lm": 4, "tile_dyn": 0}),
                   ("tile LM=4 dynamic", {"prefer_tile": 1, "tile_lm": 4, "tile_dyn": 1})):
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=100)
    for kk, v in opts.items(): s.set_option(kk, v)
    s.set_x_ref(Xref); s.set_u_ref(Uref); s.set_x0(x0)
    ms=[]
    for _ in range(8):
        s.reset(); s.set_option("timing", 1); s.solve_async(); ms.append(float(s.timing_ms()[0]))
    st = s.reduce_stats()
    print("%-22s %s  best %.3f ms (last4 %.3f)  iters %d  path %s" % (name, "", min(ms), min(ms[4:]), st[0], s.kernel_path()))
    s.close()
PY
