#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 600 python - > $O/c3_growth.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import tinympc_amd as tm
B = 262144
prob, extra = tm.load_problem("quadrotor_20hz")
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
traj = np.array(extra["y_axis_line"])
rng = np.random.default_rng(20260923)
k = rng.integers(0, 291, B)
Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
Uref = rng.normal(0, 0.05, (B, nu, N - 1))
x0 = Xref[:, :, 0].copy(); x0[:, :3] += rng.normal(0, 0.1, (B, 3))
s = tm.TinyBatchSolver.from_problem(prob, B)
s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
s.update_settings(max_iter=100)
s.set_x_ref(Xref); s.set_u_ref(Uref); s.set_x0(x0)
for i in range(14):
    s.reset(); s.set_option("timing", 1); s.solve_async(); ms = float(np.sum(s.timing_ms()))
    print(i, "%.4f ms" % ms, "verdict", s.get_option("auto_split_verdict"), "growth", s.get_option("auto_split_growth"), "growth verdict", s.get_option("auto_split_growth_verdict"), "tile", s.get_option("tile_alt_verdict"))
print("K", s.get_option("auto_split_k"), s.reduce_stats()[:2])
def run(n=8):
    ms = []
    for _ in range(n):
        s.reset(); s.set_option("timing", 1); s.solve_async(); ms.append(float(np.sum(s.timing_ms())))
    return "min %.4f median %.4f" % (min(ms), float(np.median(ms)))
print("auto as settled:", run(10))
for K in (9, 10):
    for g in (2, 4):
        s.set_option("repack_after", K); s.set_option("repack_growth", g)
        print("K", K, "growth", g, run(10))
s.set_option("repack_after", 0); print("plain", run(6))
PY
cat $O/c3_growth.txt
timeout 600 python -m pytest tests/test_gpu_repack.py -m gpu -q > $O/pytest.txt 2>&1; grep -n "passed\|failed" $O/pytest.txt | tail -3
