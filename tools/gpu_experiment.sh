#!/bin/bash
O=$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_repack.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt | cut -c1-300
python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, "tools")
import config_bench as cb, tinympc_amd as tm
# config 3 with the default (automatic) split: kernel time of the 2nd..4th solve
prob, extra = tm.load_problem("quadrotor_20hz")
nx, nu, N, B = 12, 4, 10, 262144
traj = np.array(extra["y_axis_line"]); rng = np.random.default_rng(20260923)
k = rng.integers(0, 291, B)
Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
Uref = rng.normal(0, 0.05, (B, nu, N - 1)); x0 = Xref[:, :, 0].copy(); x0[:, :3] += rng.normal(0, 0.1, (B, 3))
s = tm.TinyBatchSolver.from_problem(prob, B)
s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
s.update_settings(max_iter=100); s.set_x_ref(Xref); s.set_u_ref(Uref)
for mode in (-1, 0, 10):
    s.set_option("repack_after", mode)
    t = []
    for r in range(12):
        s.reset(); s.set_x0(x0); s.set_option("timing", 1); s.solve_async(); t.append(float(s.timing_ms()[0]))
    st = s.reduce_stats()
    print("config3 repack_after=%d: ms %s  auto K %d (predicted %.3f, measured %.3f of plain, verdict %d)  it/s %.3e" % (mode, np.round(t, 3).tolist(), s.get_option("auto_split_k") if mode < 0 else mode, s.get_option("auto_split_permille") / 1000 if mode < 0 else 0, s.get_option("auto_split_measured_permille") / 1000, s.get_option("auto_split_verdict"), st[0] / (min(t[3:]) * 1e-3)))
s.close()
PY
