#!/bin/bash
O=$1; mkdir -p $O; export O
timeout 600 python - > $O/trace_8_2_10.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import tinympc_amd as tm
for (nx, nu, N) in ((8, 2, 10), (4, 4, 10), (12, 2, 10)):
    B = 131072
    prob, rng = tm.random_problem(nx, nu, N)
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=500)
    s.set_x0(rng.uniform(-1, 1, (B, nx)))
    s.set_x_ref(np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1)), N, axis=2))
    print((nx, nu, N))
    for i in range(12):
        s.reset(); s.set_option("timing", 1); s.solve_async(); ms = float(np.sum(s.timing_ms()))
        print(" ", i, "%.4f ms" % ms, "split verdict", s.get_option("auto_split_verdict"), "growth", s.get_option("auto_split_growth"), "gv", s.get_option("auto_split_growth_verdict"),
              "tile", s.get_option("tile_alt_verdict"), "last_tile_dyn", s.get_option("last_tile_dyn"), "measured permille", s.get_option("auto_split_measured_permille"))
    print("  K", s.get_option("auto_split_k"))
    s.close()
PY
cat $O/trace_8_2_10.txt
timeout 600 python -m pytest tests/test_gpu_repack.py -m gpu -q > $O/pytest.txt 2>&1; grep -n "passed\|failed" $O/pytest.txt | tail -3
