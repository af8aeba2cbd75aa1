#!/bin/bash
O=$1; mkdir -p $O
python - <<'PY' 2>&1 | tee $O/alt_trace.txt
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import tinympc_amd as tm
for (nx,nu,N) in ((8,2,10),(4,4,10),(12,4,30)):
    B=131072
    prob, rng = tm.random_problem(nx, nu, N)
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=500)
    s.set_x0(rng.uniform(-1, 1, (B, nx)))
    s.set_x_ref(np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1)), N, axis=2))
    print((nx,nu,N))
    for n in range(14):
        s.reset(); s.set_option("timing", 1); s.solve_async(); ms=float(s.timing_ms()[0])
        print("  solve %2d  %.3f ms  split_verdict=%d tile_verdict=%d dyn=%d" % (n, ms, s.get_option("auto_split_verdict"), s.get_option("tile_alt_verdict"), s.get_option("last_tile_dyn")))
    s.close()
PY
