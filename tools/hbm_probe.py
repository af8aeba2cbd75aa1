#!/usr/bin/env python3
"""HBM-regime probe: time the solve kernel with max_iter = 0 (pure load + store of the instance records with the
real access pattern), 1 and 2 iterations, for several persistent-grid sizes.  Prints GB/s of algorithmic bytes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm

prob, extra = tm.load_problem("quadrotor_20hz")
h = extra["hover"]
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
for B in (65536, 262144):
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.set_x_ref(np.tile(np.array(h["xref"], dtype=float).reshape(nx, 1), (1, N)), broadcast=True)
    s.set_x0(np.array(h["x0"], dtype=float), broadcast=True)
    for max_iter in (0, 1, 2):
        for g in (0, 4, 8, 16):
            s.update_settings(max_iter=max_iter)
            s.set_option("grid_waves_per_cu", g)
            for _ in range(3):
                s.solve_async()
            s.set_option("timing", 20)
            for _ in range(20):
                s.solve_async()
            ms = np.median(s.timing_ms())
            print(f"B={B} max_iter={max_iter} grid_waves_per_cu={g:2d}: {ms*1e3:8.1f} us  {s.algorithmic_bytes()*B/ms/1e6:7.1f} GB/s")
    s.close()
