#!/usr/bin/env python3
"""Scale check: 2^20 (BASELINE config 5's batch) and 2^22 identical hover instances on one GPU -- index arithmetic past
2^31 bytes/elements, grid sizes, linear scaling of the solve time; every instance must equal instance 0 bit for bit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm

prob, extra = tm.load_problem("quadrotor_20hz")
h = extra["hover"]
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
for B in (1 << 16, 1 << 20, 1 << 22):
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), h["x_min"]), np.full((nx, 1), h["x_max"]), np.full((nu, 1), h["u_min"]), np.full((nu, 1), h["u_max"]))
    s.update_settings(max_iter=h["max_iter"])
    s.set_x_ref(np.tile(np.array(h["xref"], dtype=float).reshape(nx, 1), (1, N)), broadcast=True)
    s.set_x0(np.array(h["x0"], dtype=float), broadcast=True)
    s.set_option("advance_x0", 1)
    s.set_option("steps_per_launch", 20)
    s.set_option("timing", 1)
    s.solve_async()
    ms = float(s.timing_ms()[0])
    st = s.reduce_stats()
    it = s.status()["iter"]
    u = s.get("u")
    x0 = s.get("x0")
    ok = bool(np.all(u == u[:1]) and np.all(x0 == x0[:1]) and np.all(it == it[0]))
    print(f"B={B}: 20 fused MPC steps {ms:.2f} ms = {B * 20 / ms * 1e3:.3e} solves/s, accumulated iterations/instance {st[7] / B:.1f}, "
          f"last-step iterations {int(it[0])}, all instances identical: {ok}, last instance u0 {u[-1, :, 0]}")
    assert ok and st[7] / B == 691.0           # the first 20 entries of the reference's 882-iteration hover sequence
    s.close()
