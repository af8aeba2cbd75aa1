#!/usr/bin/env python3
"""Exact cost of one ADMM iteration of the (6,3,10) rocket kernel without and with cone projections: 65 536 instances,
100 iterations each (check_termination = 0: no early exit), one launch; cycles per wave-iteration of SIMD time."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm
prob, extra = tm.load_problem("rocket_landing_20hz")
m = extra["mpc"]; nx, nu, N = prob["nx"], prob["nu"], prob["N"]; B = 65536
rng = np.random.default_rng(1)
x0 = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, (B, nx)))
for name, ss, si in (("box only", 0, 0), ("input cone", 0, 1), ("state cone", 1, 0), ("both cones", 1, 1)):
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.array(m["x_min"]), np.array(m["x_max"]), np.full((nu, 1), m["u_min"]), np.full((nu, 1), m["u_max"]))
    s.set_cone_constraints(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"], m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])
    s.update_settings(max_iter=100, check_termination=0, en_state_soc=ss, en_input_soc=si)
    best = 1e9
    for _ in range(3):
        s.reset(); s.set_x0(x0); s.set_option("timing", 1); s.solve_async(); best = min(best, float(s.timing_ms()[0]))
    per_wave_iter_cycles = best * 1e-3 / 100 / (B / 4 / 1024) * 2.4e9
    print(f"{name:11s}: {best:.3f} ms per 100 iterations = {B * 100 / best * 1e3:.3e} ADMM it/s, {per_wave_iter_cycles:.0f} SIMD cycles per wave-iteration")
    s.close()
