#!/usr/bin/env python3
"""One launch of the (6,3,10) rocket kernel, 65 536 instances x 100 iterations (no early exit), for rocprofv3 counter passes:
    python tools/soc_counters.py [box|input|state|both]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm
which = sys.argv[1] if len(sys.argv) > 1 else "input"
ss, si = {"box": (0, 0), "input": (0, 1), "state": (1, 0), "both": (1, 1)}[which]
prob, extra = tm.load_problem("rocket_landing_20hz")
m = extra["mpc"]; nx, nu, N = prob["nx"], prob["nu"], prob["N"]; B = 65536
rng = np.random.default_rng(1)
x0 = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, (B, nx)))
s = tm.TinyBatchSolver.from_problem(prob, B)
s.set_bound_constraints(np.array(m["x_min"]), np.array(m["x_max"]), np.full((nu, 1), m["u_min"]), np.full((nu, 1), m["u_max"]))
s.set_cone_constraints(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"], m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])
s.update_settings(max_iter=100, check_termination=0, en_state_soc=ss, en_input_soc=si)
for _ in range(2):
    s.reset(); s.set_x0(x0); s.set_option("timing", 1); s.solve_async(); ms = float(s.timing_ms()[0])
print(which, ms, "ms")
s.close()
