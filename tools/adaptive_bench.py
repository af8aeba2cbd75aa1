#!/usr/bin/env python3
"""Throughput of the adaptive-rho variant (ADAPT) on the bench workload: 65 536 quadrotor-hover instances, the 100-step closed
loop fused in one launch (813 ADMM iterations per instance with the reference's sensitivity tables, 882 without adaptation)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm

prob, extra = tm.load_problem("quadrotor_20hz")
h = extra["hover"]
nx, nu, N, B = prob["nx"], prob["nu"], prob["N"], 65536
t = json.load(open(os.path.join(os.path.dirname(tm.__file__), "data", "sensitivity_quadrotor.json")))
sens = [np.array(t[k]["data"]).reshape(t[k]["cols"], t[k]["rows"]).T for k in ("dKinf_drho", "dPinf_drho", "dC1_drho", "dC2_drho")]
xref = np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N))
for adaptive in (0, 1):
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), h["x_min"]), np.full((nx, 1), h["x_max"]), np.full((nu, 1), h["u_min"]), np.full((nu, 1), h["u_max"]))
    s.update_settings(max_iter=h["max_iter"])
    if adaptive:
        s.set_sensitivity(*sens)
        s.set_adaptive_rho(1, 1.0, 100.0, 1)
    s.set_option("advance_x0", 1)
    s.set_option("steps_per_launch", 100)
    best = None
    for _ in range(4):
        s.reset()
        s.set_x_ref(xref, broadcast=True)
        s.set_x0(np.array(h["x0"], dtype=np.float64), broadcast=True)
        s.set_option("timing", 1)
        s.solve_async()
        ms = float(s.timing_ms()[0])
        best = ms if best is None else min(best, ms)
    st = s.reduce_stats()
    fl = tm.flops_per_iter(nx, nu, N)
    print(f"adaptive_rho={adaptive}: {best:.2f} ms per 100-step episode x {B}: {B * 100 / best * 1e3:.3e} solves/s, {st[7] / best * 1e3:.3e} ADMM it/s "
          f"({st[7] / B:.0f} iterations per instance), FP64 fraction {st[7] * fl / (best * 1e-3) / 78.6e12:.3f} (box-iteration FLOPs only)")
    s.close()
