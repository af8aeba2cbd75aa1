#!/usr/bin/env python3
"""Phase clocks of the (6,3,10) cone kernel's iteration (instrumented build: python tools/build_variants.py socclk, then
TINYMPC_AMD_LIB=$PWD/tinympc_amd/libtinympc_amd_socclk.so python tools/soc_phase_clocks.py): shader cycles per wave-iteration
(wall clock of a wave: with two waves per SIMD about twice its share of the SIMD) in the backward sweep, the forward sweep, the cone
step (+ the termination test in the cone variants) and the tail of the iteration."""
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
    s.update_settings(max_iter=100, check_termination=int(os.environ.get("CHECK", "0")), en_state_soc=ss, en_input_soc=si)
    s.set_option("uniform_bounds", int(os.environ.get("UB", "1")))
    for _ in range(2):
        s.reset(); s.set_x0(x0); s.set_option("timing", 1); s.solve_async(); ms = float(s.timing_ms()[0])
    st = s.status()
    it = np.maximum(st["iter"], 1)
    r = np.stack([st["primal_residual_state"], st["primal_residual_input"], st["dual_residual_state"], st["dual_residual_input"]], axis=1) / it[:, None]
    print(f"{name:11s}: {ms:.3f} ms; per wave-iteration cycles: backward {r[:,0].mean():.0f}, forward {r[:,1].mean():.0f}, cone step {r[:,2].mean():.0f}, tail {r[:,3].mean():.0f}; "
          f"sum {r.sum(1).mean():.0f} (kernel time x 2.4 GHz / iterations / waves per slot: {ms*1e-3*2.4e9/100/(B/4/2048):.0f})")
    s.close()
