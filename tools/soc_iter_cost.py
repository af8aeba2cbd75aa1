#!/usr/bin/env python3
"""Cost of one ADMM iteration of the (6,3,10) rocket kernel without and with cone projections: 65 536 instances, 100 iterations
each (check_termination = 0 unless CHECK=1: no early exit), one launch per setting.  No references are set: the cones are ACTIVE (every
projection pass on its exact path); the `warm` figure repeats the launch from the state it left, where no cone is active any more --
the regime of BASELINE's config 4 (identity fast path of the cone step).

    python tools/soc_iter_cost.py                 SIMD cycles per wave-iteration (kernel time x 2.4 GHz), box / input / state / both
    python tools/soc_iter_cost.py --one input     ONE setting, two launches: the command to put under rocprofv3 --pmc (SQ counter passes)
    python tools/soc_iter_cost.py --clocks        with the instrumented build (python tools/build_variants.py socclk;
                                                  TINYMPC_AMD_LIB=$PWD/tinympc_amd/libtinympc_amd_socclk.so): s_memtime phase clocks of
                                                  the iteration -- backward sweep, forward sweep, cone step (+ the termination test),
                                                  tail -- as wall cycles of a wave (two waves per SIMD: about twice its SIMD share)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm
prob, extra = tm.load_problem("rocket_landing_20hz")
m = extra["mpc"]; nx, nu, N = prob["nx"], prob["nu"], prob["N"]; B = 65536
rng = np.random.default_rng(1)
x0 = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, (B, nx)))
SETTINGS = (("box only", 0, 0), ("input cone", 0, 1), ("state cone", 1, 0), ("both cones", 1, 1))
clocks = "--clocks" in sys.argv
if "--one" in sys.argv:
    w = sys.argv[sys.argv.index("--one") + 1]
    SETTINGS = tuple(s for s in SETTINGS if s[0].split()[0] == w)
for name, ss, si in SETTINGS:
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.array(m["x_min"]), np.array(m["x_max"]), np.full((nu, 1), m["u_min"]), np.full((nu, 1), m["u_max"]))
    s.set_cone_constraints(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"], m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])
    s.update_settings(max_iter=100, check_termination=int(os.environ.get("CHECK", "0")), en_state_soc=ss, en_input_soc=si)
    s.set_option("uniform_bounds", int(os.environ.get("UB", "1")))
    best = 1e9
    for _ in range(2 if "--one" in sys.argv else 3):
        s.reset(); s.set_x0(x0); s.set_option("timing", 1); s.solve_async(); best = min(best, float(s.timing_ms()[0]))
    line = (f"{name:11s}: {best:.3f} ms per 100 iterations = {B * 100 / best * 1e3:.3e} ADMM it/s, "
            f"{best * 1e-3 / 100 / (B / 4 / 1024) * 2.4e9:.0f} SIMD cycles per wave-iteration")
    # the same 100 iterations again WITHOUT a reset: the warm state of the solve before, no cone active any more -- every pass takes
    # its identity path (what the closed loop of config 4 sees after its first solves)
    s.set_option("timing", 1); s.solve_async(); warm = float(s.timing_ms()[0])
    line += f" | warm (no cone active): {warm:.3f} ms, {warm * 1e-3 / 100 / (B / 4 / 1024) * 2.4e9:.0f} cycles"
    if clocks:
        st = s.status()
        it = np.maximum(st["iter"], 1)
        r = np.stack([st["primal_residual_state"], st["primal_residual_input"], st["dual_residual_state"], st["dual_residual_input"]], axis=1) / it[:, None]
        line += f"; wave wall cycles per iteration: backward {r[:,0].mean():.0f}, forward {r[:,1].mean():.0f}, cone step {r[:,2].mean():.0f}, tail {r[:,3].mean():.0f}"
    print(line, flush=True)
    s.close()
