#!/usr/bin/env python3
"""Per-phase wall clocks of the register kernel on BASELINE config 3 (262 144 cold solves, divergent iteration counts)
and on the same batch made uniform.  Needs the instrumented library:
    python tools/build_phase_clocks.py && TINYMPC_AMD_LIB=$PWD/tinympc_amd/libtinympc_amd_clk.so python tools/phase_clocks.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm

def run(B, chunk, uniform):
    prob, extra = tm.load_problem("quadrotor_20hz")
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    traj = np.array(extra["y_axis_line"])
    rng = np.random.default_rng(20260923)
    k = rng.integers(0, 291, B)
    Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
    Uref = rng.normal(0, 0.05, (B, nu, N - 1))
    x0 = Xref[:, :, 0].copy()
    x0[:, :3] += rng.normal(0, 0.1, (B, 3))
    if uniform:
        Xref[:] = Xref[0]; Uref[:] = Uref[0]; x0[:] = x0[0]
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=100)
    s.set_x_ref(Xref); s.set_u_ref(Uref)
    for _ in range(2):
        s.reset(); s.set_x0(x0); s.set_option("timing", 1); s.solve_async(); ms = float(s.timing_ms()[0])
    st = s.status()
    r = np.stack([st["primal_residual_state"], st["primal_residual_input"], st["dual_residual_state"], st["dual_residual_input"]], axis=1)
    it = st["iter"]
    s.close()
    return ms, it, r

for uniform in (0, 1):
    ms, it, r = run(262144, 0, uniform)
    r = np.asarray(r).reshape(-1, 4) * 0.01   # us
    m4 = it.reshape(-1, 4).max(1).mean()
    print(f"uniform={uniform}: kernel {ms:.3f} ms, iterations mean {it.mean():.2f}, mean of max over the 4 rows of a wave {m4:.2f}; per wave us: "
          f"prologue {r[:,0].mean():.2f}, load {r[:,1].mean():.2f}, compute {r[:,2].mean():.2f} ({r[:,2].mean()/m4:.3f}/iteration), "
          f"store {r[:,3].mean():.2f}; kernel time per wave slot {ms*1e3*2048/(262144/4):.1f} us")
