#!/usr/bin/env python3
"""Throughput of BASELINE configs 3 and 4 at full size on one GPU (config 2 = bench.py, config 5 = sweep_bench.py).

  config 3: quadrotor_tracking (12,4,10), 262 144 instances, per-instance random references around the y-axis line
            (SURVEY.md section 8(d) recipe), duals zeroed, ONE cold solve per instance (divergent iteration counts).
  config 4: rocket_landing (6,3,10), second-order-cone thrust constraint ON (en_input_soc = 1), 65 536 instances with
            perturbed initial states, the 90-step closed loop of examples/rocket_landing_mpc.cpp (plant on device,
            reference window refreshed per step as the example does), fused 1 launch per step.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm  # noqa: E402


def apply_opts(s):
    """TINYMPC_OPTS="one_shot=2,grid_waves_per_cu=8": solver options for an experiment, applied to every solver here."""
    for kv in filter(None, os.environ.get("TINYMPC_OPTS", "").split(",")):
        k, v = kv.split("=")
        s.set_option(k, int(v))


def config3(B=262144, reps=3):
    prob, extra = tm.load_problem("quadrotor_20hz")
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    traj = np.array(extra["y_axis_line"])
    rng = np.random.default_rng(20260923)
    k = rng.integers(0, 291, B)
    Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
    Uref = rng.normal(0, 0.05, (B, nu, N - 1))
    x0 = Xref[:, :, 0].copy()
    x0[:, :3] += rng.normal(0, 0.1, (B, 3))
    s = tm.TinyBatchSolver.from_problem(prob, B)
    apply_opts(s)
    s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=100)
    s.set_x_ref(Xref)
    s.set_u_ref(Uref)
    best, one_shot = None, {}
    for _ in range(reps):
        s.reset()
        s.set_x0(x0)
        s.set_option("timing", 1)
        s.solve_async()
        ms = float(s.timing_ms()[0])
        best = ms if best is None else min(best, ms)
    st = s.reduce_stats()
    it = s.status()["iter"]
    alg = s.algorithmic_bytes()
    for mode in (1, 2):                       # the same solves as one-shot launches: nothing of the zero state is read
        s.set_option("one_shot", mode)
        bm = None
        for _ in range(reps):
            s.reset()                         # (only so that the accumulated statistics restart)
            s.set_x0(x0)
            s.set_option("timing", 1)
            s.solve_async()
            ms = float(s.timing_ms()[0])
            bm = ms if bm is None else min(bm, ms)
        so = s.reduce_stats()
        assert so[0] == st[0], "one-shot iteration total differs"
        one_shot[f"one_shot={mode}"] = dict(kernel_ms=bm, solves_per_s=B / (bm * 1e-3), admm_iters_per_s=so[0] / (bm * 1e-3),
                                            bytes_per_solve=s.algorithmic_bytes(cold=(2 if mode == 1 else 1)))
    s.set_option("one_shot", 0)
    repack = {}
    for cap in (9, 10, 11, 12, 16, 24):      # split solves: stop at `cap`, compact the open instances, finish them densely packed
        s.set_option("repack_after", cap)
        bm = None
        for _ in range(reps):
            s.reset()
            s.set_x0(x0)
            s.set_option("timing", 1)
            s.solve_async()
            ms = float(s.timing_ms()[0])
            bm = ms if bm is None else min(bm, ms)
        so = s.reduce_stats()
        assert so[0] == st[0] and so[1] == st[1] and np.array_equal(s.status()["iter"], it), "split solve differs"
        repack[f"repack_after={cap}"] = dict(kernel_ms=bm, solves_per_s=B / (bm * 1e-3), admm_iters_per_s=so[0] / (bm * 1e-3))
    s.set_option("repack_after", 0)
    s.close()
    t = best * 1e-3
    return dict(one_shot=one_shot, repack=repack, config="quadrotor_tracking x262144, per-instance random refs, one cold solve", batch=B, kernel_ms=best,
                solves_per_s=B / t, admm_iters_per_s=st[0] / t, iters_per_solve=st[0] / B, solved_fraction=st[1] / B,
                iter_histogram={int(v): int(c) for v, c in zip(*np.unique(it, return_counts=True))},
                hbm_frac=alg * B / t / 8e12, fp64_frac=st[0] * tm.flops_per_iter(nx, nu, N) / t / 78.6e12)


def config4(B=65536):
    prob, extra = tm.load_problem("rocket_landing_20hz")
    m = extra["mpc"]
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    rng = np.random.default_rng(20260923)
    x0 = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, (B, nx)))
    xinit, xg = np.array(m["xinit"], dtype=float), np.array(m["xg"], dtype=float)
    traj = np.stack([xinit + (xg - xinit) * float(i) / (m["NTOTAL"] - 1) for i in range(m["NTOTAL"])])   # Xref window source
    s = tm.TinyBatchSolver.from_problem(prob, B)
    apply_opts(s)
    s.set_bound_constraints(np.array(m["x_min"]), np.array(m["x_max"]), np.full((nu, 1), m["u_min"]), np.full((nu, 1), m["u_max"]))
    s.set_cone_constraints(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"],
                           m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])
    s.update_settings(abs_pri_tol=m["abs_pri_tol"], max_iter=m["max_iter"], en_input_soc=1)
    uref = np.zeros((nu, N - 1)); uref[2, :] = m["uref_z"]
    steps = m["NTOTAL"] - N
    out = {}
    for T in (1, steps):
        s.reset()
        s.set_u_ref(uref, broadcast=True)
        s.set_reference_trajectory(traj)          # examples/rocket_landing_mpc.cpp:111-113 (window k .. k+N-1)
        s.set_x0(x0)
        s.set_option("advance_x0", 1)
        s.set_option("steps_per_launch", T)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps // T):
            s.solve_async()
        s.synchronize()
        dt = time.perf_counter() - t0
        st = s.reduce_stats()
        out[f"steps_per_launch={T}"] = dict(seconds=dt, solves_per_s=B * steps / dt, admm_iters_per_s=st[7] / dt,
                                            iters_per_solve=st[7] / (B * steps), solved_fraction=st[8] / (B * steps))
    s.close()
    return dict(config="rocket_landing x65536, input SOC on, 90-step closed loop (wall clock incl. launches)", batch=B, **out)


def linear_example(B=65536, tv=False):
    """examples/quadrotor_(tv_)linear_constraints.cpp: altitude ceiling + total-thrust half-spaces, boxes off; one
    cold solve of B perturbed instances on the register-resident LIN kernel and on the coverage kernel."""
    prob, _ = tm.load_problem("quadrotor_20hz")
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    rng = np.random.default_rng(1)
    x0 = np.array([-2.0, -2.0, 1.0] + [0.0] * 9) + rng.normal(0, 0.05, (B, nx))
    xg = np.array([2.0, 2.0, 4.0] + [0.0] * 9)
    Xref = np.stack([(1 - i / 49.0) * x0 + (i / 49.0) * xg for i in range(N)], axis=2)
    ax = np.zeros((1, nx)); ax[0, 2] = 1.0
    out = {}
    for name, force in (("regs", 0), ("cover", 1)):
        Bk = B if not force else B // 8
        s = tm.TinyBatchSolver.from_problem(prob, Bk)
        if tv:
            s.set_tv_linear_constraints(np.tile(ax, (N, 1)), np.linspace(1.1, 3.0, N).reshape(1, N), np.ones((N - 1, nu)), np.full((1, N - 1), 6.0))
            s.update_settings(max_iter=100, en_state_bound=0, en_input_bound=0, en_tv_state_linear=1, en_tv_input_linear=1)
        else:
            s.set_linear_constraints(ax, [3.0], np.ones((1, nu)), [6.0])
            s.update_settings(max_iter=100, en_state_bound=0, en_input_bound=0, en_state_linear=1, en_input_linear=1)
        s.set_option("force_general", force)
        s.set_x_ref(Xref[:Bk])
        ms = None
        for _ in range(2):                       # the first launch of a kernel also loads its code object
            s.reset()
            s.set_x0(x0[:Bk])
            s.set_option("timing", 1)
            s.solve_async()
            t_ = float(s.timing_ms()[0])
            ms = t_ if ms is None else min(ms, t_)
        st = s.reduce_stats()
        out[name] = dict(kernel=s.kernel_path(), batch=Bk, kernel_ms=ms, solves_per_s=Bk / (ms * 1e-3),
                         admm_iters_per_s=st[0] / (ms * 1e-3), iters_per_solve=st[0] / Bk)
        s.close()
    return dict(config=f"quadrotor {'time-varying ' if tv else ''}linear constraints example, one cold solve", **out)


if __name__ == "__main__":
    runs = {"config3": config3, "config4": config4, "linear_static": lambda: linear_example(tv=False),
            "linear_tv": lambda: linear_example(tv=True)}
    pick = sys.argv[2].split(",") if len(sys.argv) > 2 else list(runs)
    res = {k: runs[k]() for k in pick}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)
