#!/usr/bin/env python3
"""BASELINE configs[3] (tools/bench_configs.py config4: 65 536 rocket landings, 90 closed-loop MPC steps fused into one solve call)
under option "step_regroup": the launch cut into stretches of K MPC steps, every stretch over the instances ordered by the
iteration count of their last solve.  ms per episode (median of 5, min) per K; K = 0: the uncut launch; -1: automatic.
    python tools/regroup_bench.py [--cones input|state|both] [--ks 0,10,15,23,30,45,-1] [--batch 65536]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tinympc_amd as tm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cones", default="input")
    ap.add_argument("--ks", default="0,10,15,23,30,45,-1,0")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--streams", type=int, default=2)
    a = ap.parse_args()
    ess, eis = {"input": (0, 1), "state": (1, 0), "both": (1, 1)}[a.cones]
    B = a.batch
    prob, extra = tm.load_problem("rocket_landing_20hz")
    m = extra["mpc"]
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    rng = np.random.default_rng(20260923)
    x0 = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, (B, nx)))
    xinit, xg = np.array(m["xinit"], dtype=float), np.array(m["xg"], dtype=float)
    traj = np.stack([xinit + (xg - xinit) * float(i) / (m["NTOTAL"] - 1) for i in range(m["NTOTAL"])])
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.array(m["x_min"]), np.array(m["x_max"]), np.full((nu, 1), m["u_min"]), np.full((nu, 1), m["u_max"]))
    s.set_cone_constraints(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"],
                           m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])
    s.update_settings(abs_pri_tol=m["abs_pri_tol"], max_iter=m["max_iter"], en_state_soc=ess, en_input_soc=eis)
    uref = np.zeros((nu, N - 1)); uref[2, :] = m["uref_z"]
    steps = m["NTOTAL"] - N
    s.set_option("advance_x0", 1)
    s.set_option("steps_per_launch", steps)
    s.set_option("step_regroup_streams", a.streams)

    def episode():
        s.reset()
        s.set_u_ref(uref, broadcast=True)
        s.set_reference_trajectory(traj)
        s.set_x0(x0)
        s.set_option("timing", 1)
        s.solve_async()
        return float(np.sum(s.timing_ms()))
    print("streams = %d" % a.streams)
    print("| cones | K | stretches | ms median | ms min | ADMM it/s | lock-step estimate |")
    print("|---|---|---|---|---|---|---|")
    ref = None
    for K in [int(k) for k in a.ks.split(",")]:
        s.set_option("step_regroup", K)
        ms = [episode() for _ in range(a.reps + (2 if K < 0 else 1))][(2 if K < 0 else 1):]     # (automatic: its first episode is the uncut one that decides)
        st = s.reduce_stats()
        if ref is None:
            ref = st[7]
        assert st[7] == ref, "the stretches changed the iteration total"
        print("| %s | %d | %d | %.3f | %.3f | %.3e | %.3f |" % (a.cones, K, s.get_option("step_regroup_stretches"), np.median(ms), min(ms),
                                                               st[7] / (np.median(ms) * 1e-3), s.get_option("lockstep_permille") / 1000.0), flush=True)
    s.close()


if __name__ == "__main__":
    main()
