#!/usr/bin/env python3
"""Writes tinympc_amd/data/plans.txt: the settled TinyBatchPlans (tiny_batch_get_plan) of the BASELINE shapes, which the library imports
into a fresh handle of the same shape / settings / batch bucket at its first solve (batch_api.hip shipped_plans; option "plan" = 0:
off) -- so that a caller's FIRST solve already takes the launch form the probes of an earlier process settled on.  A plan is advice
about launch forms only: every form leaves the same bits.

    python tools/make_plans.py [--out tinympc_amd/data/plans.txt]          (needs the GPU; ~1 minute)

Entries: BASELINE config 3 (quadrotor tracking x 262 144, max_iter 100), config 4 with its three cone settings (rocket landing x
65 536, 90 fused steps: the stretch verdict), every config-5 sweep cell the one-row kernel holds (x 131 072, max_iter 500)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["TINYMPC_AMD_PLANS"] = "0"                  # the handles below settle by themselves
import tinympc_amd as tm  # noqa: E402


def line(plan: bytes, note: str, soc=None) -> str:
    f = tm.TinyBatchSolver.plan_fields(plan)
    import struct
    hist = struct.unpack_from("<1024I", plan, len(plan) - 4096)
    nz = [(i, c) for i, c in enumerate(hist) if c] if f["hist_valid"] else []
    head = ("plan" if soc is None else "plan_soc %d" % soc) + " %d %d %d %d %d %d %d %d %d %d %d %d %d %d %.9g %.9g %.9g %.9g %.9g %d" % (
        f["nx"], f["nu"], f["N"], f["batch"], f["max_iter"], f["check_termination"], f["auto_verdict"], f["auto_cap"], f["auto_cap_max_iter"],
        f["auto_growth"], f["growth_verdict"], f["auto_probes"], f["tile_verdict"], f["regroup_verdict"], f["auto_plain_rate"], f["auto_split_rate"],
        f["auto_gain"], f["tile_rate"], f["lockstep_ratio"], len(nz))
    return "# %s\n%s %s\n" % (note, head, " ".join("%d:%d" % p for p in nz))


def settle_cold(s, n):
    for _ in range(n):                                 # (the verdicts are read when the NEXT solve is decided, from events that must have
        s.reset()                                      # arrived by then: wait after every solve, as a caller that reads its results does)
        s.solve_async()
        s.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tinympc_amd", "data", "plans.txt"))
    args = ap.parse_args()
    import time
    t_start = time.perf_counter()

    def flush():                                       # (the file grows plan by plan: a run that is cut short leaves what it has)
        with open(args.out, "w") as f:
            f.write("".join(out))
        print("%.1f s: %d plans" % (time.perf_counter() - t_start, len(out) - 1), file=sys.stderr, flush=True)
    out = ["# tinympc_amd/data/plans.txt -- written by tools/make_plans.py (settled TinyBatchPlans of the BASELINE shapes; format: batch_api.hip shipped_plans)\n"]
    # config 3
    prob, extra = tm.load_problem("quadrotor_20hz")
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    B = 262144
    traj = np.array(extra["y_axis_line"])
    rng = np.random.default_rng(20260923)
    k = rng.integers(0, 291, B)
    Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
    Uref = rng.normal(0, 0.05, (B, nu, N - 1))
    x0 = Xref[:, :, 0].copy()
    x0[:, :3] += rng.normal(0, 0.1, (B, 3))
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=100)
    s.set_x_ref(Xref); s.set_u_ref(Uref); s.set_x0(x0)
    settle_cold(s, 17)
    out.append(line(s.get_plan(), "BASELINE config 3: quadrotor tracking x 262 144, per-instance references, one cold solve"))
    s.close()
    flush()
    # config 4, three cone settings: the stretch verdict of the fused episode
    for ss, si in ((0, 1), (1, 0), (1, 1)):
        prob, extra = tm.load_problem("rocket_landing_20hz")
        m = extra["mpc"]
        nx, nu, N = prob["nx"], prob["nu"], prob["N"]
        B = 65536
        rng = np.random.default_rng(20260923)
        x0 = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, (B, nx)))
        xinit, xg = np.array(m["xinit"], dtype=float), np.array(m["xg"], dtype=float)
        trj = np.stack([xinit + (xg - xinit) * float(i) / (m["NTOTAL"] - 1) for i in range(m["NTOTAL"])])
        s = tm.TinyBatchSolver.from_problem(prob, B)
        s.set_bound_constraints(np.array(m["x_min"]), np.array(m["x_max"]), np.full((nu, 1), m["u_min"]), np.full((nu, 1), m["u_max"]))
        s.set_cone_constraints(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"], m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])
        s.update_settings(abs_pri_tol=m["abs_pri_tol"], max_iter=m["max_iter"], en_state_soc=ss, en_input_soc=si)
        uref = np.zeros((nu, N - 1)); uref[2, :] = m["uref_z"]
        s.set_option("advance_x0", 1)
        s.set_option("steps_per_launch", m["NTOTAL"] - N)
        for _ in range(3):
            s.reset()
            s.set_u_ref(uref, broadcast=True)
            s.set_reference_trajectory(trj)
            s.set_x0(x0)
            s.solve_async()
            s.synchronize()
        # (one entry per cone setting -- `plan_soc <mask>`, bit 0 inputs, bit 1 states: stretches pay with the thrust cone alone)
        out.append(line(s.get_plan(), "BASELINE config 4: rocket landing x 65 536, %s, 90 fused steps (the stretch verdict)" %
                        {(0, 1): "input cone", (1, 0): "state cone", (1, 1): "both cones"}[(ss, si)], soc=(1 if si else 0) | (2 if ss else 0)))
        s.close()
    flush()
    # the config-5 cells the one-row kernel holds
    for nx in (4, 8, 12):
        for nu in (2, 4, 8):
            for N in (10, 30):
                if nx + nu > 16:
                    continue
                prob, rng = tm.random_problem(nx, nu, N)
                B = 131072
                s = tm.TinyBatchSolver.from_problem(prob, B)
                if s.kernel_path() != "regs":
                    s.close()
                    continue
                s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
                s.update_settings(max_iter=500)
                s.set_x0(rng.uniform(-1, 1, (B, nx)))
                s.set_x_ref(np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1)), N, axis=2))
                settle_cold(s, 11)
                out.append(line(s.get_plan(), "BASELINE config 5 cell (%d,%d,%d) x 131 072, one cold solve, max_iter 500" % (nx, nu, N)))
                s.close()
                flush()
    with open(args.out, "w") as f:
        f.write("".join(out))
    print("wrote", args.out, len(out) - 1, "plans")


if __name__ == "__main__":
    main()
