#!/usr/bin/env python3
"""The one-row kernel's PREFETCH form (option "prefetch") against the plain form, and under its own options: the warm regime (one
launch per MPC step, steps 70-99 of the hover episode: 1-2 iterations per solve) at 65 536 and 262 144 instances with the shared and
with per-instance reference records, and BASELINE config 3 (262 144 cold tracking solves, automatic split).
    python tools/prefetch_probe.py [warm] [config3] > profiles/rNN_prefetch_probe.md
OPTS="prefetch_static=50;prefetch_vz=1" adds rows with those options on top of prefetch = -1."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm  # noqa: E402

what = [a for a in sys.argv[1:] if not a.startswith("-")] or ["warm", "config3"]
extra_sets = [dict(kv.split("=") for kv in grp.split(",")) for grp in os.environ.get("OPTS", "").split(";") if grp]
SETS = [{"prefetch": 0}, {"prefetch": -1}] + [dict({"prefetch": -1}, **{k: int(v) for k, v in e.items()}) for e in extra_sets]
prob, extra = tm.load_problem("quadrotor_20hz")
nx, nu, N = prob["nx"], prob["nu"], prob["N"]


def tag(o):
    return ", ".join("%s=%s" % kv for kv in o.items())


if "warm" in what:
    h = extra["hover"]
    print("## warm regime: one launch per MPC step, steps 70-99 of the hover episode (us per launch: mean / min over the 30 steps, median of 3 episodes)\n")
    print("| batch | reference records | options | us mean | us min | last_prefetch |")
    print("|---|---|---|---|---|---|")
    for B in [int(v) for v in os.environ.get("BATCHES", "65536,262144").split(",")]:
        for share in (1, 0):
            for o in SETS:
                if "kpi_skew" in o:                       # (not an option of the handle: where tiny_batch_setup puts the record arrays)
                    o = dict(o)
                    os.environ["TINYMPC_KPI_SKEW"] = str(o.pop("kpi_skew"))
                else:
                    os.environ.pop("TINYMPC_KPI_SKEW", None)
                s = tm.TinyBatchSolver.from_problem(prob, B)
                s.set_bound_constraints(np.full((nx, 1), h["x_min"]), np.full((nx, 1), h["x_max"]), np.full((nu, 1), h["u_min"]), np.full((nu, 1), h["u_max"]))
                s.update_settings(max_iter=h["max_iter"])
                s.set_option("advance_x0", 1)
                s.set_option("share_ref", share)
                for k, v in o.items():
                    s.set_option(k, v)
                runs = []
                for _ in range(3):
                    s.reset()
                    s.set_x_ref(np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N)), broadcast=True)
                    s.set_x0(np.array(h["x0"], dtype=np.float64), broadcast=True)
                    s.set_option("timing", 100)
                    for _ in range(100):
                        s.solve_async()
                    s.synchronize()
                    runs.append(s.timing_ms())
                ms = np.median(np.array(runs), axis=0)[70:100]
                print(f"| {B} | {'shared' if share else 'per instance'} | {tag(o)}{' skew ' + os.environ['TINYMPC_KPI_SKEW'] if 'TINYMPC_KPI_SKEW' in os.environ else ''} | {ms.mean() * 1e3:.1f} | {ms.min() * 1e3:.1f} | {s.get_option('last_prefetch')} grid {s.get_option('last_prefetch_grid')} lds {s.get_option('last_prefetch_lds')} |", flush=True)
                s.close()

if "config3" in what:
    B = 262144
    traj = np.array(extra["y_axis_line"])
    rng = np.random.default_rng(20260923)
    k = rng.integers(0, 291, B)
    Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
    Uref = rng.normal(0, 0.05, (B, nu, N - 1))
    x0 = Xref[:, :, 0].copy()
    x0[:, :3] += rng.normal(0, 0.1, (B, 3))
    print("\n## BASELINE config 3: 262 144 cold tracking solves (ms per solve call: median / min of the settled repetitions)\n")
    print("| options | repack_after | ms median | ms min | split K | verdict | last_prefetch |")
    print("|---|---|---|---|---|---|---|")
    for o in SETS:
        for ra in (0, -1):
            s = tm.TinyBatchSolver.from_problem(prob, B)
            s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
            s.update_settings(max_iter=100)
            s.set_x_ref(Xref); s.set_u_ref(Uref); s.set_x0(x0)
            s.set_option("repack_after", ra)
            for kk, v in o.items():
                s.set_option(kk, v)
            ms = []
            for _ in range(5 if ra == 0 else 14):
                s.reset()
                s.set_option("timing", 1)
                s.solve_async()
                ms.append(float(np.sum(s.timing_ms())))
            settled = ms[2:] if ra == 0 else ms[7:]
            print(f"| {tag(o)} | {ra} | {np.median(settled):.4f} | {np.min(settled):.4f} | {s.get_option('auto_split_k')} | {s.get_option('auto_split_verdict')} | {s.get_option('last_prefetch')} |", flush=True)
            s.close()
