#!/usr/bin/env python3
"""BASELINE configs[4]: the random (nx, nu, N) sweep -- one cold batched solve per cell, per-instance random
x0 / Xref, max_iter 500, u in [-0.5, 0.5].  Prints a markdown roofline table and writes JSON.

    python tools/sweep_bench.py --batch 131072 --out profiles/r01_sweep.json

Which kernel serves a cell is reported: "regs" = one-row register-resident kernel (admm_kernel.hip.h), "tile" =
W x R-row tile kernel (tile_kernel.hip.h), "cover" = coverage kernel (general_kernel.hip.h).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm  # noqa: E402


def run_cell(nx, nu, N, B, reps):
    prob, rng = tm.random_problem(nx, nu, N)
    s = tm.TinyBatchSolver.from_problem(prob, B)
    for kv in filter(None, os.environ.get("TINYMPC_OPTS", "").split(",")):       # experiments: prefer_tile=1 ...
        s.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=500)
    x0 = rng.uniform(-1, 1, (B, nx))
    xr = np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1)), N, axis=2)
    best = None
    for _ in range(reps + 1):
        s.reset()
        s.set_x0(x0)
        s.set_x_ref(xr)
        s.set_option("timing", 1)
        s.solve_async()
        ms = float(s.timing_ms()[0])
        best = ms if best is None else min(best, ms)
    path = s.kernel_path()
    st = s.reduce_stats()
    iters, solved = st[0], st[1]
    alg = s.algorithmic_bytes()
    s.close()
    fl = tm.flops_per_iter(nx, nu, N)
    t = best * 1e-3
    return dict(nx=nx, nu=nu, N=N, batch=B, kernel=path,
                ms=best, solves_per_s=B / t, iters_per_s=iters / t, iters_per_solve=iters / B, solved_fraction=solved / B,
                fp64_tflops=iters * fl / t / 1e12, fp64_frac=iters * fl / t / 78.6e12,
                hbm_gbs=alg * B / t / 1e9, hbm_frac=alg * B / t / 8e12, bytes_per_solve=alg, flops_per_iter=fl)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=131072)
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--cells", default="")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    cells = ([tuple(int(v) for v in c.split(",")) for c in args.cells.split(";")] if args.cells else
             [(nx, nu, N) for nx in (4, 8, 12, 20) for nu in (2, 4, 8) for N in (10, 30, 50)])
    rows = []
    print("| nx | nu | N | kernel | ms | solves/s | ADMM it/s | it/solve | solved | FP64 frac | HBM frac |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for nx, nu, N in cells:
        r = run_cell(nx, nu, N, args.batch, args.reps)
        rows.append(r)
        print(f"| {nx} | {nu} | {N} | {r['kernel']} | {r['ms']:.3f} | {r['solves_per_s']:.3e} | {r['iters_per_s']:.3e} | "
              f"{r['iters_per_solve']:.1f} | {r['solved_fraction']:.3f} | {r['fp64_frac']:.3f} | {r['hbm_frac']:.4f} |", flush=True)
    if args.out:
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
