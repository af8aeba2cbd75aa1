#!/usr/bin/env python3
"""GPU: times every compiled-in FORM of the tile kernel (tile_dims.txt: rows along the horizon R, LDS-offload set LM; static tiles or
the dynamic slot form) on the config-5 sweep cells it serves, one cold solve of 131 072 instances each (the recipe of
tools/sweep_bench.py), and prints a markdown table with the fastest form per cell -- the order of tile_dims.txt follows it.
    python tools/tile_forms.py [--batch 131072] [--reps 2] [--cells "4,2,50;..."] > profiles/r03_tile_forms.md"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def forms():
    out = {}
    for line in open(os.path.join(ROOT, "tinympc_amd", "csrc", "tile_dims.txt")):
        f = line.split("#")[0].split()
        if len(f) >= 5:
            nx, nu, N, W, R = map(int, f[:5])
            lm = int(f[5]) if len(f) >= 6 else 99
            if (W, R, lm) not in out.setdefault((nx, nu, N), []):
                out[(nx, nu, N)].append((W, R, lm))
    return out


def time_form(nx, nu, N, B, reps, opts):
    prob, rng = tm.random_problem(nx, nu, N)
    s = tm.TinyBatchSolver.from_problem(prob, B)
    for k, v in opts.items():
        s.set_option(k, v)
    s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=500)
    s.set_x0(rng.uniform(-1, 1, (B, nx)))
    s.set_x_ref(np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1)), N, axis=2))
    best = None
    for _ in range(reps + 1):
        s.reset()
        s.set_option("timing", 1)
        s.solve_async()
        ms = float(s.timing_ms()[0])
        best = ms if best is None else min(best, ms)
    st = s.reduce_stats()
    path = s.kernel_path()
    s.close()
    return best, st[0], path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=131072)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--cells", default="")
    args = ap.parse_args()
    fm = forms()
    cells = [tuple(int(v) for v in c.split(",")) for c in args.cells.split(";")] if args.cells else sorted(fm, key=lambda c: (c[2], c[0], c[1]))
    print("| cell | form (R, LM, grid) | ms | ADMM it/s | FP64 frac | |")
    print("|---|---|---|---|---|---|")
    for cell in cells:
        nx, nu, N = cell
        rows = []
        if nx + nu <= 16 and N <= 30:
            ms, iters, path = time_form(nx, nu, N, args.batch, args.reps + 2, {})
            rows.append(("one-row kernel (automatic split)", ms, iters, path))
        for W, R, lm in fm.get(cell, []):
            for dyn in (0, 1):
                o = {"prefer_tile": 1, "tile_w": W, "tile_r": R, "tile_lm": lm, "tile_dyn": dyn}       # (LM 99 = the entry without an LM column)
                ms, iters, path = time_form(nx, nu, N, args.batch, args.reps, o)
                rows.append(("%sR=%d LM=%s %s" % ("half rows " if W == 0 else "", R, "auto" if lm == 99 else lm, "dynamic" if dyn else "static"), ms, iters, path))
        best = min(r[1] for r in rows)
        for name, ms, iters, path in rows:
            fl = tm.flops_per_iter(nx, nu, N)
            print("| (%d,%d,%d) | %s [%s] | %.3f | %.3e | %.3f | %s |" % (nx, nu, N, name, path, ms, iters / ms * 1e3, iters * fl / (ms * 1e-3) / 78.6e12,
                                                                       "<- fastest" if ms == best else ""), flush=True)


if __name__ == "__main__":
    main()
