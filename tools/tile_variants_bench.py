#!/usr/bin/env python3
"""What VERDICT r03 ("missing" item 5) said had never been timed: wide / long shapes WITH a cone or half-spaces.  Those run on
run-time instantiated variants of the tile kernel (all arrays in registers, trajectory in LDS, one wave per SIMD; tile_kernel.hip.h SOC /
LIN) or, where that cannot be built, on the coverage kernel.  Per shape: box only (the compiled-in form), + an input cone, + static
half-spaces, + both; 32 768 instances (BATCH), one cold solve, max_iter 200; median of 3 settled repetitions; FP64 fraction on box FLOPs.
Environment: SHAPES="12,4,10;6,3,10" (other shapes, e.g. the one-row kernel's: profiles/r04_onerow_variants_bench.md), ONLY=<substring of
the constraint column>, TILE_R=<rows along the horizon of the variant form> (experiments), TINYMPC_AMD_JIT_DEFINES (jit.hip)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm

B = int(os.environ.get("BATCH", "32768"))
print("| shape | constraints | kernel | ms (median) | ADMM it/s | it/solve | solved | FP64 frac (box FLOPs) |")
print("|---|---|---|---|---|---|---|---|")
SHAPES = [tuple(int(v) for v in c.split(",")) for c in os.environ["SHAPES"].split(";")] if os.environ.get("SHAPES") else [(12, 8, 10), (12, 8, 30), (20, 8, 30), (12, 4, 50)]
for (nx, nu, N) in SHAPES:
    prob, rng = tm.random_problem(nx, nu, N)
    x0 = rng.uniform(-1, 1, (B, nx)); xr = np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1)), N, axis=2)
    Ax = rng.standard_normal((2, nx)); bx = np.full(2, 2.0)
    Au = rng.standard_normal((2, nu)); bu = np.full(2, 0.4)
    for name, cone, lin in (("box", 0, 0), ("box + input cone", 1, 0), ("box + 2 + 2 half-spaces", 0, 1), ("box + cone + half-spaces", 1, 1)):
        if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
            continue
        s = tm.TinyBatchSolver.from_problem(prob, B)
        s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
        if cone:
            s.set_cone_constraints([], [], [], [0], [3], [0.5])
        if lin:
            s.set_linear_constraints(Ax, bx, Au, bu)
        s.update_settings(max_iter=200, en_input_soc=cone, en_state_linear=lin, en_input_linear=lin)
        if os.environ.get("TILE_R") and (cone or lin):
            s.set_option("tile_r", int(os.environ["TILE_R"]))            # experiments: rows along the horizon of the variant form
        s.set_x0(x0); s.set_x_ref(xr)
        ms = []
        try:
            for _ in range(5):
                s.reset(); s.set_option("timing", 1); s.solve_async(); ms.append(float(np.sum(s.timing_ms())))
            st = s.reduce_stats()
            m = float(np.median(ms[2:]))
            fl = tm.flops_per_iter(nx, nu, N)
            print(f"| ({nx},{nu},{N}) | {name} | {s.kernel_path()} | {m:.3f} | {st[0]/m*1e3:.3e} | {st[0]/B:.1f} | {st[1]/B:.3f} | {st[0]*fl/(m*1e-3)/78.6e12:.3f} |", flush=True)
        except Exception as e:                            # noqa: BLE001
            print(f"| ({nx},{nu},{N}) | {name} | failed: {e!r} | | | | | |", flush=True)
        s.close()
print("\nrun-time instantiated:", *tm.jit_used(), sep="\n  ", file=sys.stderr)
