#!/usr/bin/env python3
"""CPU only: compile a cross product of template-argument combinations of both kernels through the library's own hipRTC path
(tiny_jit_compile) -- cone x debug x half-space family x heterogeneous x table stride on three one-row shapes, cone x
half-space family x stride on four tile shapes -- so that a header edit which breaks a rarely used combination shows up
without a GPU.  Expected failures: only time-varying tables that exceed the LDS (the host never asks for those).
    python tools/jit_combo_check.py"""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm
names = []
b = lambda v: "true" if v else "false"
for (nx, nu, N) in ((5, 3, 7), (12, 4, 10), (8, 4, 30)):
    for soc, dbg, lin, het in itertools.product((0, 1), (0, 1), (0, 1, 2, 3), (0, 1)):
        for kmax in ((4,) if lin == 0 else (4, 8, 32)):
            mode = 2
            names.append(f"tinympc_amd::admm_solve_kernel<{nx}, {nu}, {N}, {b(soc)}, {b(dbg)}, {mode}, {lin}, {b(het)}, {kmax}>")
for (nx, nu, N, W, R) in ((20, 4, 10, 2, 1), (6, 2, 60, 1, 2), (12, 8, 30, 2, 2), (16, 8, 6, 2, 1)):
    for soc, lin in itertools.product((0, 1, 2, 3), (0, 1, 2, 3)):      # (tile SOC: bit 0 input family, bit 1 state family)
        for kmax in ((4,) if lin == 0 else (4, 16)):
            names.append(f"tinympc_amd::admm_tile_kernel<{nx}, {nu}, {N}, {W}, {R}, {soc}, {lin}, {kmax}>")
print(len(names), "instantiations", flush=True)
bad = 0
t0 = time.time()
for i, n in enumerate(names):
    try:
        tm.jit_compile(n)
    except RuntimeError as e:
        bad += 1
        print("FAILED", n, str(e)[:300], flush=True)
    if i % 20 == 19:
        print(i + 1, "done", round(time.time() - t0), "s", flush=True)
print("failures:", bad)
