#!/usr/bin/env python3
"""The one-row kernel's HALF form (nx+nu <= 8: two instances per DPP row) against its one-instance-per-row form and the library's default
dispatch, on the config-5 cells it applies to: 131 072 instances, one cold solve, max_iter 500; median of the settled repetitions."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm
import torch

B = int(os.environ.get("BATCH", "131072"))
print("| cell | form | ms (median) | min | ADMM it/s | FP64 frac | it/solve |")
print("|---|---|---|---|---|---|---|")
for (nx, nu, N) in ((4, 2, 10), (4, 4, 10), (4, 2, 30), (4, 4, 30)):
    prob, rng = tm.random_problem(nx, nu, N)
    x0 = rng.uniform(-1, 1, (B, nx)); xr = rng.uniform(-0.2, 0.2, (B, nx, 1))
    for name, opts in (("one row, plain", dict(no_tile=1, repack_after=0, half_rows=0)), ("HALF rows, plain", dict(no_tile=1, repack_after=0, half_rows=1)),
                       ("one row, default dispatch", dict(half_rows=0)), ("HALF rows, default dispatch", dict())):
        s = tm.TinyBatchSolver.from_problem(prob, B)
        s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
        s.update_settings(max_iter=500)
        for k, v in opts.items():
            s.set_option(k, v)
        s.set_x0(x0)
        xr_d = torch.from_numpy(np.ascontiguousarray(xr[:, :, 0])).cuda()[:, None, :].expand(B, N, nx).contiguous()
        torch.cuda.synchronize()
        s.set_device("Xref", xr_d.data_ptr()); s.synchronize()
        ms = []
        for _ in range(11):
            s.reset(); s.set_option("timing", 1); s.solve_async(); ms.append(float(np.sum(s.timing_ms())))
        st = s.reduce_stats()
        m = float(np.median(ms[6:]))
        fl = tm.flops_per_iter(nx, nu, N)
        print(f"| ({nx},{nu},{N}) | {name} ({s.kernel_path()}, half={s.get_option('last_half_rows')}) | {m:.3f} | {min(ms[6:]):.3f} | {st[0]/m*1e3:.3e} | {st[0]*fl/(m*1e-3)/78.6e12:.3f} | {st[0]/B:.1f} |", flush=True)
        s.close()
