"""GPU, round 6: BASELINE config 3 (262 144 cold tracking solves, max_iter 100) under the split solve's knobs, one at a time around the
settled form (K = 10, growth 4, 8 follow-up waves per CU, dynamic tile counter, PREFETCH first stage with 75 % static tiles): is anything
left on the table that the cost model / the plan does not already take?   python tools/experiments/config3_knobs.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tinympc_amd as tm
B = 262144
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
s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
s.update_settings(max_iter=100)
s.set_x_ref(Xref); s.set_u_ref(Uref); s.set_x0(x0)
s.set_option("plan", 0)
base = dict(repack_after=10, repack_growth=4, repack_waves_per_cu=8, repack_dynamic=1, prefetch_static=75, repack_sort=-1, prefetch=-1)
def run(opts, n=8):
    for kk, v in dict(base, **opts).items():
        s.set_option(kk, v)
    ms = []
    for _ in range(n):
        s.reset(); s.set_option("timing", 1); s.solve_async(); ms.append(float(np.sum(s.timing_ms())))
    return float(np.median(ms[2:])), float(np.min(ms[2:]))
print("| variation | ms median | ms min |\n|---|---|---|")
if os.environ.get("ROUND2"):
    vars_ = [dict(), dict(prefetch_static=50), dict(prefetch_static=25), dict(prefetch_static=40), dict(prefetch_static=60), dict(prefetch_static=50, repack_sort=0),
             dict(prefetch_static=50, repack_after=11), dict(prefetch_static=50, repack_dynamic=0), dict(prefetch_static=50, repack_waves_per_cu=4),
             dict(prefetch_static=50, repack_growth=3), dict(prefetch_static=50, repack_sort=0, repack_dynamic=0), dict(repack_sort=0)] * 2
    for v in vars_:
        med, mn = run(v, 16)
        print("| %s | %.4f | %.4f |" % (", ".join("%s=%s" % kv for kv in v.items()) or "the settled form", med, mn), flush=True)
    s.close()
    sys.exit(0)
vars_ = [dict()] + [dict(repack_after=v) for v in (8, 9, 11, 12, 14, 16, 20)] + [dict(repack_growth=v) for v in (2, 3, 8, 16)] + \
        [dict(repack_waves_per_cu=v) for v in (2, 4, 16, 32)] + [dict(repack_dynamic=0)] + [dict(prefetch_static=v) for v in (0, 50, 90, 100)] + \
        [dict(repack_sort=0), dict(repack_sort=1), dict(prefetch=0), dict(repack_after=9, repack_growth=8), dict(repack_after=12, repack_growth=8), dict(repack_after=10, repack_growth=16),
         dict(repack_after=10, repack_growth=4, repack_waves_per_cu=16, repack_sort=1), dict(repack_after=0), dict()]
for v in vars_:
    med, mn = run(v)
    print("| %s | %.4f | %.4f |" % (", ".join("%s=%s" % kv for kv in v.items()) or "the settled form", med, mn), flush=True)
s.close()
