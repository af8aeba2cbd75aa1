"""GPU, round 6: a split solve's tail on the tile kernel's dynamic slot form (option "repack_tail" = 1) against the follow-up stages
(= 0): BASELINE config 3 and a few one-row sweep cells; bit-identity of every record and kernel time (median of interleaved
repetitions).   python tools/experiments/repack_tail_ab.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tinympc_amd as tm

FIELDS = ("x", "u", "vnew", "znew", "g", "y", "v", "z")


def timed(s, n):
    ms = []
    for _ in range(n):
        s.reset(); s.set_option("timing", 1); s.solve_async(); ms.append(float(np.sum(s.timing_ms())))
    return ms


def ab(name, s, K, growth=None, reps=4):
    s.set_option("plan", 0)
    s.set_option("repack_after", K)
    if growth:
        s.set_option("repack_growth", growth)
    out = {}
    times = {0: [], 1: []}
    for r in range(reps):
        for tail in (0, 1):
            s.set_option("repack_tail", tail)
            times[tail] += timed(s, 5)[1:]
            if r == 0:
                st = s.status()
                out[tail] = dict(iter=st["iter"].copy(), solved=st["solved"].copy(), took=int(s.get_option("last_tail_tile")), stats=s.reduce_stats().copy(),
                                 **{f: s.get(f) for f in FIELDS})
    same = all(np.array_equal(out[0][k], out[1][k]) for k in ("iter", "solved") + FIELDS) and np.array_equal(out[0]["stats"][:2], out[1]["stats"][:2])
    print("| %s | K=%d | %.4f | %.4f | %+.1f %% | tail form taken: %d | bit-identical: %s |" % (name, K, np.median(times[0]), np.median(times[1]),
          100 * (np.median(times[1]) / np.median(times[0]) - 1), out[1]["took"], same), flush=True)


print("| workload | first stage | follow-up stages (ms) | tile tail (ms) | change | | |\n|---|---|---|---|---|---|---|")
B = 262144
prob, extra = tm.load_problem("quadrotor_20hz")
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
traj = np.array(extra["y_axis_line"])
rng = np.random.default_rng(20260923)
k = rng.integers(0, 291, B)
Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
Uref = rng.normal(0, 0.05, (B, nu, N - 1))
x0 = Xref[:, :, 0].copy(); x0[:, :3] += rng.normal(0, 0.1, (B, 3))
s = tm.TinyBatchSolver.from_problem(prob, B)
s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
s.update_settings(max_iter=100)
s.set_x_ref(Xref); s.set_u_ref(Uref); s.set_x0(x0)
for K in (10, 9, 11):
    ab("config 3", s, K, 4)
s.close()
for (nx, nu, N), K in (((12, 4, 10), 0), ((8, 4, 10), 0), ((4, 4, 10), 0), ((12, 2, 10), 0), ((4, 2, 30), 0), ((12, 4, 30), 0), ((8, 2, 30), 0)):
    prob, rng = tm.random_problem(nx, nu, N)
    B = 131072
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=500)
    s.set_x0(rng.uniform(-1, 1, (B, nx)))
    s.set_x_ref(np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1)), N, axis=2))
    if K == 0:                                           # the K the cost model picks for this batch
        s.set_option("plan", 0)
        timed(s, 3)
        K = int(s.get_option("auto_split_k")) or 8
    ab("sweep (%d,%d,%d)" % (nx, nu, N), s, K)
    s.close()
