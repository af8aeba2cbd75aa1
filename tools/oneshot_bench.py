"""GPU: one_shot launches on wide / long shapes, the fast box form (round 6) against the all-in-registers form (round 5)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm
B = 65536
print("| cell | plain cold solve (ms) | one_shot = 2, all-in-registers form | one_shot = 2, fast box form | last_tile_form |")
print("|---|---|---|---|---|")
for nx, nu, N in [(20, 8, 10), (12, 8, 30), (20, 4, 30), (20, 8, 30), (8, 4, 50), (12, 2, 50), (20, 4, 50), (20, 8, 50)]:
    prob, rng = tm.random_problem(nx, nu, N)
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=500)
    s.set_x0(rng.uniform(-1, 1, (B, nx)))
    s.set_x_ref(np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1)), N, axis=2))
    def run(n=3):
        best = None
        for _ in range(n):
            s.reset()
            s.set_option("timing", 1)
            s.solve_async()
            ms = float(np.sum(s.timing_ms()))
            best = ms if best is None else min(best, ms)
        return best
    plain = run()
    it0 = s.status()["iter"].copy(); u0 = s.get("u").copy()
    s.set_option("one_shot", 2)
    s.set_option("one_shot_fast", 0)
    slow = run()
    assert np.array_equal(s.status()["iter"], it0) and np.array_equal(s.get("u"), u0)
    s.set_option("one_shot_fast", 1)
    fast = run()
    assert np.array_equal(s.status()["iter"], it0) and np.array_equal(s.get("u"), u0)
    print("| (%d,%d,%d) | %.2f | %.2f | %.2f | %d |" % (nx, nu, N, plain, slow, fast, s.get_option("last_tile_form")), flush=True)
    s.close()
