"""GPU, round 6: the library's DEFAULT dispatch of the one-row config-5 cells (shipped plan: plain / split / K / stage schedule / tile
alternative as settled) against the same with option "repack_tail" = 1 (a split solve's tail on the tile kernel's dynamic form).
Interleaved repetitions, median kernel ms, bit-identity.   python tools/experiments/repack_tail_default_ab.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tinympc_amd as tm


def timed(s, n):
    ms = []
    for _ in range(n):
        s.reset(); s.set_option("timing", 1); s.solve_async(); ms.append(float(np.sum(s.timing_ms())))
    return ms


print("| cell | default (ms) | default + repack_tail = 1 (ms) | change | split K | tail form taken | tile alternative verdict | bit-identical |\n|---|---|---|---|---|---|---|---|")
tot = [0.0, 0.0]
for N in (10, 30):
    for nx in (4, 8, 12):
        for nu in (2, 4, 8):
            if nx + nu > 16:
                continue
            prob, rng = tm.random_problem(nx, nu, N)
            B = 131072
            s = tm.TinyBatchSolver.from_problem(prob, B)
            if s.kernel_path() != "regs":
                s.close(); continue
            s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
            s.update_settings(max_iter=500)
            s.set_x0(rng.uniform(-1, 1, (B, nx)))
            s.set_x_ref(np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1)), N, axis=2))
            timed(s, 3)
            t = {0: [], 1: []}
            out = {}
            for r in range(3):
                for tail in (0, 1):
                    s.set_option("repack_tail", tail)
                    t[tail] += timed(s, 4)[1:]
                    if r == 0:
                        st = s.status()
                        out[tail] = (st["iter"].copy(), s.get("u"), s.get("g"), s.get("v"), int(s.get_option("last_tail_tile")))
            same = all(np.array_equal(a, b) for a, b in zip(out[0][:4], out[1][:4]))
            m0, m1 = float(np.median(t[0])), float(np.median(t[1]))
            tot[0] += m0; tot[1] += min(m0, m1)
            print("| (%d,%d,%d) | %.3f | %.3f | %+.1f %% | %d | %d | %d | %s |" % (nx, nu, N, m0, m1, 100 * (m1 / m0 - 1), s.get_option("auto_split_k"), out[1][4],
                                                                           s.get_option("tile_alt_verdict"), same), flush=True)
            s.close()
print("sum of the cells: default %.1f ms, best of both per cell %.1f ms" % (tot[0], tot[1]))
