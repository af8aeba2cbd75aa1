#!/usr/bin/env python3
"""Differential fuzzing of the reference-named entry points (tiny_setup ... tiny_solve on plain-data mirrors of the
reference structs) against the oracle: a SEQUENCE of solves on one TinySolver with settings, bounds, cones, half-spaces
and the workspace poked in between, exactly as a reference caller would (exercises the family-hash that decides
whether the device tables are re-uploaded).  python tools/fuzz_compat.py [n_trials] [seed]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pod, scenarios as sc, tinympc_amd as tm
from cpu_solvers import OracleSolver, build_oracle

SHAPES = [(4, 1, 10), (12, 4, 10), (6, 3, 10), (2, 2, 3), (4, 8, 10), (8, 4, 30), (12, 8, 10), (5, 3, 7), (20, 4, 10)]
FIELDS = ("x", "u", "q", "r", "p", "d", "v", "vnew", "z", "znew", "g", "y", "vcnew", "zcnew", "gc", "yc",
          "vlnew", "zlnew", "gl", "yl", "vlnew_tv", "zlnew_tv", "gl_tv", "yl_tv")
_keep = []


def P(t):
    return C.POINTER(t)


def proto(L):
    S = P(pod.TinySolver)
    L.tiny_setup.argtypes = [P(S)] + [P(pod.Mat)] * 5 + [C.c_double] + [C.c_int] * 4
    L.tiny_set_bound_constraints.argtypes = [S] + [P(pod.Mat)] * 4
    L.tiny_set_cone_constraints.argtypes = [S] + [P(pod.VecXi), P(pod.VecXi), P(pod.Vec)] * 2
    L.tiny_set_linear_constraints.argtypes = [S, P(pod.Mat), P(pod.Vec), P(pod.Mat), P(pod.Vec)]
    L.tiny_set_tv_linear_constraints.argtypes = [S] + [P(pod.Mat)] * 4
    L.tiny_solve.argtypes = [S]
    L.tiny_destroy.argtypes = [S]


def trial(seed):
    L = tm.lib()
    proto(L)
    rng = np.random.default_rng(seed)
    nx, nu, N = SHAPES[rng.integers(len(SHAPES))]
    M = rng.standard_normal((nx, nx))
    prob = dict(nx=nx, nu=nu, N=N, rho=float(rng.choice([0.5, 1.0, 5.0])), A=M * rng.uniform(0.5, 1.0) / np.max(np.abs(np.linalg.eigvals(M))),
                B=rng.standard_normal((nx, nu)) / np.sqrt(nx), f=rng.normal(0, 0.02, nx) * rng.integers(0, 2),
                Q=rng.uniform(0.5, 10, nx), R=rng.uniform(0.1, 2, nu))
    ms = [pod.mat(prob["A"]), pod.mat(prob["B"]), pod.mat(prob["f"]), pod.mat(np.diag(prob["Q"])), pod.mat(np.diag(prob["R"]))]
    sp = P(pod.TinySolver)()
    assert L.tiny_setup(C.byref(sp), *[C.byref(m[0]) for m in ms], prob["rho"], nx, nu, N, 0) == 0
    cfg = sc.default_config(prob)
    o = sc.make_solver(OracleSolver, prob, cfg)
    w, st = sp.contents.work.contents, sp.contents.settings.contents
    desc = f"seed {seed} shape {(nx, nu, N)}"
    try:
        for call in range(int(rng.integers(2, 6))):
            # ---- poke the problem family the way a caller would
            if call == 0 or rng.random() < 0.5:
                b = [rng.uniform(-2.0, -0.2, (nx, N)), rng.uniform(0.2, 2.0, (nx, N)), rng.uniform(-1.0, -0.1, (nu, N - 1)), rng.uniform(0.1, 1.0, (nu, N - 1))]
                pm = [pod.mat(a) for a in b]
                assert L.tiny_set_bound_constraints(sp, *[C.byref(m[0]) for m in pm]) == 0
                o.set_bounds(*b)
            if nx >= 3 and nu >= 3 and rng.random() < 0.3:
                cs = ([int(rng.integers(0, nx - 2))], [3], [float(rng.uniform(0.3, 1.2))], [int(rng.integers(0, nu - 2))], [3], [float(rng.uniform(0.3, 1.2))])
                pc = [pod.veci(cs[0]), pod.veci(cs[1]), pod.vec(cs[2]), pod.veci(cs[3]), pod.veci(cs[4]), pod.vec(cs[5])]
                assert L.tiny_set_cone_constraints(sp, *[C.byref(m[0]) for m in pc]) == 0
                o.set_cones(*cs)
                st.en_state_soc, st.en_input_soc = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            if rng.random() < 0.3:
                ns, ni = int(rng.integers(1, 4)), int(rng.integers(1, 3))
                ln = (rng.standard_normal((ns, nx)), rng.uniform(0.1, 1.0, ns), rng.standard_normal((ni, nu)), rng.uniform(0.05, 0.5, ni))
                pl = [pod.mat(ln[0]), pod.vec(ln[1]), pod.mat(ln[2]), pod.vec(ln[3])]
                assert L.tiny_set_linear_constraints(sp, *[C.byref(m[0]) for m in pl]) == 0
                o.set_linear(*ln)
                st.en_state_linear, st.en_input_linear = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            if rng.random() < 0.25:
                ns, ni = int(rng.integers(1, 3)), int(rng.integers(1, 3))
                tv = (rng.standard_normal((ns * N, nx)), rng.uniform(0.1, 1.0, (ns, N)), rng.standard_normal((ni * (N - 1), nu)), rng.uniform(0.05, 0.5, (ni, N - 1)))
                pt = [pod.mat(a) for a in tv]
                assert L.tiny_set_tv_linear_constraints(sp, *[C.byref(m[0]) for m in pt]) == 0
                o.set_tv_linear(*tv)
                st.en_tv_state_linear, st.en_tv_input_linear = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            st.max_iter = int(rng.integers(0, 35))
            st.check_termination = int(rng.integers(1, 4))
            st.abs_pri_tol, st.abs_dua_tol = float(10 ** rng.uniform(-4, -1)), float(10 ** rng.uniform(-4, -1))
            st.en_state_bound, st.en_input_bound = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            for k in ("max_iter", "check_termination", "abs_pri_tol", "abs_dua_tol", "en_state_bound", "en_input_bound", "en_state_soc",
                      "en_input_soc", "en_state_linear", "en_input_linear", "en_tv_state_linear", "en_tv_input_linear"):
                o.set(k, getattr(st, k))
            # ---- poke the workspace: new references / x0 every call, sometimes a random warm state
            xr, ur, x0 = rng.normal(0, 0.3, (nx, N)), rng.normal(0, 0.05, (nu, N - 1)), rng.uniform(-0.5, 0.5, nx)
            pod.to_np(w.Xref)[...] = xr; pod.to_np(w.Uref)[...] = ur; pod.to_np(w.x)[:, 0] = x0
            o["Xref"] = xr; o["Uref"] = ur; o["x"][:, 0] = x0
            if rng.random() < 0.3:
                for k in FIELDS:
                    a = pod.to_np(getattr(w, k))
                    a[...] = rng.normal(0, 0.3, a.shape)
                    o[k] = a
            rc, orc = L.tiny_solve(sp), o.solve()
            what = f"{desc} call {call} max_iter {st.max_iter} ct {st.check_termination}"
            if rc != orc or w.iter != int(o.get("iter")) or w.status != int(o.get("status")) or sp.contents.solution.contents.solved != int(o.get("sol_solved")):
                return f"{what}: rc/iter/status/solved {rc, w.iter, w.status, sp.contents.solution.contents.solved} vs oracle {orc, int(o.get('iter')), int(o.get('status')), int(o.get('sol_solved'))}"
            for k in FIELDS:
                a, r = pod.to_np(getattr(w, k)), o[k]
                e = float(np.max(np.abs(a - r)) / max(np.max(np.abs(r)), 1e-300)) if a.size else 0.0
                if e > 1e-9:
                    return f"{what}: field {k} off by {e:.2e}"
            sx = pod.to_np(sp.contents.solution.contents.x)
            if sx.size and np.max(np.abs(sx - o["vnew"])) > 1e-9 * max(1.0, np.max(np.abs(o["vnew"]))):
                return f"{what}: solution->x differs"
            for k in ("primal_residual_state", "dual_residual_state", "primal_residual_input", "dual_residual_input"):
                if abs(getattr(w, k) - o.get(k)) > 1e-9 * max(1.0, abs(o.get(k))):
                    return f"{what}: {k} {getattr(w, k)} vs {o.get(k)}"
    finally:
        o.close()
        L.tiny_destroy(sp)
    return None


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    assert build_oracle()
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    bad, msgs = 0, []
    os.dup2(devnull, 1)                       # "Solver converged in N iterations"
    for seed in range(s0, s0 + n):
        try:
            r = trial(seed)
        except Exception as e:                      # noqa: BLE001
            r = f"seed {seed}: {type(e).__name__}: {e}"
        if r:
            bad += 1
            msgs.append(r)
    os.dup2(saved, 1)
    for m in msgs:
        print("MISMATCH", m)
    print(f"{n} trials, {bad} mismatches")
    sys.exit(1 if bad else 0)
