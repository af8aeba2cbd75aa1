#!/usr/bin/env python3
"""Differential fuzzing of the HIP path against the oracle: random shapes (register, tile and coverage kernels), random
problem data, random settings (check_termination 0..4, odd max_iter, tolerances, every enable switch, cones, static and
time-varying half-spaces), random warm states.  Bar: identical iteration counts / solved flags, fields within 1e-9.
    python tools/fuzz_parity.py [n_trials] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios as sc
from cpu_solvers import OracleSolver, build_oracle
from hip_runner import run_cases_hip

SHAPES = [(4, 1, 10), (12, 4, 10), (6, 3, 10), (2, 2, 3), (8, 8, 10), (4, 8, 10), (12, 2, 10), (8, 4, 30), (4, 2, 30),
          (12, 8, 10), (20, 4, 10), (8, 2, 50), (20, 2, 30), (5, 3, 7), (9, 2, 12), (3, 1, 4), (16, 8, 6), (7, 7, 5)]


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def random_suite(seed):
    rng = np.random.default_rng(seed)
    nx, nu, N = SHAPES[rng.integers(len(SHAPES))]
    M = rng.standard_normal((nx, nx))
    A = M * rng.uniform(0.5, 1.02) / np.max(np.abs(np.linalg.eigvals(M)))
    prob = dict(nx=nx, nu=nu, N=N, rho=float(rng.choice([0.1, 1.0, 5.0, 17.3])), A=A, B=rng.standard_normal((nx, nu)) / np.sqrt(nx),
                f=rng.normal(0, 0.05, nx) * rng.integers(0, 2), Q=rng.uniform(0.5, 10, nx), R=rng.uniform(0.1, 2, nu))
    kw = dict(max_iter=int(rng.integers(0, 40)) * int(rng.random() > 0.08), check_termination=int(rng.integers(0, 5)),
              abs_pri_tol=float(10 ** rng.uniform(-5, -1)), abs_dua_tol=float(10 ** rng.uniform(-5, -1)),
              en_state_bound=int(rng.integers(0, 2)), en_input_bound=int(rng.integers(0, 2)),
              x_min=rng.uniform(-2.0, -0.1, (nx, N)), x_max=rng.uniform(0.1, 2.0, (nx, N)),
              u_min=rng.uniform(-1.0, -0.05, (nu, N - 1)), u_max=rng.uniform(0.05, 1.0, (nu, N - 1)))
    if nx >= 3 and rng.random() < 0.4:
        kw.update(en_state_soc=int(rng.integers(0, 2)), en_input_soc=int(rng.integers(0, 2)) if nu >= 3 else 0,
                  state_cone=([int(rng.integers(0, nx - 2))], [3], [float(rng.uniform(0.2, 1.5))]),
                  input_cone=([int(rng.integers(0, nu - 2))], [3], [float(rng.uniform(0.2, 1.5))]) if nu >= 3 else ([], [], []))
    if rng.random() < 0.35:
        ns, ni = int(rng.integers(0, 10)), int(rng.integers(0, 7))      # <= 4: compiled-in LIN variants, <= 8: run-time ones, more: coverage
        kw.update(en_state_linear=int(rng.integers(0, 2)), en_input_linear=int(rng.integers(0, 2)),
                  linear=(rng.standard_normal((ns, nx)), rng.uniform(0.1, 1.0, ns), rng.standard_normal((ni, nu)), rng.uniform(0.05, 0.5, ni)))
    if rng.random() < 0.3:
        ns, ni = int(rng.integers(0, 7)), int(rng.integers(0, 4))
        kw.update(en_tv_state_linear=int(rng.integers(0, 2)), en_tv_input_linear=int(rng.integers(0, 2)),
                  tv_linear=(rng.standard_normal((ns * N, nx)), rng.uniform(0.1, 1.0, (ns, N)),
                             rng.standard_normal((ni * (N - 1), nu)), rng.uniform(0.05, 0.5, (ni, N - 1))))
    cfg = sc.default_config(prob, **kw)
    B = int(rng.integers(1, 9))
    cases = sc.zero_cases(prob, B)
    warm = rng.random() < 0.6
    for k, v in cases.items():
        if k in ("x0", "Xref", "Uref") or warm:
            cases[k] = rng.normal(0.0, 0.4, v.shape)
    return dict(problem=prob, config=cfg, cases=cases), kw


def trial(seed):
    suite, kw = random_suite(seed)
    prob, cfg, cases = suite["problem"], suite["config"], suite["cases"]
    nx, nu, N, B = prob["nx"], prob["nu"], prob["N"], cases["x0"].shape[0]
    ref = sc.run_cases(OracleSolver, suite)
    orng = np.random.default_rng(seed + 77_000_000)     # launch options: their own stream, so that the problems of a seed stay what they were
    opts = {}
    if orng.random() < 0.35:                             # split solve: stop at K, carry the open instances on (must change nothing)
        opts = {"repack_after": int(orng.integers(1, max(2, kw["max_iter"]))), "repack_growth": int(orng.integers(2, 4))}
    if orng.random() < 0.4:                              # round 3: the tile kernel's launch forms (must change nothing either): dynamic slots on
        opts.update({"tile_dyn": int(orng.integers(0, 2)), "prefer_tile": int(orng.integers(0, 2)), "tile_r": int(orng.integers(0, 3))})   # a persistent grid, the one-row-layout / half-row forms of one-row shapes, the other R of a shape
    out = run_cases_hip(suite, options=opts)
    desc = f"seed {seed} shape {(nx, nu, N)} B {B} max_iter {kw['max_iter']} ct {kw['check_termination']} opts {opts} flags " + \
           "".join(str(cfg[k]) for k in ("en_state_bound", "en_input_bound", "en_state_soc", "en_input_soc", "en_state_linear",
                                          "en_input_linear", "en_tv_state_linear", "en_tv_input_linear"))
    for k in ("iter", "sol_solved", "status"):
        if not np.array_equal(out[k].astype(int), ref[k].astype(int)):
            return f"{desc}: {k} {out[k].astype(int).tolist()} vs {ref[k].astype(int).tolist()}"
    worst, wk = 0.0, ""
    for k, v in ref.items():
        if v.ndim >= 2 and k in out:
            e = rel_err(out[k], v)
            if e > worst:
                worst, wk = e, k
    if worst > 1e-9:
        return f"{desc}: worst field error {worst:.3e} in {wk}"
    return None


ADAPT_SHAPES = [(4, 1, 10), (12, 4, 10), (6, 3, 10), (2, 2, 3), (8, 8, 10), (4, 8, 10), (12, 2, 10), (5, 3, 7), (9, 2, 12), (3, 1, 4), (7, 7, 5)]


def adaptive_trial(seed):
    """Adaptive rho (admm.cpp:397-423, rho_benchmark.cpp) on random shapes of the one-row kernel: random sensitivity tables,
    clip range and per-instance cache state, optional cones; identical iteration counts, fields and the moved cache to 1e-7
    (the re-estimated rho feeds back through a square root of a ratio of maxima: last-bit differences of the FMA-contracted
    sweeps are amplified over the adaptations of a solve)."""
    rng = np.random.default_rng(seed + 555_000)
    nx, nu, N = ADAPT_SHAPES[rng.integers(len(ADAPT_SHAPES))]
    M = rng.standard_normal((nx, nx))
    A = M * rng.uniform(0.5, 1.0) / np.max(np.abs(np.linalg.eigvals(M)))
    prob = dict(nx=nx, nu=nu, N=N, rho=float(rng.choice([1.0, 5.0, 17.3])), A=A, B=rng.standard_normal((nx, nu)) / np.sqrt(nx),
                f=rng.normal(0, 0.05, nx) * rng.integers(0, 2), Q=rng.uniform(0.5, 10, nx), R=rng.uniform(0.1, 2, nu))
    kw = dict(max_iter=int(rng.integers(0, 60)), check_termination=int(rng.integers(0, 4)), abs_pri_tol=float(10 ** rng.uniform(-4, -1)),
              abs_dua_tol=float(10 ** rng.uniform(-4, -1)), en_state_bound=int(rng.integers(0, 2)), en_input_bound=int(rng.integers(0, 2)),
              x_min=rng.uniform(-2.0, -0.1, (nx, N)), x_max=rng.uniform(0.1, 2.0, (nx, N)),
              u_min=rng.uniform(-1.0, -0.05, (nu, N - 1)), u_max=rng.uniform(0.05, 1.0, (nu, N - 1)))
    if nx >= 3 and nu >= 3 and rng.random() < 0.3:
        kw.update(en_state_soc=int(rng.integers(0, 2)), en_input_soc=int(rng.integers(0, 2)),
                  state_cone=([int(rng.integers(0, nx - 2))], [3], [float(rng.uniform(0.2, 1.5))]),
                  input_cone=([int(rng.integers(0, nu - 2))], [3], [float(rng.uniform(0.2, 1.5))]))
    sens = {"dKinf_drho": rng.normal(0, 2e-3, (nu, nx)), "dPinf_drho": rng.normal(0, 5e-2, (nx, nx)),
            "dC1_drho": rng.normal(0, 1e-3, (nu, nu)), "dC2_drho": rng.normal(0, 1e-3, (nx, nx))}
    lo = float(rng.uniform(0.3, 2.0))
    cfg = sc.adaptive_cfg(sc.default_config(prob, **kw), rho_min=lo, rho_max=lo * float(rng.uniform(2, 50)), clip=int(rng.random() < 0.8), sensitivity=sens)
    B = int(rng.integers(1, 9))
    cases = sc.zero_cases(prob, B)
    warm = rng.random() < 0.5
    for k, v in cases.items():
        if k in ("x0", "Xref", "Uref") or warm:
            cases[k] = rng.normal(0.0, 0.4, v.shape)
    suite = dict(problem=prob, config=cfg, cases=cases)
    if rng.random() < 0.5:                               # every instance starts from its own rho (and slightly moved Kinf / Pinf)
        o = sc.make_solver(OracleSolver, prob, cfg)
        cases["cache_rho"] = prob["rho"] * rng.uniform(0.5, 2.0, B)
        for k in sc.CACHE_STATE:
            cases["cache_" + k] = o[k][None] * (1.0 + rng.normal(0, 1e-3, (B,) + o[k].shape))
        o.close()
    ref = sc.run_cases(OracleSolver, suite)
    out = run_cases_hip(suite)
    desc = f"seed {seed} adaptive shape {(nx, nu, N)} B {B} max_iter {kw['max_iter']} ct {kw['check_termination']} soc {cfg['en_state_soc']}{cfg['en_input_soc']}"
    for k in ("iter", "sol_solved", "status"):
        if not np.array_equal(out[k].astype(int), ref[k].astype(int)):
            return f"{desc}: {k} {out[k].astype(int).tolist()} vs {ref[k].astype(int).tolist()}"
    if not np.allclose(out["rho"], ref["rho"], rtol=1e-7, atol=0):
        return f"{desc}: rho {out['rho']} vs {ref['rho']}"
    worst, wk = 0.0, ""
    for k, v in ref.items():
        if v.ndim >= 2 and k in out:
            e = rel_err(out[k], v)
            if e > worst:
                worst, wk = e, k
    if worst > 1e-7:
        return f"{desc}: worst field error {worst:.3e} in {wk}"
    return None


PHASES = ("update_linear_cost", "backward_pass_grad", "forward_pass", "update_slack", "update_dual")


def phase_trial(seed):
    """The exported phase functions (tiny_batch_phase) one after the other on a fully random workspace."""
    from hip_runner import make_batch
    suite, kw = random_suite(seed)
    prob, cfg, cases = suite["problem"], suite["config"], suite["cases"]
    B = cases["x0"].shape[0]
    rng = np.random.default_rng(seed + 77777)
    fields = ["x", "u", "q", "r", "p", "d", "v", "vnew", "z", "znew", "g", "y"]
    if (cfg["en_state_soc"] and cfg["state_cone"] is not None and len(cfg["state_cone"][0])) or \
       (cfg["en_input_soc"] and cfg["input_cone"] is not None and len(cfg["input_cone"][0])):
        fields += ["vcnew", "zcnew", "gc", "yc"]
    if cfg["en_state_linear"] or cfg["en_input_linear"]:
        fields += ["vlnew", "zlnew", "gl", "yl"]
    if cfg["en_tv_state_linear"] or cfg["en_tv_input_linear"]:
        fields += ["vlnew_tv", "zlnew_tv", "gl_tv", "yl_tv"]
    s = make_batch(suite)
    oracles = [sc.make_solver(OracleSolver, prob, cfg) for _ in range(B)]
    desc = f"seed {seed} shape {(prob['nx'], prob['nu'], prob['N'])} B {B}"
    try:
        for k in fields + ["Xref", "Uref"]:
            a = rng.normal(0, 0.5, (B,) + oracles[0][k].shape)
            s.set(k, a)
            for b, o in enumerate(oracles):
                o[k] = a[b]
        for name in PHASES:
            s.phase(name)
            for o in oracles:
                o.phase(name)
            for k in fields:
                got = s.get(k)
                for b, o in enumerate(oracles):
                    e = rel_err(got[b], o[k])
                    if e > 1e-9:
                        return f"{desc}: after {name}: {k}[{b}] off by {e:.2e}"
        conv = s.phase("termination_condition")
        st = s.status()
        for b, o in enumerate(oracles):
            o.set("check_termination", 1)
            if bool(o.phase("termination_condition")) != bool(conv[b]):
                return f"{desc}: termination_condition[{b}] differs"
            for k in ("primal_residual_state", "dual_residual_state", "primal_residual_input", "dual_residual_input"):
                if abs(st[k][b] - o.get(k)) > 1e-9 * max(1.0, abs(o.get(k))):
                    return f"{desc}: {k}[{b}] {st[k][b]} vs {o.get(k)}"
    finally:
        s.close()
        for o in oracles:
            o.close()
    return None


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "phases":
        trial = phase_trial
    if len(sys.argv) > 3 and sys.argv[3] == "adaptive":
        trial = adaptive_trial
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    assert build_oracle()
    bad = 0
    for seed in range(s0, s0 + n):
        if os.environ.get("FUZZ_VERBOSE"):
            print("seed", seed, flush=True)
        try:
            r = trial(seed)
        except Exception as e:                      # noqa: BLE001
            r = f"seed {seed}: {type(e).__name__}: {e}"
        if r:
            bad += 1
            print("MISMATCH", r, flush=True)
    print(f"{n} trials, {bad} mismatches")
    sys.exit(1 if bad else 0)
