#!/usr/bin/env python3
"""Differential fuzzing of the batched C ABI as a STATE MACHINE: a random sequence of operations on one TinyBatch
(bounds, cones, half-spaces, settings, field writes, reset, solves, kernel-path switches via force_general / no_tile /
debug) mirrored on one oracle per instance; after every solve every record must agree.  Stresses the lazy table
rebuilds and the path-switch logic.   python tools/fuzz_api_sequence.py [n_trials] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios as sc
import tinympc_amd as tm
from cpu_solvers import OracleSolver, build_oracle

SHAPES = [(4, 1, 10), (12, 4, 10), (6, 3, 10), (2, 2, 3), (4, 8, 10), (8, 4, 30), (12, 8, 10), (5, 3, 7), (20, 4, 10), (8, 2, 50)]
STATE = ("vnew", "znew", "g", "y", "v", "z", "x", "u")


def trial(seed):
    rng = np.random.default_rng(seed)
    nx, nu, N = SHAPES[rng.integers(len(SHAPES))]
    M = rng.standard_normal((nx, nx))
    prob = dict(nx=nx, nu=nu, N=N, rho=float(rng.choice([0.5, 1.0, 5.0])), A=M * rng.uniform(0.5, 1.0) / np.max(np.abs(np.linalg.eigvals(M))),
                B=rng.standard_normal((nx, nu)) / np.sqrt(nx), f=rng.normal(0, 0.02, nx) * rng.integers(0, 2),
                Q=rng.uniform(0.5, 10, nx), R=rng.uniform(0.1, 2, nu))
    B = int(rng.integers(1, 7))
    s = tm.TinyBatchSolver.from_problem(prob, B)
    cfg = sc.default_config(prob)
    os_ = [sc.make_solver(OracleSolver, prob, cfg) for _ in range(B)]
    flags = dict(max_iter=20, check_termination=1, abs_pri_tol=1e-3, abs_dua_tol=1e-3, en_state_bound=1, en_input_bound=1,
                 en_state_soc=0, en_input_soc=0, en_state_linear=0, en_input_linear=0, en_tv_state_linear=0, en_tv_input_linear=0)
    have = dict(cones=False, lin=False, tv=False)
    log = []
    desc = f"seed {seed} shape {(nx, nu, N)} B {B}"

    def push_settings():
        s.update_settings(flags["abs_pri_tol"], flags["abs_dua_tol"], flags["max_iter"], flags["check_termination"],
                          flags["en_state_bound"], flags["en_input_bound"], flags["en_state_soc"], flags["en_input_soc"],
                          flags["en_state_linear"], flags["en_input_linear"], flags["en_tv_state_linear"], flags["en_tv_input_linear"])
        for o in os_:
            for k, v in flags.items():
                o.set(k, v)
    try:
        push_settings()
        x0 = rng.uniform(-0.5, 0.5, (B, nx))
        s.set_x0(x0)
        for b, o in enumerate(os_):
            o["x"][:, 0] = x0[b]
        for step in range(int(rng.integers(6, 16))):
            op = rng.integers(0, 11)
            log.append(int(op))
            if op == 0:
                bd = [rng.uniform(-2.0, -0.2, (nx, N)), rng.uniform(0.2, 2.0, (nx, N)), rng.uniform(-1.0, -0.1, (nu, N - 1)), rng.uniform(0.1, 1.0, (nu, N - 1))]
                s.set_bound_constraints(*bd)
                for o in os_:
                    o.set_bounds(*bd)
            elif op == 1 and nx >= 3 and nu >= 3:
                cs = ([int(rng.integers(0, nx - 2))], [3], [float(rng.uniform(0.3, 1.2))], [int(rng.integers(0, nu - 2))], [3], [float(rng.uniform(0.3, 1.2))])
                s.set_cone_constraints(*cs)
                for o in os_:
                    o.set_cones(*cs)
                have["cones"] = True
            elif op == 2:
                ns, ni = int(rng.integers(1, 4)), int(rng.integers(1, 3))
                ln = (rng.standard_normal((ns, nx)), rng.uniform(0.1, 1.0, ns), rng.standard_normal((ni, nu)), rng.uniform(0.05, 0.5, ni))
                s.set_linear_constraints(*ln)
                for o in os_:
                    o.set_linear(*ln)
                have["lin"] = True
            elif op == 3:
                ns, ni = int(rng.integers(1, 3)), int(rng.integers(1, 3))
                tv = (rng.standard_normal((ns * N, nx)), rng.uniform(0.1, 1.0, (ns, N)), rng.standard_normal((ni * (N - 1), nu)), rng.uniform(0.05, 0.5, (ni, N - 1)))
                s.set_tv_linear_constraints(*tv)
                for o in os_:
                    o.set_tv_linear(*tv)
                have["tv"] = True
            elif op == 4:
                flags.update(max_iter=int(rng.integers(0, 30)), check_termination=int(rng.integers(1, 4)),
                             abs_pri_tol=float(10 ** rng.uniform(-4, -1)), abs_dua_tol=float(10 ** rng.uniform(-4, -1)),
                             en_state_bound=int(rng.integers(0, 2)), en_input_bound=int(rng.integers(0, 2)))
                if have["cones"]:
                    flags.update(en_state_soc=int(rng.integers(0, 2)), en_input_soc=int(rng.integers(0, 2)))
                if have["lin"]:
                    flags.update(en_state_linear=int(rng.integers(0, 2)), en_input_linear=int(rng.integers(0, 2)))
                if have["tv"]:
                    flags.update(en_tv_state_linear=int(rng.integers(0, 2)), en_tv_input_linear=int(rng.integers(0, 2)))
                push_settings()
            elif op == 5:
                k = STATE[rng.integers(len(STATE))]
                a = rng.normal(0, 0.3, (B,) + os_[0][k].shape)
                s.set(k, a)
                for b, o in enumerate(os_):
                    o[k] = a[b]
                    if k == "x":
                        o["x"][:, 0] = x0[b]               # x0 is a separate record in the batched ABI (tiny_set_x0)
            elif op == 6:
                xr, ur = rng.normal(0, 0.3, (B, nx, N)), rng.normal(0, 0.05, (B, nu, N - 1))
                x0 = rng.uniform(-0.5, 0.5, (B, nx))
                s.set_x_ref(xr); s.set_u_ref(ur); s.set_x0(x0)
                for b, o in enumerate(os_):
                    o["Xref"] = xr[b]; o["Uref"] = ur[b]; o["x"][:, 0] = x0[b]
            elif op == 7:
                s.reset()
                for o in os_:
                    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z", "vcnew", "zcnew", "gc", "yc", "vlnew", "zlnew", "gl", "yl",
                              "vlnew_tv", "zlnew_tv", "gl_tv", "yl_tv"):
                        o[k] = np.zeros_like(o[k])
                x0 = s.get("x0")
                for b, o in enumerate(os_):
                    o["x"][:, 0] = x0[b]
            elif op == 8:
                s.set_option(["force_general", "no_tile", "debug"][rng.integers(0, 3)], int(rng.integers(0, 2)))
            else:
                rc = s.solve()
                st = s.status()
                orc = 0
                for b, o in enumerate(os_):
                    orc |= o.solve()
                    if int(o.get("iter")) != int(st["iter"][b]) or int(o.get("sol_solved")) != int(st["solved"][b]):
                        return f"{desc} ops {log} [{s.kernel_path()}]: instance {b} iter/solved {int(st['iter'][b])}/{int(st['solved'][b])} vs oracle {int(o.get('iter'))}/{int(o.get('sol_solved'))}"
                if rc != orc:
                    return f"{desc} ops {log}: return code {rc} vs {orc}"
                fields = ["vnew", "znew", "g", "y", "v", "z"] + (["x", "u"] if flags["max_iter"] > 0 else [])
                if flags["en_state_soc"] and have["cones"]:
                    fields += ["vcnew", "gc"]
                if flags["en_input_soc"] and have["cones"]:
                    fields += ["zcnew", "yc"]
                if flags["en_state_linear"]:
                    fields += ["vlnew", "gl"]
                if flags["en_input_linear"]:
                    fields += ["zlnew", "yl"]
                if flags["en_tv_state_linear"]:
                    fields += ["vlnew_tv", "gl_tv"]
                if flags["en_tv_input_linear"]:
                    fields += ["zlnew_tv", "yl_tv"]
                for k in fields:
                    got = s.get(k)
                    for b, o in enumerate(os_):
                        e = float(np.max(np.abs(got[b] - o[k])) / max(np.max(np.abs(o[k])), 1e-300))
                        if e > 1e-9:
                            return f"{desc} ops {log} [{s.kernel_path()}]: {k}[{b}] off by {e:.2e}"
    finally:
        s.close()
        for o in os_:
            o.close()
    return None


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
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
