#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL TinyMPC reference (oracle/_ref/libtinympc_ref.so).

Run in the build container (needs /root/reference to build the _ref library):
    make -C oracle ref && python oracle/gen_golden.py
The reference ships no tests or golden vectors; these fixtures are
outputs of the reference itself on seeded inputs, and they are what pins both the C
restatement (tests/test_oracle_golden.py) and the HIP path (tests/test_gpu_*.py).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scenarios as sc  # noqa: E402
from cpu_solvers import RefSolver, build_ref  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    """python oracle/gen_golden.py [--only name,name,...]   (--only: just those single-solve suites of section 2)"""
    if build_ref() is None:
        sys.exit("oracle/_ref/libtinympc_ref.so missing and /root/reference absent")
    os.makedirs(OUT, exist_ok=True)
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    if only is None:
        cache_kats()
    single_solve_suites(only)
    if only is None:
        episode_and_phase_kats()


def cache_kats():
    # 1. cache known-answer values (tiny_api.cpp:307-381) for the four problem families
    kat = {}
    for name in ("codegen_random", "cartpole", "quadrotor_20hz", "rocket_landing_20hz"):
        prob, _ = sc.load_problem(name)
        s = sc.make_solver(RefSolver, prob, sc.default_config(prob))
        for k in ("Kinf", "Pinf", "Quu_inv", "AmBKt", "APf", "BPf", "Q", "R"):
            kat[f"{name}.{k}"] = s[k].copy()
        s.close()
    np.savez_compressed(os.path.join(OUT, "cache_kat.npz"), **kat)


def single_solve_suites(only=None):
    # 2. single-solve suites
    suites = {
        "hover_warm": sc.hover_suite(RefSolver),
        "tracking_random": sc.tracking_random_suite(),
        "rocket_random_isoc": sc.rocket_random_suite(en_state_soc=0, en_input_soc=1),
        "rocket_random_bothsoc": sc.rocket_random_suite(B=6, en_state_soc=1, en_input_soc=1),
        "rocket_episode_isoc": sc.rocket_episode_suite(RefSolver, 0, 1),
        "rocket_episode_nosoc": sc.rocket_episode_suite(RefSolver, 0, 0, steps=(0, 3, 40)),
        "cartpole_episode": sc.cartpole_suite(RefSolver),
        "random_state_quad": sc.random_state_suite("quadrotor_20hz", B=8, seed=7),
        "random_state_rocket_soc": sc.random_state_suite("rocket_landing_20hz", B=8, seed=11, soc=True),
        # overlapping cones (admm.cpp:111-135 applies them sequentially per column): quadrotor 3 state + 2 input cones sharing rows,
        # rocket two input cones on the same three rows
        "random_state_quad_overlap_soc": sc.random_state_suite("quadrotor_20hz", B=6, seed=13, soc="overlap"),
        "random_state_rocket_overlap_soc": sc.random_state_suite("rocket_landing_20hz", B=5, seed=14, soc="overlap"),
        "sweep_4_2_10": sc.sweep_suite(4, 2, 10),
        "sweep_8_4_30": sc.sweep_suite(8, 4, 30, B=3),
        "sweep_12_2_10": sc.sweep_suite(12, 2, 10, B=3),
        "sweep_20_8_10": sc.sweep_suite(20, 8, 10, B=2),
        # one golden per tile-kernel shape class (tile_dims.txt): long one-row-wide (W=1,R=2), wide N=30 and wide N=50 (W=2,R=2)
        "sweep_4_2_50": sc.sweep_suite(4, 2, 50, B=3),
        "sweep_12_4_50": sc.sweep_suite(12, 4, 50, B=3),
        "sweep_12_8_30": sc.sweep_suite(12, 8, 30, B=2),
        "sweep_20_8_50": sc.sweep_suite(20, 8, 50, B=2),
        # the tile kernel's cone variant (VERDICT r04 nit): a wide shape with an input cone, a long one with both families on
        "sweep_20_4_10_isoc": sc.sweep_cone_suite(20, 4, 10, B=3, max_iter=250),
        "sweep_12_8_30_isoc": sc.sweep_cone_suite(12, 8, 30, B=3, max_iter=150),
        "sweep_8_4_50_bothsoc": sc.sweep_cone_suite(8, 4, 50, B=2, max_iter=60, state_cone=True),
        # adaptive rho (admm.cpp:397-423, rho_benchmark.cpp) with the reference's own sensitivity tables; the real reference
        # runs with the stack under solve() scrubbed (oracle/ref_shim.cpp: its RhoAdapter flag is uninitialised)
        "adaptive_hover": sc.hover_adaptive_suite(RefSolver),
        "adaptive_tracking": sc.tracking_adaptive_suite(),
        "adaptive_tracking_noclip": sc.tracking_adaptive_suite(B=8, seed=5, clip=0, max_iter=40),
        "linear_random_all": sc.random_linear_suite("quadrotor_20hz", B=4, seed=21),
        "linear_random_rocket_soc": sc.random_linear_suite("rocket_landing_20hz", B=4, seed=22, soc=True),
        "linear_random_tv_only": sc.random_linear_suite("cartpole", B=4, seed=23, static=False, box=False),
    }
    for tv in ((False, True) if only is None else ()):
        subs, its, solved = sc.linear_example_suite(RefSolver, tv)
        for n, sub in enumerate(subs):
            suites[f"linear_example_{'tv' if tv else 'static'}_{n}"] = sub
        print("linear example", "tv" if tv else "static", "solved", int(solved.sum()), "of", len(solved),
              "converged-iteration sum", int(its[solved == 1].sum()))
    slim = ("x", "u", "vnew", "znew", "g", "y", "v", "z", "vcnew", "zcnew", "gc", "yc")
    for name, suite in suites.items():
        if only is not None and name not in only:
            continue
        soc = suite["config"]["en_state_soc"] or suite["config"]["en_input_soc"]
        fields = [f for f in slim if soc or f not in ("vcnew", "zcnew", "gc", "yc")]
        if suite["config"].get("en_state_linear") or suite["config"].get("en_input_linear"):
            fields += ["vlnew", "zlnew", "gl", "yl"]
        if suite["config"].get("en_tv_state_linear") or suite["config"].get("en_tv_input_linear"):
            fields += ["vlnew_tv", "zlnew_tv", "gl_tv", "yl_tv"]
        if name.startswith("random_state"):
            fields += ["q", "r", "p", "d", "sol_x", "sol_u"]
        out = sc.run_cases(RefSolver, suite, fields=fields)
        if suite["config"].get("adaptive_rho"):
            print(f"{name:26s} rho after the solve: {np.round(out['rho'], 4).tolist()}")
        sc.save_suite(os.path.join(OUT, name + ".npz"), suite, out)
        ep = suite.get("episode")
        print(f"{name:26s} B={len(out['iter'])} iters={out['iter'].astype(int).tolist()}"
              + (f" episode_total={int(ep['iters'].sum())}" if ep else ""))


def episode_and_phase_kats():
    # 3. tracking episode summary (examples/quadrotor_tracking.cpp): per-step iterations
    prob, extra = sc.load_problem("quadrotor_20hz")
    cfg = sc._hover_cfg(prob, extra)
    s = sc.make_solver(RefSolver, prob, cfg)
    traj = np.array(extra["y_axis_line"])
    N, nx, nu = prob["N"], prob["nx"], prob["nu"]
    s["Xref"] = traj[0:N].T
    x0 = s["Xref"][:, 0].copy()
    its, u0s = [], []
    for k in range(301 - N):
        s["x"][:, 0] = x0
        s["Xref"] = traj[k:k + N].T
        s["y"] = np.zeros((nu, N - 1))
        s["g"] = np.zeros((nx, N))
        s.solve()
        its.append(int(s.get("sol_iter")))
        u0s.append(s["u"][:, 0].copy())
        x0 = prob["A"] @ x0 + prob["B"] @ s["u"][:, 0]
    s.close()
    np.savez_compressed(os.path.join(OUT, "tracking_episode.npz"), iters=np.array(its, dtype=np.int32),
                        u0=np.array(u0s), x_final=x0)
    print("tracking episode total", sum(its))

    # 4. project_soc known answers (admm.cpp:39-60) incl. the three branches and float truncation
    rng = np.random.default_rng(5)
    S = rng.normal(0, 1, (64, 3))
    S[0] = [0.3, 0.4, 10.0]      # inside cone
    S[1] = [0.3, 0.4, -10.0]     # below cone -> origin
    S[2] = [3.0, 4.0, 1.0]       # outside
    S[3] = [0.0, 0.0, 0.0]
    mus = rng.uniform(0.1, 1.5, 64)
    mus[:4] = [0.5, 0.5, 0.25, 0.3]
    prob, _ = sc.load_problem("codegen_random")
    s = sc.make_solver(RefSolver, prob, sc.default_config(prob))
    P = np.stack([s.project_soc(S[i], mus[i]) for i in range(64)])
    s.close()
    np.savez_compressed(os.path.join(OUT, "project_soc_kat.npz"), s=S, mu=mus, out=P)

    # 5. per-phase known answers on a random workspace (every phase symbol is exported, admm.hpp:12-17)
    suite = sc.random_state_suite("rocket_landing_20hz", B=1, seed=3, soc=True)
    s = sc.make_solver(RefSolver, suite["problem"], suite["config"])
    rng = np.random.default_rng(3)
    for k in s.STATE_FIELDS + ("Xref", "Uref"):
        s[k] = rng.normal(0, 0.5, s[k].shape)
    ph = {"in." + k: s[k].copy() for k in s.STATE_FIELDS + ("Xref", "Uref")}
    for name in ("update_linear_cost", "backward_pass_grad", "forward_pass", "update_slack", "update_dual"):
        s.phase(name)
        for k in s.STATE_FIELDS:
            ph[f"{name}.{k}"] = s[k].copy()
    s.set("check_termination", 1)
    ph["termination"] = np.array([s.phase("termination_condition")] + [s.get(k) for k in (
        "primal_residual_state", "dual_residual_state", "primal_residual_input", "dual_residual_input")])
    s.close()
    np.savez_compressed(os.path.join(OUT, "phase_kat.npz"), **ph)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes:", tot)


if __name__ == "__main__":
    main()
