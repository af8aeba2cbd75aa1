"""Differential fuzzing (tools/fuzz_parity.py) as a test: random shapes over all three kernels, random problem data,
settings (check_termination 0..4, max_iter 0..39), enable switches, cones, static / time-varying half-spaces and warm
states against the oracle.  This harness found, in round 1: a 16-entry cone-overlap table indexed with up to 32 rows, a
staging buffer sized for state fields only (nu*(N-1) > nx*N), cone records of a switched-off family being overwritten,
and uninitialised q/r/p/d and x after max_iter = 0."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tools"))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
sys.path.insert(0, HERE)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [1, 2001, 4001])
def test_fuzz_hip_vs_oracle(first):
    import fuzz_parity
    from cpu_solvers import build_oracle
    assert build_oracle()
    bad = [r for r in (fuzz_parity.trial(seed) for seed in range(first, first + 120)) if r]
    assert not bad, bad[:5]


def test_fuzz_adaptive_rho_vs_oracle():
    """tools/fuzz_parity.py adaptive: random shapes of the one-row kernel (compiled-in and run-time instantiated ADAPT variants,
    with and without cones), random sensitivity tables / clip ranges / per-instance cache state."""
    import fuzz_parity
    from cpu_solvers import build_oracle
    assert build_oracle()
    bad = [r for r in (fuzz_parity.adaptive_trial(seed) for seed in range(1, 91)) if r]
    assert not bad, bad[:5]


def test_input_field_larger_than_state_field():
    """(4, 8, 10): nu*(N-1) = 72 > nx*N = 40 -- every host-layout field must fit the staging buffer."""
    import scenarios as sc
    from cpu_solvers import OracleSolver
    from hip_runner import run_cases_hip
    suite = sc.sweep_suite(4, 8, 10, B=5)
    rng = np.random.default_rng(3)
    for k in ("Uref", "znew", "y", "z"):
        suite["cases"][k] = rng.normal(0, 0.3, suite["cases"][k].shape)
    out, ref = run_cases_hip(suite), sc.run_cases(OracleSolver, suite)
    assert np.array_equal(out["iter"].astype(int), ref["iter"].astype(int))
    for k in ("x", "u", "znew", "y", "z"):
        assert np.max(np.abs(out[k] - ref[k])) <= 1e-9 * max(1.0, np.max(np.abs(ref[k]))), k


def test_cones_on_wide_shapes_and_disabled_family_records():
    """A cone on rows >= 16 of a (20, 4, 10) problem (coverage kernel), and yc of a family whose cone switch is off must
    come back untouched from the register kernel."""
    import scenarios as sc
    from cpu_solvers import OracleSolver
    from hip_runner import run_cases_hip
    prob = sc.sweep_suite(20, 4, 10, B=1)["problem"]
    cfg = sc.default_config(prob, max_iter=25, en_state_soc=1, en_input_soc=0, u_min=-0.5, u_max=0.5,
                            state_cone=([17], [3], [0.6]), input_cone=([1], [3], [0.4]))
    cases = sc.zero_cases(prob, 4)
    rng = np.random.default_rng(11)
    for k, v in cases.items():
        cases[k] = rng.normal(0, 0.3, v.shape)
    suite = dict(problem=prob, config=cfg, cases=cases)
    out, ref = run_cases_hip(suite), sc.run_cases(OracleSolver, suite)
    assert np.array_equal(out["iter"].astype(int), ref["iter"].astype(int))
    for k in ("x", "u", "vcnew", "gc"):
        assert np.max(np.abs(out[k] - ref[k])) <= 1e-9 * max(1.0, np.max(np.abs(ref[k]))), k
    suite2, _ = sc.load_suite(os.path.join(HERE, "golden", "rocket_random_isoc.npz"))       # input cone on, state cone off
    suite2["cases"]["gc"] = rng.normal(0, 0.3, suite2["cases"]["gc"].shape)
    out2 = run_cases_hip(suite2)
    assert np.array_equal(out2["gc"], suite2["cases"]["gc"])


@pytest.mark.parametrize("first", [1, 3001])
def test_fuzz_closed_loop_vs_oracle(first):
    """tools/fuzz_closed_loop.py: fused MPC steps, plant step, moving reference window with per-instance offsets,
    reset_duals, heterogeneous families, one-shot launches -- per-step iteration sequences and final states against the
    oracle stepped instance by instance.  Found in round 1: fused steps leaked u_0 into the input lanes' dummy slot 0,
    whose "slack" then entered the dual residual (one extra iteration on some steps)."""
    import fuzz_closed_loop
    from cpu_solvers import build_oracle
    assert build_oracle()
    bad = [r for r in (fuzz_closed_loop.trial(seed) for seed in range(first, first + 100)) if r]
    assert not bad, bad[:5]


def test_fuzz_reference_entry_points_vs_oracle(capfd):
    """tools/fuzz_compat.py: sequences of tiny_solve calls on one plain-data TinySolver with bounds, cones, half-spaces,
    settings and the workspace poked between the calls (the family hash must notice every change)."""
    import fuzz_compat
    from cpu_solvers import build_oracle
    assert build_oracle()
    # 5111, 5126, 5128: every slack family on at once (cones + static + time-varying half-spaces): 24 fields + status come
    # back in one transfer -- found by a round-2 campaign when the single-launch transfer table held 24 entries
    bad = [r for r in (fuzz_compat.trial(seed) for seed in list(range(1, 81)) + [5111, 5126, 5128]) if r]
    capfd.readouterr()                            # drop the "Solver converged in N iterations" lines
    assert not bad, bad[:5]


def test_large_solver_group_through_the_threaded_gather(capfd):
    """tiny_solve_batch on 700 TinySolvers (>= 512: the gather / scatter over the host structs is split over threads):
    every solver gets its own warm state and reference; 25 of them are checked against the oracle, all for bookkeeping."""
    import ctypes as C
    import pod
    import scenarios as sc
    import tinympc_amd as tm
    import fuzz_compat
    from cpu_solvers import OracleSolver
    L = tm.lib()
    fuzz_compat.proto(L)
    L.tiny_solve_batch.argtypes = [C.POINTER(C.POINTER(pod.TinySolver)), C.c_int]
    prob, extra = sc.load_problem("quadrotor_20hz")
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    n = 700
    rng = np.random.default_rng(77)
    keep, arr = [], (C.POINTER(pod.TinySolver) * n)()
    bounds = [np.full((nx, N), -5.0), np.full((nx, N), 5.0), np.full((nu, N - 1), -0.5), np.full((nu, N - 1), 0.5)]
    state = []
    for k in range(n):
        ms = [pod.mat(prob["A"]), pod.mat(prob["B"]), pod.mat(prob["f"]), pod.mat(np.diag(prob["Q"])), pod.mat(np.diag(prob["R"]))]
        pb = [pod.mat(a) for a in bounds]
        keep += [ms, pb]
        sp = C.POINTER(pod.TinySolver)()
        assert L.tiny_setup(C.byref(sp), *[C.byref(m[0]) for m in ms], prob["rho"], nx, nu, N, 0) == 0
        assert L.tiny_set_bound_constraints(sp, *[C.byref(m[0]) for m in pb]) == 0
        sp.contents.settings.contents.max_iter = 40
        w = sp.contents.work.contents
        st = {f: rng.normal(0, 0.2, pod.to_np(getattr(w, f)).shape) for f in ("Xref", "Uref", "vnew", "znew", "g", "y", "v", "z")}
        st["x0"] = rng.uniform(-0.5, 0.5, nx)
        for f, a in st.items():
            if f == "x0":
                pod.to_np(w.x)[:, 0] = a
            else:
                pod.to_np(getattr(w, f))[...] = a
        state.append(st)
        arr[k] = sp
    rc = L.tiny_solve_batch(arr, n)
    capfd.readouterr()
    cfg = sc.default_config(prob, max_iter=40, x_min=bounds[0], x_max=bounds[1], u_min=bounds[2], u_max=bounds[3])
    o = sc.make_solver(OracleSolver, prob, cfg)
    unsolved = 0
    for k in range(n):
        s = arr[k].contents
        unsolved += 1 - s.solution.contents.solved
        if k % 28 == 0:
            for f, a in state[k].items():
                if f == "x0":
                    o["x"][:, 0] = a
                else:
                    o[f] = a
            o.solve()
            w = s.work.contents
            assert w.iter == int(o.get("iter")) and s.solution.contents.solved == int(o.get("sol_solved")), k
            for f in ("x", "u", "vnew", "znew", "g", "y", "v", "z", "q", "r", "p", "d"):
                assert np.max(np.abs(pod.to_np(getattr(w, f)) - o[f])) <= 1e-9 * max(1.0, np.max(np.abs(o[f]))), (k, f)
    assert rc == int(unsolved > 0)
    o.close()
    for k in range(n):
        L.tiny_destroy(arr[k])


def test_fuzz_phase_functions_vs_oracle():
    """tiny_batch_phase (the reference's exported phase functions) on random shapes / families / workspaces."""
    import fuzz_parity
    from cpu_solvers import build_oracle
    assert build_oracle()
    bad = [r for r in (fuzz_parity.phase_trial(seed) for seed in range(1, 101)) if r]
    assert not bad, bad[:5]


def test_fuzz_api_state_machine_vs_oracle():
    """tools/fuzz_api_sequence.py: random operation sequences on one TinyBatch (constraints, settings, field writes,
    reset, solves, kernel-path switches) mirrored on per-instance oracles."""
    import fuzz_api_sequence
    from cpu_solvers import build_oracle
    assert build_oracle()
    bad = [r for r in (fuzz_api_sequence.trial(seed) for seed in range(1, 121)) if r]
    assert not bad, bad[:5]
