"""CPU tests: the plain-C restatement (oracle/tinympc_oracle.c) against the golden fixtures
generated from the REAL reference (oracle/gen_golden.py), and against the live compiled
reference when oracle/_ref/libtinympc_ref.so is present.

Tolerances: the oracle differs from the Eigen build only in summation order, so results agree
to ~1e-12; the bar used here is 1e-9 relative (four orders tighter than the 1e-5 contract) and
bit-identical iteration counts / status codes.
"""
import glob
import os

import numpy as np
import pytest

import scenarios as sc
from cpu_solvers import OracleSolver, RefSolver, have_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUITES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                if os.path.basename(p)[:-4] not in ("cache_kat", "project_soc_kat", "phase_kat", "tracking_episode"))
RTOL = 1e-9


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def assert_outputs_match(out, ref, rtol, what=""):
    for k in ("iter", "status", "sol_iter", "sol_solved", "ret"):
        assert np.array_equal(out[k].astype(int), ref[k].astype(int)), f"{what}: {k} differs {out[k]} vs {ref[k]}"
    for k, v in ref.items():
        if v.ndim >= 2:
            for b in range(v.shape[0]):
                e = rel_err(out[k][b], v[b])
                assert e <= rtol, f"{what}: field {k} case {b} rel err {e:.3e} > {rtol}"
        elif k.startswith(("primal_", "dual_")):
            assert np.allclose(out[k], v, rtol=1e-7, atol=1e-12), f"{what}: {k}"
        elif k == "rho":                                        # adaptive rho: cache->rho after the solve
            assert np.allclose(out[k], v, rtol=rtol, atol=0.0), f"{what}: rho {out[k]} vs {v}"


def test_suites_present():
    assert len(SUITES) >= 10


@pytest.mark.parametrize("name", SUITES)
def test_oracle_matches_reference_golden(name):
    suite, ref = sc.load_suite(os.path.join(GOLDEN, name + ".npz"))
    out = sc.run_cases(OracleSolver, suite)
    assert_outputs_match(out, ref, RTOL, name)


def test_cache_known_answers():
    """tiny_precompute_and_set_cache (tiny_api.cpp:307-381) incl. the double rho."""
    kat = np.load(os.path.join(GOLDEN, "cache_kat.npz"))
    for name, riccati in (("codegen_random", None), ("cartpole", 454), ("quadrotor_20hz", 55),
                          ("rocket_landing_20hz", 218)):
        prob, _ = sc.load_problem(name)
        s = sc.make_solver(OracleSolver, prob, sc.default_config(prob))
        for k in ("Kinf", "Pinf", "Quu_inv", "AmBKt", "APf", "BPf", "Q", "R"):
            assert rel_err(s[k], kat[f"{name}.{k}"]) < 1e-12, (name, k)
        if riccati:
            assert int(s.get("riccati_iters")) == riccati
        s.close()
    # values printed by the reference's own codegen_random example
    prob, _ = sc.load_problem("codegen_random")
    s = sc.make_solver(OracleSolver, prob, sc.default_config(prob))
    np.testing.assert_allclose(s["Kinf"], [[0.2207297053937700, 0.2665243149225075],
                                           [0.1025073019418153, 1.1227833027341685]], rtol=1e-13)
    np.testing.assert_allclose(s["Quu_inv"], [[0.0267133031136454, -0.0473209370478584],
                                              [-0.0473209370478584, 0.1298393696141099]], rtol=1e-12)
    np.testing.assert_allclose(s["AmBKt"], [[0.0302889779932443, 0.0145738764831047],
                                            [0.8320771470299722, -0.1888805624241985]], rtol=1e-12)
    s.close()
    # cartpole: printed Q = diag(12,3,12,3) and R = 3 -> user value + 2*rho on the cache path
    prob, _ = sc.load_problem("cartpole")
    s = sc.make_solver(OracleSolver, prob, sc.default_config(prob))
    np.testing.assert_allclose(s["Q"].ravel(), [11, 2, 11, 2])     # work->Q carries ONE rho
    np.testing.assert_allclose(s["Kinf"].ravel(), [-1.8281816030708571, -2.4111848780256802,
                                                   20.6738188202801112, 3.3664150316402646], rtol=1e-12)
    s.close()


def test_project_soc_known_answers():
    kat = np.load(os.path.join(GOLDEN, "project_soc_kat.npz"))
    prob, _ = sc.load_problem("codegen_random")
    s = sc.make_solver(OracleSolver, prob, sc.default_config(prob))
    for i in range(len(kat["mu"])):
        got = s.project_soc(kat["s"][i], kat["mu"][i])
        assert np.array_equal(got, kat["out"][i]), (i, got, kat["out"][i])   # bit-exact incl. float truncation
    s.close()


def test_phase_known_answers():
    """Each exported phase (admm.hpp:12-17) on a random workspace, SOC enabled."""
    ph = np.load(os.path.join(GOLDEN, "phase_kat.npz"))
    suite = sc.random_state_suite("rocket_landing_20hz", B=1, seed=3, soc=True)
    s = sc.make_solver(OracleSolver, suite["problem"], suite["config"])
    for k in s.STATE_FIELDS + ("Xref", "Uref"):
        s[k] = ph["in." + k]
    for name in ("update_linear_cost", "backward_pass_grad", "forward_pass", "update_slack", "update_dual"):
        s.phase(name)
        for k in s.STATE_FIELDS:
            assert rel_err(s[k], ph[f"{name}.{k}"]) < 1e-12, (name, k)
    s.set("check_termination", 1)
    t = [s.phase("termination_condition")] + [s.get(k) for k in (
        "primal_residual_state", "dual_residual_state", "primal_residual_input", "dual_residual_input")]
    np.testing.assert_allclose(t, ph["termination"], rtol=1e-12)
    s.close()


def test_hover_episode_sequence():
    """BASELINE config 2 golden: 882 iterations, the exact per-step sequence of SURVEY.md 8(c)."""
    suite, _ = sc.load_suite(os.path.join(GOLDEN, "hover_warm.npz"))
    prob, extra = sc.load_problem("quadrotor_20hz")
    s = sc.make_solver(OracleSolver, prob, suite["config"])
    s["Xref"] = np.tile(np.array(extra["hover"]["xref"], dtype=float).reshape(-1, 1), (1, prob["N"]))
    total, iters, u0, _ = s.closed_loop(extra["hover"]["x0"], 100)
    s.close()
    assert total == 882
    assert np.array_equal(iters, suite["episode"]["iters"])
    assert list(iters[:12]) == [100, 100, 100, 100, 100, 58, 43, 14, 7, 7, 7, 7]
    assert rel_err(u0, suite["episode"]["u0"]) < 1e-9
    np.testing.assert_allclose(u0[0], [0.516686858116, 0.508049172249, 0.521974339384, 0.528649689974], rtol=1e-10)


def test_tracking_episode_sequence():
    """examples/quadrotor_tracking.cpp: 725 iterations over 291 steps."""
    gold = np.load(os.path.join(GOLDEN, "tracking_episode.npz"))
    prob, extra = sc.load_problem("quadrotor_20hz")
    s = sc.make_solver(OracleSolver, prob, sc._hover_cfg(prob, extra))
    traj = np.array(extra["y_axis_line"])
    N, nx, nu = prob["N"], prob["nx"], prob["nu"]
    s["Xref"] = traj[0:N].T
    x0 = s["Xref"][:, 0].copy()
    its = []
    for k in range(301 - N):
        s["x"][:, 0] = x0
        s["Xref"] = traj[k:k + N].T
        s["y"] = np.zeros((nu, N - 1))
        s["g"] = np.zeros((nx, N))
        s.solve()
        its.append(int(s.get("sol_iter")))
        x0 = prob["A"] @ x0 + prob["B"] @ s["u"][:, 0]
    s.close()
    assert sum(its) == 725 and np.array_equal(its, gold["iters"])
    assert rel_err(x0, gold["x_final"]) < 1e-9


def test_edge_max_iter_zero_and_check_termination():
    suite = sc.tracking_random_suite(B=2)
    suite["config"]["max_iter"] = 0
    out = sc.run_cases(OracleSolver, suite)
    assert np.all(out["ret"] == 1) and np.all(out["iter"] == 0) and np.all(out["status"] == 11)
    suite["config"]["max_iter"] = 100
    suite["config"]["check_termination"] = 5
    out = sc.run_cases(OracleSolver, suite)
    assert np.all(out["iter"] % 5 == 0)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libtinympc_ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("maker", [
    lambda: sc.random_state_suite("quadrotor_20hz", B=4, seed=101),
    lambda: sc.random_state_suite("rocket_landing_20hz", B=4, seed=102, soc=True),
    lambda: sc.random_state_suite("cartpole", B=4, seed=103),
    lambda: sc.sweep_suite(8, 2, 10, B=2),
    lambda: sc.tracking_random_suite(B=4, seed=999),
])
def test_oracle_matches_live_reference(maker):
    suite = maker()
    assert_outputs_match(sc.run_cases(OracleSolver, suite), sc.run_cases(RefSolver, suite), RTOL, "live")


def test_oracle_matches_live_reference_on_the_fuzzed_space():
    """The random configurations the GPU fuzzers draw (tools/fuzz_parity.random_suite: every switch, cones, static and
    time-varying half-spaces, check_termination 1..4, warm states, 18 shapes) through the REAL reference (oracle/_ref)
    and the oracle: same iteration counts, fields within 1e-9.  Build container only."""
    import sys
    from cpu_solvers import RefSolver, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref is built only where /root/reference exists")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from fuzz_parity import random_suite
    for seed in range(1, 81):
        suite, _ = random_suite(seed)
        if suite["config"]["check_termination"] == 0:
            continue                                  # the reference divides by zero (admm.cpp:312)
        assert_outputs_match(sc.run_cases(OracleSolver, suite), sc.run_cases(RefSolver, suite), RTOL, f"fuzz seed {seed}")
