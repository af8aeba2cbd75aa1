"""GPU: adaptive rho (admm.cpp:397-423, rho_benchmark.cpp:14-249) on the one-row kernel.

The goldens (tests/golden/adaptive_*.npz) come from the REAL reference with the stack under solve() scrubbed
(oracle/ref_shim.cpp: upstream reads an uninitialised RhoAdapter flag, profiles/r02_adaptive_rho_probe.txt); they are
checked by name in test_gpu_parity.py like every other suite.  Here: the whole closed-loop episode with the cache state
persisting on the device from step to step, per-instance divergence of rho inside one batch, and reset semantics."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scenarios as sc  # noqa: E402
import tinympc_amd as tm  # noqa: E402
from cpu_solvers import OracleSolver  # noqa: E402
from hip_runner import make_batch, run_cases_hip  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def test_hover_episode_with_adaptive_rho_reproduces_the_reference():
    """100 closed-loop steps, one launch per step, rho / Kinf / Pinf carried on the device: the per-step iteration counts
    and rho values of the real reference (813 iterations in total, rho 5 -> 2.443 -> 1.165 -> 1.0)."""
    suite, _ = sc.load_suite(os.path.join(GOLDEN, "adaptive_hover.npz"))
    ep = suite["episode"]
    prob, cfg = suite["problem"], suite["config"]
    _, extra = sc.load_problem("quadrotor_20hz")
    h = extra["hover"]
    nx, N, B = prob["nx"], prob["N"], 8
    s = make_batch(suite, batch=B)
    s.set_option("advance_x0", 1)
    s.set_x_ref(np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N)), broadcast=True)
    s.set_x0(np.array(h["x0"], dtype=np.float64), broadcast=True)
    its, rhos = [], []
    for _ in range(len(ep["iters"])):
        s.solve()
        st = s.status()
        assert np.all(st["iter"] == st["iter"][0]) and np.all(st["solved"] == st["solved"][0])
        its.append(int(st["iter"][0]) * (1 if st["solved"][0] else -1))
        r = s.get_cache_state("rho")
        assert np.all(r == r[0])
        rhos.append(float(r[0]))
    assert its == ep["iters"].tolist() and int(np.abs(its).sum()) == 813
    assert np.allclose(rhos, ep["rho"], rtol=1e-9, atol=0.0)
    assert rel_err(s.get("x0")[0], ep["x_final"]) < 1e-6
    # fused: the same episode in ONE launch (rho adapts inside the launch, per solve)
    s.reset()
    s.set_x0(np.array(h["x0"], dtype=np.float64), broadcast=True)
    s.set_option("steps_per_launch", len(ep["iters"]))
    s.solve()
    assert s.reduce_stats()[7] == 813 * B
    assert np.allclose(s.get_cache_state("rho"), ep["rho"][-1], rtol=1e-9)
    s.close()


@pytest.mark.parametrize("clip", [1, 0])
def test_per_instance_rho_paths_in_one_batch_match_the_oracle(clip):
    suite = sc.tracking_adaptive_suite(B=203, seed=9000 + clip, rho_min=0.7, rho_max=30.0, clip=clip, max_iter=80)
    ref = sc.run_cases(OracleSolver, suite)
    out = run_cases_hip(suite)
    assert np.array_equal(out["iter"].astype(int), ref["iter"].astype(int))
    assert np.array_equal(out["sol_solved"].astype(int), ref["sol_solved"].astype(int))
    assert len(np.unique(np.round(ref["rho"], 6))) > 20                  # every instance went its own way
    assert np.allclose(out["rho"], ref["rho"], rtol=1e-9, atol=0.0)
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z", "Kinf", "Pinf", "C1", "C2"):
        for b in range(out[k].shape[0]):
            assert rel_err(out[k][b], ref[k][b]) < 1e-9, (k, b)


def test_long_solves_take_more_rho_steps_than_the_c1_c2_log_holds():
    """C1 / C2 take their Taylor steps lazily, 32 logged rho steps at a time (flush_c in admm_kernel.hip.h): tolerances nobody
    meets keep every instance iterating to max_iter = 300 -> 59 adaptations, one flush inside the loop and one at the end; the
    cache must come out as the oracle's one-step-at-a-time updates leave it."""
    suite = sc.tracking_adaptive_suite(B=37, seed=4242, rho_min=0.7, rho_max=30.0, clip=1, max_iter=300)
    suite["config"] = dict(suite["config"], abs_pri_tol=0.0, abs_dua_tol=0.0)     # (residual < 0 never holds)
    ref = sc.run_cases(OracleSolver, suite)
    out = run_cases_hip(suite)
    assert np.all(ref["iter"].astype(int) == 300) and np.array_equal(out["iter"].astype(int), ref["iter"].astype(int))
    assert np.allclose(out["rho"], ref["rho"], rtol=1e-9, atol=0.0)
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z", "Kinf", "Pinf", "C1", "C2"):
        for b in range(out[k].shape[0]):
            assert rel_err(out[k][b], ref[k][b]) < 1e-9, (k, b)


def test_reset_restores_the_setup_cache_and_off_means_off():
    suite = sc.tracking_adaptive_suite(B=12, seed=31)
    first = run_cases_hip(suite)
    s = make_batch(suite)
    cases = suite["cases"]
    for rep in range(2):                                  # solve, reset, solve again: identical
        s.reset()
        s.set_x0(cases["x0"]); s.set_x_ref(cases["Xref"]); s.set_u_ref(cases["Uref"])
        s.solve()
        assert np.array_equal(s.status()["iter"], first["iter"].astype(int))
        assert np.array_equal(s.get_cache_state("rho"), first["rho"])
    # adaptive off: the plain kernel, the family's cache, the plain suite's results
    plain = run_cases_hip(dict(suite, config={k: v for k, v in suite["config"].items() if not k.startswith(("adaptive", "sensitivity"))}))
    s.set_adaptive_rho(0)
    s.reset()
    s.set_x0(cases["x0"]); s.set_x_ref(cases["Xref"]); s.set_u_ref(cases["Uref"])
    s.solve()
    assert np.array_equal(s.status()["iter"], plain["iter"].astype(int))
    assert np.array_equal(s.get("x"), plain["x"])
    s.close()
