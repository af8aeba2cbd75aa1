"""Round 6: the one-row kernel's PREFETCH form (admm_kernel.hip.h PF) -- persistent waves that draw their tiles from a ticket counter and
whose NEXT tile's records travel into the wave's LDS buffer by LDS-DMA while the current tile iterates.  A launch form, not an
algorithm: every test holds it to the oracle (the restatement of admm.cpp:331-455) AND bit for bit to the plain form."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import scenarios as sc                      # noqa: E402
from cpu_solvers import OracleSolver        # noqa: E402
from hip_runner import make_batch, run_cases_hip   # noqa: E402
from test_gpu_parity import assert_match, RTOL     # noqa: E402

pytestmark = pytest.mark.gpu

FIELDS = ("iter", "sol_solved", "status", "x", "u", "vnew", "znew", "g", "y", "v", "z", "primal_residual_state", "primal_residual_input",
          "dual_residual_state", "dual_residual_input")


def took_prefetch(suite, opts, warm_fields=None):
    s = make_batch(suite)
    for k, v in opts.items():
        s.set_option(k, v)
    c = suite["cases"]
    s.set_x0(c["x0"]); s.set("Xref", c["Xref"]); s.set("Uref", c["Uref"])
    if warm_fields:
        for k in warm_fields:
            s.set(k, c[k])
    s.solve()
    took = s.get_option("last_prefetch")
    s.close()
    return took


@pytest.mark.parametrize("waves", [0, 1, 3, 16])
@pytest.mark.parametrize("B", [37, 203])
def test_prefetch_form_cold_and_warm_against_the_oracle_and_the_plain_form(B, waves):
    """Quadrotor tracking (per-instance reference records: they travel through the tile buffer), ragged batches, cold solve and a warm
    one from the oracle's state.  waves = 1 / 3 / 16: a persistent grid of that many waves (16: eight ticket shards), so that every wave walks many tiles and the
    ticket counter, the buffer hand-over and the in-flight stores of the tile before are all exercised; 0: one wave per tile."""
    suite = sc.tracking_random_suite(B=B, seed=11 + B)
    ref = sc.run_cases(OracleSolver, suite)
    base = {"no_tile": 1, "repack_after": 0}
    pf = dict(base, prefetch=1, prefetch_waves=waves)
    assert took_prefetch(suite, pf) == 1 and took_prefetch(suite, dict(base, prefetch=0)) == 0
    a = run_cases_hip(suite, options=pf)
    p = run_cases_hip(suite, options=dict(base, prefetch=0))
    assert_match(a, ref, RTOL, f"prefetch form, cold, B={B}, waves={waves}")
    for k in FIELDS:
        assert np.array_equal(a[k], p[k]), (k, B, waves)
    warm = dict(problem=suite["problem"], config=dict(suite["config"], max_iter=60), cases=dict(suite["cases"]))
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z"):
        warm["cases"][k] = ref[k]
    warm["cases"]["x0"] = suite["cases"]["x0"] * 0.9
    wref = sc.run_cases(OracleSolver, warm)
    w = run_cases_hip(warm, options=pf)
    assert_match(w, wref, RTOL, f"prefetch form, warm, B={B}, waves={waves}")
    wp = run_cases_hip(warm, options=dict(base, prefetch=0))
    for k in FIELDS:
        assert np.array_equal(w[k], wp[k]), (k, B, waves)
    assert len(np.unique(wref["iter"])) > 1


def test_prefetch_form_with_one_shared_reference_record():
    """Identical hover instances (BASELINE config 2's data): the reference record is shared -- formed once per wave, kept in LDS -- and
    v|z travels through the buffer.  Three successive warm solves from ONE handle (alternating launch direction, ticket counter never
    reset, x0 advanced on the device) against the same sequence on the plain form."""
    suite = sc.tracking_random_suite(B=131, seed=3)
    suite = dict(suite, cases={k: np.repeat(v[:1], 131, axis=0) for k, v in suite["cases"].items()})

    def episode(prefetch, waves):
        s = make_batch(suite)
        for k, v in (("no_tile", 1), ("repack_after", 0), ("prefetch", prefetch), ("prefetch_waves", waves), ("advance_x0", 1)):
            s.set_option(k, v)
        c = suite["cases"]
        s.set_x0(c["x0"])
        s.set("Xref", c["Xref"][0], broadcast=True)
        s.set("Uref", c["Uref"][0], broadcast=True)
        outs = []
        for _ in range(4):
            s.solve()
            assert s.get_option("last_prefetch") == prefetch
            outs.append({k: s.get(k) for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z", "x0")})
            outs[-1]["iter"] = np.asarray(s.status()["iter"])
        acc = np.asarray(s.reduce_stats()[7:9])
        s.close()
        return outs, acc
    plain, acc0 = episode(0, 0)
    for waves in (0, 2):
        pf, acc1 = episode(1, waves)
        assert np.array_equal(acc0, acc1)                     # the accumulated iteration / solve counters (atomics in the prefetch form)
        for a, b in zip(plain, pf):
            for k in a:
                assert np.array_equal(a[k], b[k]), (k, waves)


@pytest.mark.parametrize("dims", [(4, 2, 10), (8, 4, 10), (12, 2, 10)])
def test_prefetch_form_on_sweep_shapes_half_rows_and_split_solves(dims):
    """The config-5 cells the one-row kernel holds: the form of the plain and of the HALF kernel (nx+nu <= 8: eight instances per
    wave and tile), and as the first stage of a split solve (repack_after = 6: the follow-up stages walk index lists on the plain
    form) -- against the oracle and bit for bit against the plain launches."""
    suite = sc.sweep_suite(*dims, B=75, max_iter=120)
    ref = sc.run_cases(OracleSolver, suite)
    base = {"no_tile": 1}
    plain = run_cases_hip(suite, options=dict(base, prefetch=0, repack_after=0))
    assert_match(plain, ref, RTOL, f"plain {dims}")
    # (prefetch_static: percent of a wave's tiles it takes by grid stride -- by rule 75 for warm launches, 50 for cold ones -- before it draws tickets)
    for opts in (dict(prefetch=1, repack_after=0), dict(prefetch=1, repack_after=0, prefetch_waves=2), dict(prefetch=1, repack_after=6, prefetch_waves=2),
                 dict(prefetch=1, repack_after=0, half_rows=0, prefetch_waves=3), dict(prefetch=1, repack_after=0, prefetch_waves=2, prefetch_static=0),
                 dict(prefetch=1, repack_after=0, prefetch_waves=3, prefetch_static=100), dict(prefetch=1, repack_after=0, prefetch_waves=8, prefetch_static=50)):
        assert took_prefetch(suite, dict(base, **opts)) == 1, (dims, opts)
        o = run_cases_hip(suite, options=dict(base, **opts))
        for k in FIELDS:
            assert np.array_equal(o[k], plain[k]), (k, dims, opts)
    assert len(np.unique(ref["iter"])) > 3


def test_prefetch_rule_and_what_does_not_take_the_form():
    """Option prefetch = -1 (default): a small batch stays on the plain form (nothing to prefetch: one tile per wave); fused steps,
    a capped grid and prefetch = 0 never take it; prefetch = 1 forces it wherever the variant has it."""
    suite = sc.tracking_random_suite(B=64, seed=5)
    assert took_prefetch(suite, {"no_tile": 1}) == 0
    assert took_prefetch(suite, {"no_tile": 1, "prefetch": 1}) == 1
    assert took_prefetch(suite, {"no_tile": 1, "prefetch": 1, "steps_per_launch": 3}) == 0
    assert took_prefetch(suite, {"no_tile": 1, "prefetch": 1, "grid_waves_per_cu": 2}) == 0
    assert took_prefetch(suite, {"no_tile": 1, "prefetch": 0}) == 0


def _kernel_dims_shapes():
    shapes = []
    with open(os.path.join(os.path.dirname(HERE), "tinympc_amd", "csrc", "kernel_dims.txt")) as f:
        for line in f:
            w = line.split("#")[0].split()
            if len(w) >= 3:
                shapes.append((int(w[0]), int(w[1]), int(w[2])))
    return shapes


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dims", _kernel_dims_shapes())
def test_default_dispatch_of_every_compiled_in_shape_at_a_batch_the_rule_applies_to(dims):
    """Every kernel_dims.txt shape, a batch large enough for the PREFETCH rule (several tiles per resident wave, cold launch), the
    library's DEFAULT dispatch, three cold solves in a row (plain probe, split probe, whatever the clock then keeps): the launches
    finish -- round 6 found the one long-horizon shape with the form, (8,2,30), hanging in its ticketed tiles; the form is now
    confined to the two-waves-per-SIMD shapes -- and leave the bits of the plain launch (prefetch = 0, no split, no tile form)."""
    import tinympc_amd as tm
    nx, nu, N = dims
    B = 40000
    prob, rng = tm.random_problem(nx, nu, N)
    x0 = rng.uniform(-1, 1, (B, nx))
    xr = np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1)), N, axis=2)

    def run(opts, n):
        s = tm.TinyBatchSolver.from_problem(prob, B)
        s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
        s.update_settings(max_iter=60)
        for k, v in opts.items():
            s.set_option(k, v)
        s.set_x0(x0); s.set_x_ref(xr)
        took = 0
        for _ in range(n):
            s.reset()
            s.solve_async()
            s.synchronize()
            took |= int(s.get_option("last_prefetch"))
        out = (s.status()["iter"].copy(), s.get("u").copy(), s.get("y").copy())
        s.close()
        return out, took
    plain, took0 = run({"prefetch": 0, "repack_after": 0, "no_tile": 1, "plan": 0}, 1)
    auto, took = run({"plan": 0}, 3)
    assert took0 == 0
    if N > 12:
        assert took == 0, "the PREFETCH form is confined to the two-waves-per-SIMD shapes"
    for a, b in zip(plain, auto):
        assert np.array_equal(a, b), dims
