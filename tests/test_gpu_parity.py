"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
  (1) the golden fixtures generated from the REAL reference (tests/golden/, oracle/gen_golden.py),
  (2) the plain-C oracle on fresh seeded inputs,
  (3) size-independent properties at BASELINE.json's full batch sizes.

Bar (BASELINE.json north_star): x/u trajectories within 1e-5 relative of the reference Eigen CPU path
and IDENTICAL iteration counts.  The kernel differs from the reference only in floating-point
association (FMA contraction, fused Quu_inv*B' table), so the tests assert a much tighter 1e-9.
"""
import glob
import os

import numpy as np
import pytest

import scenarios as sc
from cpu_solvers import OracleSolver
from hip_runner import make_batch, run_cases_hip

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUITES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                if os.path.basename(p)[:-4] not in ("cache_kat", "project_soc_kat", "phase_kat", "tracking_episode"))
CONTRACT_RTOL = 1e-5     # BASELINE.json
RTOL = 1e-9              # what we actually hold the kernel to


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def supported(suite):
    """Every shape with nx + nu <= 32 runs: register-resident kernel where instantiated, else the coverage kernel."""
    p = suite["problem"]
    return p["nx"] + p["nu"] <= 32


def assert_match(out, ref, rtol, what):
    for k in ("iter", "status", "sol_solved"):
        assert np.array_equal(out[k].astype(int), ref[k].astype(int)), f"{what}: {k}: {out[k]} vs {ref[k]}"
    worst = 0.0
    for k, v in ref.items():
        if v.ndim >= 2 and k in out:
            for b in range(v.shape[0]):
                e = rel_err(out[k][b], v[b])
                worst = max(worst, e)
                assert e <= rtol, f"{what}: field {k} case {b} rel err {e:.3e} > {rtol}"
    for k in ("primal_residual_state", "primal_residual_input", "dual_residual_state", "dual_residual_input"):
        assert np.allclose(out[k], ref[k], rtol=1e-6, atol=1e-11), f"{what}: {k}: {out[k]} vs {ref[k]}"
    if "rho" in ref:                                  # adaptive rho: cache->rho after the solve (Kinf, Pinf, C1, C2 are matrices: above)
        assert "rho" in out and np.allclose(out["rho"], ref["rho"], rtol=rtol, atol=0.0), f"{what}: rho {out['rho']} vs {ref['rho']}"
        for k in ("Kinf", "Pinf", "C1", "C2"):
            assert k in out, k
    return worst


@pytest.mark.parametrize("name", SUITES)
def test_hip_matches_reference_golden(name):
    suite, ref = sc.load_suite(os.path.join(GOLDEN, name + ".npz"))
    if not supported(suite):
        pytest.skip("(nx,nu,N) not instantiated in this round (SURVEY.md section 7 step 6)")
    out = run_cases_hip(suite, debug=name.startswith("random_state"))
    worst = assert_match(out, ref, RTOL, name)
    assert worst <= CONTRACT_RTOL


@pytest.mark.parametrize("name", ["sweep_20_4_10_isoc", "sweep_12_8_30_isoc", "sweep_8_4_50_bothsoc"])
def test_cone_goldens_of_wide_and_long_shapes_run_on_the_tile_kernel(name):
    """the reference-generated cone goldens of the tile shapes (VERDICT r04 nit) are served by the tile kernel's cone variant, not by
    the coverage kernel -- test_hip_matches_reference_golden holds that path to them"""
    from hip_runner import make_batch
    suite, _ = sc.load_suite(os.path.join(GOLDEN, name + ".npz"))
    s = make_batch(suite)
    assert s.kernel_path() == "tile", s.kernel_path()
    s.close()


@pytest.mark.parametrize("name", SUITES)
def test_coverage_kernel_matches_reference_golden(name):
    """force_general = 1: the shape-agnostic coverage kernel (general_kernel.hip.h) on the same fixtures."""
    suite, ref = sc.load_suite(os.path.join(GOLDEN, name + ".npz"))
    out = run_cases_hip(suite, options={"force_general": 1})
    assert_match(out, ref, RTOL, name + " [general]")


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_dpp_modes_agree_with_golden(mode):
    """dpp_mode 0 = fused v_fmac_f64_dpp row_newbcast (two accumulator chains), 2 = one chain,
    1 = compiler-scheduled v_mov_b32_dpp + v_fma_f64: same results."""
    suite, ref = sc.load_suite(os.path.join(GOLDEN, "random_state_quad.npz"))
    out = run_cases_hip(suite, options={"dpp_mode": mode})
    assert_match(out, ref, RTOL, f"dpp_mode={mode}")


@pytest.mark.parametrize("maker", [
    lambda: sc.random_state_suite("quadrotor_20hz", B=37, seed=201),
    lambda: sc.random_state_suite("rocket_landing_20hz", B=21, seed=202, soc=True),
    lambda: sc.random_state_suite("cartpole", B=9, seed=203),
    lambda: sc.random_state_suite("quadrotor_20hz", B=11, seed=204, soc="overlap"),        # cones sharing rows: sequential projections
    lambda: sc.random_state_suite("rocket_landing_20hz", B=7, seed=205, soc="overlap"),
    lambda: sc.tracking_random_suite(B=130, seed=4242),
    lambda: sc.rocket_random_suite(B=33, seed=77),
    lambda: sc.sweep_suite(8, 2, 10, B=5),
    lambda: sc.sweep_suite(4, 4, 10, B=5),
    lambda: sc.sweep_suite(8, 8, 10, B=3),
    # shapes / constraint families served by the coverage kernel
    lambda: sc.sweep_suite(12, 8, 10, B=3),
    lambda: sc.sweep_suite(20, 8, 30, B=2, max_iter=120),
    lambda: sc.sweep_suite(20, 4, 50, B=2, max_iter=60),
    lambda: sc.sweep_suite(4, 2, 50, B=5, max_iter=200),
    lambda: sc.random_linear_suite("quadrotor_20hz", B=9, seed=301),
    lambda: sc.random_linear_suite("rocket_landing_20hz", B=5, seed=302, soc=True),
    lambda: sc.random_linear_suite("quadrotor_20hz", B=3, seed=303, tv=False, box=False),
])
def test_hip_matches_oracle_seeded(maker):
    """Ragged batch sizes (not multiples of 4 -> partially filled wavefronts), divergent iteration counts."""
    suite = maker()
    ref = sc.run_cases(OracleSolver, suite)
    out = run_cases_hip(suite, debug=True)
    assert_match(out, ref, RTOL, "seeded")


@pytest.mark.parametrize("dims,B,max_iter", [((12, 8, 10), 7, 150), ((20, 2, 10), 5, 150), ((20, 4, 10), 6, 150), ((20, 8, 10), 9, 200),
                                              ((4, 2, 50), 9, 200), ((12, 4, 50), 6, 120), ((8, 8, 50), 5, 120), ((4, 4, 50), 9, 120), ((4, 8, 50), 5, 120), ((8, 2, 50), 6, 120),
                                              ((8, 4, 50), 5, 120), ((12, 2, 50), 5, 120), ((12, 8, 50), 3, 100), ((20, 2, 50), 3, 100), ((20, 2, 30), 3, 100), ((20, 4, 30), 3, 100),
                                              ((12, 8, 30), 3, 120), ((20, 8, 30), 3, 120), ((20, 4, 50), 3, 80), ((20, 8, 50), 2, 80)])
def test_tile_kernel_matches_oracle(dims, B, max_iter):
    """Wide (nx+nu > 16: W = 2 rows across the knot vector) and long (R = 2 rows along the horizon) shapes on the
    register-resident tile kernel (tile_kernel.hip.h); ragged batches leave part of the last wavefront empty."""
    import tinympc_amd as tm
    suite = sc.sweep_suite(*dims, B=B, max_iter=max_iter)
    probe = make_batch(suite)
    assert probe.kernel_path() == "tile", probe.kernel_path()
    probe.close()
    ref = sc.run_cases(OracleSolver, suite)
    out = run_cases_hip(suite)
    assert_match(out, ref, RTOL, f"tile {dims}")
    warm = dict(problem=suite["problem"], config=suite["config"], cases=dict(suite["cases"]))   # second solve from the warm state
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z"):
        warm["cases"][k] = ref[k]
    warm["cases"]["x0"] = suite["cases"]["x0"] * 0.8
    assert_match(run_cases_hip(warm), sc.run_cases(OracleSolver, warm), RTOL, f"tile warm {dims}")


@pytest.mark.parametrize("dims", [(4, 2, 30), (4, 4, 30)])
@pytest.mark.parametrize("dyn", [0, 1])
def test_half_row_form_matches_oracle(dims, dyn):
    """Round 3: shapes with nx+nu <= 8 can run TWO instances per DPP row (tile kernel, W = 0: 8 instances per wave, every column of a
    mat-vec as a pair of bank-masked `v_fmac_f64_dpp row_newbcast` -- low halves first, one wait state, high halves, because a
    bank-masked DPP op re-writes its disabled lanes with a vdst value it read WITHOUT interlock, tools/ubench/ubench_dpp_bankmask2.hip).
    Ragged batch (37 = 4 full waves + 5), cold and warm solves, static tiles and dynamic slots, against the oracle."""
    suite = sc.sweep_suite(*dims, B=37, max_iter=300)
    opts = {"prefer_tile": 1, "tile_w": 0, "tile_dyn": dyn}
    s = make_batch(suite)
    for k, v in opts.items():
        s.set_option(k, v)
    assert s.kernel_path() == "tile"
    s.close()
    ref = sc.run_cases(OracleSolver, suite)
    out = run_cases_hip(suite, options=opts)
    assert_match(out, ref, RTOL, f"half rows {dims}")
    assert len(np.unique(ref["iter"])) > 3
    warm = dict(problem=suite["problem"], config=suite["config"], cases=dict(suite["cases"]))
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z"):
        warm["cases"][k] = ref[k]
    warm["cases"]["x0"] = suite["cases"]["x0"] * 0.7
    assert_match(run_cases_hip(warm, options=opts), sc.run_cases(OracleSolver, warm), RTOL, f"half rows warm {dims}")


@pytest.mark.parametrize("dims", [(4, 2, 10), (4, 4, 10), (2, 2, 3), (4, 2, 30)])
def test_one_row_kernel_half_rows_match_the_oracle_and_the_full_row_form(dims):
    """Round 4: shapes with nx+nu <= 8 run TWO instances per DPP row on the one-row kernel itself (HALF: eight per wave, every column
    of a mat-vec a pair of bank-masked FMAs inside the fused step blocks).  Ragged batch (37 = 4 full waves + 5), cold, warm and
    out-of-iterations solves against the oracle, and bit for bit against the one-instance-per-row form (option half_rows = 0)."""
    suite = sc.sweep_suite(*dims, B=37, max_iter=300)
    base = {"no_tile": 1, "repack_after": 0}
    ref = sc.run_cases(OracleSolver, suite)
    half = run_cases_hip(suite, options=dict(base, half_rows=1))
    full = run_cases_hip(suite, options=dict(base, half_rows=0))
    assert half["half_rows"] == 1 and full["half_rows"] == 0
    assert_match(half, ref, RTOL, f"one-row half rows {dims}")
    for k in ("iter", "sol_solved", "x", "u", "vnew", "znew", "g", "y", "v", "z", "primal_residual_state", "dual_residual_input"):
        assert np.array_equal(half[k], full[k]), (k, dims)
    warm = dict(problem=suite["problem"], config=dict(suite["config"], max_iter=6), cases=dict(suite["cases"]))
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z"):
        warm["cases"][k] = ref[k]
    warm["cases"]["x0"] = suite["cases"]["x0"] * 0.7
    wref = sc.run_cases(OracleSolver, warm)
    whalf = run_cases_hip(warm, options=dict(base, half_rows=1))
    assert whalf["half_rows"] == 1
    assert_match(whalf, wref, RTOL, f"one-row half rows, warm {dims}")


@pytest.mark.parametrize("dims", [(4, 2, 10), (4, 4, 10)])
def test_one_row_kernel_half_rows_in_fused_steps_and_split_solves(dims):
    """The HALF form under the launch modes the one-row kernel has: T closed-loop MPC steps fused into one launch against T launches,
    and a split solve (repack_after = K: the open instances re-packed EIGHT per wave) against the plain one -- all bit for bit."""
    T = 5
    suite = sc.sweep_suite(*dims, B=43, max_iter=40)

    def run(fused, half, repack=0):
        s = make_batch(suite)
        for k, v in (("no_tile", 1), ("repack_after", repack), ("half_rows", half), ("advance_x0", 1)):
            s.set_option(k, v)
        s.set_x0(suite["cases"]["x0"]); s.set("Xref", suite["cases"]["Xref"]); s.set("Uref", suite["cases"]["Uref"])
        if fused:
            s.set_option("steps_per_launch", T)
            s.solve_async()
        else:
            for _ in range(T):
                s.solve_async()
        out = {k: s.get(k) for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z", "x0")}
        out["iter"] = np.asarray(s.status()["iter"])
        out["acc"] = np.asarray(s.reduce_stats()[7:9])
        assert s.get_option("last_half_rows") == half
        s.close()
        return out
    a = run(False, 0)
    assert a["acc"][0] > 0
    for other in (run(False, 1), run(True, 1), run(True, 0)):
        for k in a:
            assert np.array_equal(a[k], other[k]), (k, dims)
    # one cold solve, split at K = 8 (stages 8, 16, 32, 40), against the plain launch
    def cold(half, repack):
        o = run_cases_hip(dict(suite, config=dict(suite["config"], max_iter=120)), options={"no_tile": 1, "repack_after": repack, "half_rows": half})
        assert o["half_rows"] == half
        return o
    p0, p1, s1 = cold(0, 0), cold(1, 0), cold(1, 8)
    assert len(np.unique(p0["iter"])) > 3 and p0["iter"].max() > 16
    for k in ("iter", "sol_solved", "x", "u", "vnew", "znew", "g", "y", "v", "z"):
        assert np.array_equal(p0[k], p1[k]) and np.array_equal(p0[k], s1[k]), (k, dims)


@pytest.mark.parametrize("dims,lm,other", [((12, 2, 50), 22, 0), ((4, 2, 50), 23, 99), ((12, 4, 50), 23, 0), ((20, 4, 30), 20, 12), ((20, 8, 50), 23, 14), ((20, 8, 50), 54, 14), ((4, 4, 50), 54, 99), ((8, 2, 50), 54, 0)])
@pytest.mark.parametrize("dyn", [0, 1])
def test_tile_forms_that_keep_v_in_its_record_match_the_oracle(dims, lm, other, dyn):
    """Round 3, LM bit 4: v|z (work->v, the slack of the iteration before) is held neither in registers nor in LDS -- the slot update
    streams the old vnew|znew to the instance's v|z record, a solve's first iteration reads the record back, a solve that runs out of
    iterations writes vnew over it (admm.cpp:431-446).  Cold, warm (v != vnew on entry: the first iteration's dual residual uses the
    stored v) and out-of-iterations solves, ragged batch, against the oracle and bit for bit against the form that keeps v in LDS."""
    suite = sc.sweep_suite(*dims, B=11, max_iter=120)
    opts = {"tile_lm": lm, "tile_dyn": dyn}
    ref = sc.run_cases(OracleSolver, suite)
    out = run_cases_hip(suite, options=opts)
    assert out["tile_form"] % 1000 == lm, out["tile_form"]     # the entry asked for is the one that ran
    assert_match(out, ref, RTOL, f"v in its record {dims}")
    assert 0 < ref["sol_solved"].sum() < len(ref["sol_solved"]) or ref["iter"].max() == 120 or len(np.unique(ref["iter"])) > 2
    alt = run_cases_hip(suite, options={"tile_lm": other, "tile_dyn": dyn} if other else {"tile_lm": 0, "tile_r": 2, "tile_dyn": dyn})
    assert alt["tile_form"] % 1000 == other and alt["tile_form"] != out["tile_form"]     # (a form that holds v|z in LDS or registers; 99 = the shape's automatic LDS set)
    for k in ("iter", "sol_solved", "x", "u", "vnew", "znew", "g", "y", "v", "z"):
        assert np.array_equal(out[k], alt[k]), (k, dims)
    warm = dict(problem=suite["problem"], config=dict(suite["config"], max_iter=7), cases=dict(suite["cases"]))
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z"):
        warm["cases"][k] = ref[k]
    warm["cases"]["x0"] = suite["cases"]["x0"] * 0.7
    wref = sc.run_cases(OracleSolver, warm)
    assert (wref["sol_solved"] == 0).any()                    # some run out of their 7 iterations: v = vnew on exit
    assert_match(run_cases_hip(warm, options=opts), wref, RTOL, f"v in its record, warm {dims}")


@pytest.mark.parametrize("dims", [(4, 2, 50), (20, 2, 50)])
def test_fused_steps_on_a_tile_form_that_keeps_v_in_its_record(dims):
    """Closed loop on the long tile forms (v|z in its HBM record): T MPC steps fused into one launch leave exactly what T launches
    leave -- every solve's first iteration must see the v|z its predecessor left (converged: the slack of ITS last but one
    iteration; out of iterations: vnew), whether the predecessor ran in the same launch or the one before."""
    T = 4
    suite = sc.sweep_suite(*dims, B=9, max_iter=25)

    def run(fused):
        s = make_batch(suite)
        assert s.kernel_path() == "tile"
        s.set_option("advance_x0", 1)
        s.set_x0(suite["cases"]["x0"]); s.set("Xref", suite["cases"]["Xref"]); s.set("Uref", suite["cases"]["Uref"])
        if fused:
            s.set_option("steps_per_launch", T)
            s.solve_async()
        else:
            for _ in range(T):
                s.solve_async()
        out = {k: s.get(k) for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z", "x0")}
        out["iter"] = np.asarray(s.status()["iter"])
        out["acc"] = np.asarray(s.reduce_stats()[7:9])
        s.close()
        return out
    a, b = run(False), run(True)
    assert a["acc"][0] > 0 and 0 < a["acc"][1] < 9 * T             # some solves converge, some run out of their 25 iterations
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("dims", [(4, 2, 30), (8, 2, 10)])
def test_one_row_shape_forms_are_interchangeable_with_an_affine_term(dims):
    """The launch forms a one-row shape can take (one-row kernel; tile kernel in its one-row layout; half rows) are swapped by the
    dispatcher from solve to solve, so they must agree BIT FOR BIT -- also with fdyn != 0, where the placement of the forward
    constant decides the rounding ((f + A x) + B u in all of them)."""
    suite = sc.sweep_suite(*dims, B=29, max_iter=200)
    rng = np.random.default_rng(7)
    suite["problem"] = dict(suite["problem"], f=rng.normal(0, 0.05, dims[0]))
    ref = sc.run_cases(OracleSolver, suite)
    one_row = run_cases_hip(suite)
    assert_match(one_row, ref, RTOL, f"affine {dims}")
    forms = [{"prefer_tile": 1, "tile_w": 1, "tile_dyn": 0}, {"prefer_tile": 1, "tile_w": 1, "tile_dyn": 1}]
    if dims == (4, 2, 30):
        forms += [{"prefer_tile": 1, "tile_w": 0, "tile_dyn": 0}, {"prefer_tile": 1, "tile_w": 0, "tile_dyn": 1}]
    for o in forms:
        out = run_cases_hip(suite, options=o)
        for k in ("iter", "sol_solved", "x", "u", "vnew", "znew", "g", "y", "v", "z"):
            assert np.array_equal(out[k], one_row[k]), (k, o)


def test_linear_constraints_register_resident_vs_coverage():
    """The LIN variants of the one-row kernel (half-space projections as DPP broadcast-FMA row sums) and the coverage
    kernel must agree with the oracle AND the fast path must really be the one that ran."""
    suite = sc.random_linear_suite("quadrotor_20hz", B=13, seed=77)
    ref = sc.run_cases(OracleSolver, suite)
    probe = make_batch(suite)
    assert probe.kernel_path() == "regs"
    probe.set_option("force_general", 1)
    assert probe.kernel_path() == "cover"
    probe.close()
    assert_match(run_cases_hip(suite), ref, RTOL, "LIN regs")
    assert_match(run_cases_hip(suite, options={"force_general": 1}), ref, RTOL, "LIN cover")
    for tv in (False, True):                                  # the reference's own linear-constraint examples
        cfg = sc.linear_example_cfg(suite["problem"], tv, 3)
        ex = dict(problem=suite["problem"], config=cfg, cases=suite["cases"])
        assert_match(run_cases_hip(ex), sc.run_cases(OracleSolver, ex), RTOL, f"example tv={tv}")


def test_persistent_grid_and_replication():
    """Same cases tiled 64x (4 instances per wave, many waves, persistent grid-stride tiles) must give
    bit-identical results to the single copy: no cross-instance interference."""
    suite, ref = sc.load_suite(os.path.join(GOLDEN, "tracking_random.npz"))
    one = run_cases_hip(suite)
    many = run_cases_hip(suite, replicate=64, options={"grid_waves_per_cu": 1})
    B = suite["cases"]["x0"].shape[0]
    for k in ("x", "u", "vnew", "g", "v"):
        for r in range(64):
            assert np.array_equal(many[k][r * B:(r + 1) * B], one[k]), k
    assert np.array_equal(many["iter"][:B], one["iter"])


def test_edge_cases():
    suite = sc.tracking_random_suite(B=3)
    suite["config"]["max_iter"] = 0                      # loop never runs: solution = loaded slack, ret 1, status 11
    out = run_cases_hip(suite)
    assert np.all(out["iter"] == 0) and np.all(out["status"] == 11) and out["batch_ret"] == 1
    suite["config"]["max_iter"] = 100
    suite["config"]["check_termination"] = 5
    ref = sc.run_cases(OracleSolver, suite)
    out = run_cases_hip(suite)
    assert_match(out, ref, RTOL, "check_termination=5")
    assert np.all(out["iter"] % 5 == 0)
    suite["config"]["check_termination"] = 1
    suite["config"]["en_state_bound"] = 0                # disabled boxes behave as (-inf, inf)
    suite["config"]["en_input_bound"] = 0
    ref = sc.run_cases(OracleSolver, suite)
    out = run_cases_hip(suite)
    assert_match(out, ref, RTOL, "bounds off")
    one = sc.tracking_random_suite(B=1)                  # batch of one
    assert_match(run_cases_hip(one), sc.run_cases(OracleSolver, one), RTOL, "B=1")


@pytest.mark.parametrize("options", [{}, {"prefer_tile": 1}, {"force_general": 1}])
def test_non_finite_instance_stays_in_its_row(options):
    """An instance fed NaN / Inf shares its wave with three healthy ones: it must terminate (iteration cap at the latest)
    and the neighbours' results must be bit-identical to a run without it -- on all three kernels."""
    suite = sc.tracking_random_suite(B=9)
    suite["config"]["max_iter"] = 60
    clean = run_cases_hip(suite, options=options)
    bad = {k: v.copy() for k, v in suite["cases"].items()}
    bad["x0"][1, 0] = np.nan
    bad["x0"][6, 2] = np.inf
    bad["Xref"][4] = np.nan
    dirty = run_cases_hip(dict(suite, cases=bad), options=options)
    healthy = [0, 2, 3, 5, 7, 8]
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z"):
        assert np.array_equal(dirty[k][healthy], clean[k][healthy]), k
    assert np.array_equal(dirty["iter"][healthy], clean["iter"][healthy])
    assert np.all(dirty["iter"][[1, 4, 6]] <= 60) and np.all(np.isin(dirty["status"][[1, 4, 6]], (1, 11)))
    assert np.isnan(dirty["x"][1]).any() and np.isnan(dirty["x"][4]).any()


@pytest.mark.parametrize("dims", [(4, 2, 50), (12, 2, 50)])
def test_a_diverged_instance_does_not_reach_the_others_through_the_shared_pad(dims):
    """on the tile forms that stream v|z to its record, the lanes WITHOUT a row (and the input lanes' dummy slot 0) of every
    instance point at one pad behind the records.  A diverged instance leaves non-finite values there; healthy instances -- in the same
    launch and in the NEXT one on the same batch -- must keep their iteration counts (check_termination = 1: their first test would
    otherwise see an infinite dual residual) and their results, bit for bit."""
    suite = sc.sweep_suite(*dims, B=13, max_iter=60)
    s = make_batch(suite)
    assert s.kernel_path() == "tile"

    def solve(x0):
        s.reset()
        s.set("Xref", suite["cases"]["Xref"]); s.set("Uref", suite["cases"]["Uref"]); s.set_x0(x0)
        s.solve()
        out = {k: s.get(k) for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z")}
        out["iter"] = np.asarray(s.status()["iter"]).copy()
        return out
    clean = solve(suite["cases"]["x0"])
    assert s.get_option("last_tile_form") % 1000 >= 16            # a form with v|z in its record (LM bit 4)
    bad = suite["cases"]["x0"].copy()
    bad[3, 0] = np.nan
    bad[8, 1] = np.inf
    dirty = solve(bad)
    after = solve(suite["cases"]["x0"])                            # the pad now holds what the diverged instances left
    s.close()
    healthy = [i for i in range(13) if i not in (3, 8)]
    for k in clean:
        assert np.array_equal(dirty[k][healthy], clean[k][healthy]), (k, "same launch")
        assert np.array_equal(after[k], clean[k]), (k, "next launch")


def test_hover_closed_loop_full_batch():
    """BASELINE config 2 at full size: 65 536 identical quadrotor-hover instances, 100 closed-loop MPC
    steps on device (advance_x0).  Properties: every instance reproduces the reference's golden
    iteration sequence (882 total) and all instances stay bit-identical to each other."""
    suite, _ = sc.load_suite(os.path.join(GOLDEN, "hover_warm.npz"))
    prob, extra = sc.load_problem("quadrotor_20hz")
    B = 65536
    s = make_batch(suite, batch=B)
    s.set_option("advance_x0", 1)
    s.set_x_ref(np.tile(np.array(extra["hover"]["xref"], dtype=float).reshape(-1, 1), (1, prob["N"])), broadcast=True)
    s.set_x0(np.array(extra["hover"]["x0"], dtype=float), broadcast=True)
    gold = suite["episode"]["iters"]
    total = 0
    for k in range(100):
        s.solve_async()
        if k in (0, 5, 6, 7, 20, 60, 99):
            st = s.reduce_stats()
            assert st[0] == float(gold[k]) * B, (k, st[0] / B, gold[k])
            u = s.get("u")[:, :, 0]
            assert np.all(u == u[0]), f"instances diverged at step {k}"
            # closed loop: per-solve differences (~1e-15) compound through the plant and u -> 0 near hover,
            # so the trajectory-level bar here is the contract's 1e-5 (single-solve parity is held to 1e-9 above)
            assert rel_err(u[0], suite["episode"]["u0"][k]) < (RTOL if k < 10 else 1e-6)
        total += gold[k]
    assert total == 882
    s.close()


@pytest.mark.parametrize("T", [100, 10])
def test_fused_closed_loop_steps(T):
    """steps_per_launch = T: T closed-loop MPC steps (solve, plant step, solve, ...) inside ONE launch
    with the ADMM state held in registers.  Must reproduce the reference's per-step iteration
    sequence (100 100 100 100 100 58 43 14 7 ... total 882) and applied controls, and leave the same
    warm state behind as 100 separate launches."""
    suite, _ = sc.load_suite(os.path.join(GOLDEN, "hover_warm.npz"))
    prob, extra = sc.load_problem("quadrotor_20hz")
    B = 1000                      # not a multiple of 4*anything special: ragged last wave
    xref = np.tile(np.array(extra["hover"]["xref"], dtype=float).reshape(-1, 1), (1, prob["N"]))
    x0 = np.array(extra["hover"]["x0"], dtype=float)
    gold = suite["episode"]
    # reference run: one launch per step
    a = make_batch(suite, batch=B)
    a.set_option("advance_x0", 1)
    a.set_x_ref(xref, broadcast=True)
    a.set_x0(x0, broadcast=True)
    for _ in range(100):
        a.solve_async()
    fa = {k: a.get(k) for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z", "x0")}
    sa = a.reduce_stats()
    a.close()
    f = make_batch(suite, batch=B)
    f.set_option("steps_per_launch", T)
    f.set_option("step_log", 1)
    f.set_x_ref(xref, broadcast=True)
    f.set_x0(x0, broadcast=True)
    its, u0s = [], []
    for _ in range(100 // T):
        f.solve_async()
        it, u0 = f.step_log(T)
        its.append(it)
        u0s.append(u0)
    its, u0s = np.concatenate(its), np.concatenate(u0s)
    assert np.array_equal(np.abs(its[:, 0]), gold["iters"]) and np.abs(its[:, 0]).sum() == 882
    assert np.all(its == its[:, :1])                                   # every instance identical
    assert np.array_equal(its[:, 0] < 0, gold["iters"] == 100)         # first five solves hit max_iter
    assert rel_err(u0s[:10, 0], gold["u0"][:10]) < RTOL and rel_err(u0s[:, 0], gold["u0"]) < 1e-6
    ff = {k: f.get(k) for k in fa}
    sf = f.reduce_stats()
    f.close()
    for k in fa:
        assert np.array_equal(ff[k], fa[k]), f"fused launch left a different {k}"
    assert sf[7] == sa[7] == 882.0 * B and sf[8] == sa[8] == 95.0 * B
    # the byte-saving launch forms: no x|u write-back ("store_primal" = 0), per-instance reference records instead of the
    # shared one ("share_ref" = 0) and the v|z store that first-check convergence skips -- everything a later solve or the
    # caller can see (slack, duals, v|z, plant state, statistics) must stay bit-identical, one launch per step
    for opts in (dict(store_primal=0), dict(share_ref=0), dict(store_primal=0, share_ref=0), dict(store_primal=2)):
        c = make_batch(suite, batch=B)
        c.set_option("advance_x0", 1)
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_x_ref(xref, broadcast=True)
        c.set_x0(x0, broadcast=True)
        for _ in range(100):
            c.solve_async()
        for k in fa:
            if k in ("x", "u") and opts.get("store_primal", 1) == 0:
                continue
            if k in ("x", "u") and opts.get("store_primal", 1) == 2:           # first knot only: x_0, x_1, u_0
                n_first = 2 if k == "x" else 1
                assert np.array_equal(c.get(k)[:, :, :n_first], fa[k][:, :, :n_first]), (opts, k)
                continue
            assert np.array_equal(c.get(k), fa[k]), (opts, k)
        sc_ = c.reduce_stats()
        assert np.array_equal(sc_, sa), opts
        c.close()


@pytest.mark.parametrize("T", [97, 1])
def test_fused_tracking_episode(T):
    """examples/quadrotor_tracking.cpp on device: a shared 301-point reference trajectory whose N-knot window advances
    one knot per MPC step, duals reset before every solve, plant stepped on device.  291 steps in 3 launches of 97
    (or 291 launches of 1) must reproduce the reference's per-step iteration sequence (725 total) and final state."""
    gold = np.load(os.path.join(GOLDEN, "tracking_episode.npz"))
    suite, _ = sc.load_suite(os.path.join(GOLDEN, "hover_warm.npz"))
    prob, extra = sc.load_problem("quadrotor_20hz")
    traj = np.array(extra["y_axis_line"])
    B = 37
    s = make_batch(suite, batch=B)
    s.set_reference_trajectory(traj)
    s.set_option("reset_duals", 1)
    s.set_option("advance_x0", 1)
    s.set_option("steps_per_launch", T)
    s.set_option("step_log", 1)
    s.set_x0(traj[0], broadcast=True)
    its = []
    for _ in range(291 // T):
        s.solve_async()
        if T > 1:
            its.append(s.step_log(T)[0])
        else:
            its.append(s.status()["iter"][None, :] * np.where(s.status()["solved"] == 1, 1, -1)[None, :])
    its = np.concatenate(its)
    assert np.all(its == its[:, :1])
    assert np.array_equal(np.abs(its[:, 0]), gold["iters"]) and int(np.abs(its[:, 0]).sum()) == 725
    x_final = s.get("x0")
    assert rel_err(x_final[0], gold["x_final"]) < 1e-6 and np.all(x_final == x_final[0])
    s.close()


def test_tracking_full_batch_properties():
    """BASELINE config 3 at full size (262 144 instances, per-instance random references): ALL 4096 unique
    instances are checked against the oracle (iteration counts, solved flags, every field), every replica of them
    must agree bit for bit, and the whole batch goes through properties (iteration counts in range, residuals
    below tolerance when solved, solution inside the box)."""
    B = 262144
    base = sc.tracking_random_suite(B=4096, seed=1234)
    suite = dict(problem=base["problem"], config=base["config"],
                 cases={k: np.concatenate([v] * (B // 4096), axis=0) for k, v in base["cases"].items()})
    out = run_cases_hip(suite)
    ref = sc.run_cases(OracleSolver, base)
    assert np.array_equal(out["iter"][:4096].astype(int), ref["iter"].astype(int))
    assert np.array_equal(out["sol_solved"][:4096].astype(int), ref["sol_solved"].astype(int))
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z"):
        for b in range(4096):                                                # per instance: a relative error each
            assert rel_err(out[k][b], ref[k][b]) < RTOL, (k, b)
    for k in ("iter", "sol_solved", "x", "u", "vnew", "znew", "g", "y", "v", "z"):
        v = out[k].reshape(B // 4096, 4096, -1)
        assert np.array_equal(v, np.broadcast_to(v[:1], v.shape)), k         # replicas agree bit for bit
    solved = out["sol_solved"] == 1
    assert solved.mean() > 0.99
    assert np.all(out["primal_residual_state"][solved] < 1e-3) and np.all(out["dual_residual_input"][solved] < 1e-3)
    assert np.all(np.abs(out["znew"]) <= 0.5 + 1e-15) and np.all(np.abs(out["vnew"]) <= 5 + 1e-15)


@pytest.mark.parametrize("dims,B", [((4, 2, 50), 131072), ((12, 8, 10), 32768), ((20, 8, 30), 32768)])
def test_sweep_cells_full_batch_on_the_dynamic_tile_form(dims, B):
    """BASELINE config 5 cells at benchmark batch sizes (the sweep cells were only parity-tested at B = 2-9): 512 unique
    instances of the cell's recipe (per-instance random x0 / Xref, max_iter 500, iteration counts from a handful to 500) are
    checked against the oracle field by field, their replicas must agree bit for bit -- on the DYNAMIC form of the tile kernel
    (persistent grid, slots draw instances from a device-wide counter: which wave solves which instance differs from launch to
    launch, the results must not) -- and the static-tile form of the same kernel must give the very same bits."""
    import tinympc_amd as tm
    U = 512
    base = sc.sweep_suite(*dims, B=U, max_iter=500)
    suite = dict(problem=base["problem"], config=base["config"],
                 cases={k: np.concatenate([v] * (B // U), axis=0) for k, v in base["cases"].items()})
    probe = make_batch(suite)
    assert probe.kernel_path() == "tile"
    probe.close()
    ref = sc.run_cases(OracleSolver, base)
    assert len(np.unique(ref["iter"])) > 5                                     # the cell does diverge
    outs = {}
    for dyn in (1, 0):
        out = run_cases_hip(suite, options={"tile_dyn": dyn})
        outs[dyn] = out
        assert np.array_equal(out["iter"][:U].astype(int), ref["iter"].astype(int))
        assert np.array_equal(out["sol_solved"][:U].astype(int), ref["sol_solved"].astype(int))
        for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z"):
            for b in range(U):
                assert rel_err(out[k][b], ref[k][b]) < RTOL, (k, b, dyn)
            v = out[k].reshape(B // U, U, -1)
            assert np.array_equal(v, np.broadcast_to(v[:1], v.shape)), (k, dyn)   # replicas agree bit for bit
    for k in ("iter", "sol_solved", "x", "u", "vnew", "znew", "g", "y", "v", "z", "primal_residual_state", "dual_residual_input"):
        assert np.array_equal(outs[0][k], outs[1][k]), k                        # dynamic form == static form, bit for bit
    if os.environ.get("TINYMPC_TEST_OPTS"):                                     # (a rerun of the suite under forced options: the default is not in play)
        return
    s = make_batch(suite)                                                       # and the default really takes the dynamic form at this size
    s.set_x0(suite["cases"]["x0"]); s.set("Xref", suite["cases"]["Xref"])
    s.solve()
    assert s.get_option("last_tile_dyn") == 1
    s.close()


def test_rocket_soc_full_batch_properties():
    """BASELINE config 4 at full per-job size: 65 536 rocket-landing instances with the second-order-cone
    thrust constraint on, perturbed initial states.  ALL 2048 unique instances are checked against the oracle,
    their replicas must agree bit for bit, and all of them go through properties of the cone projection: zcnew lies inside the cone ||u_xy|| <= mu * u_z
    (up to rounding), box slack inside the box, iteration counts in range."""
    B = 65536
    base = sc.rocket_random_suite(B=2048, seed=31337)
    suite = dict(problem=base["problem"], config=base["config"],
                 cases={k: np.concatenate([v] * (B // 2048), axis=0) for k, v in base["cases"].items()})
    out = run_cases_hip(suite)
    ref = sc.run_cases(OracleSolver, base)
    assert np.array_equal(out["iter"][:2048].astype(int), ref["iter"].astype(int))
    assert np.array_equal(out["sol_solved"][:2048].astype(int), ref["sol_solved"].astype(int))
    for k in ("x", "u", "vnew", "znew", "zcnew", "yc", "g", "y"):
        for b in range(2048):
            assert rel_err(out[k][b], ref[k][b]) < RTOL, (k, b)
    for k in ("iter", "x", "u", "znew", "zcnew", "yc", "g"):
        v = out[k].reshape(B // 2048, 2048, -1)
        assert np.array_equal(v, np.broadcast_to(v[:1], v.shape)), k
    zc = out["zcnew"]                                    # [B, 3, N-1]
    mu = float(np.float32(base["config"]["input_cone"][2][0]))
    nrm = np.sqrt(zc[:, 0] ** 2 + zc[:, 1] ** 2)
    assert np.all(nrm <= mu * zc[:, 2] * (1 + 1e-6) + 1e-9)
    assert np.all(out["znew"] <= 105 + 1e-12) and np.all(out["znew"] >= -10 - 1e-12)
    assert out["iter"].min() >= 1 and out["iter"].max() <= 100


@pytest.mark.parametrize("name", ["tracking_random", "rocket_random_isoc", "linear_random_all"])
def test_one_shot_equals_reset_then_solve(name):
    """one_shot = 1 / 2 (SURVEY 8(d) bytes_cold): the warm-start records are neither read nor (except the results)
    written; the results must be bit-identical to a solve from the reset state, whatever garbage the records hold."""
    suite, _ = sc.load_suite(os.path.join(GOLDEN, name + ".npz"))
    cases = suite["cases"]
    B = cases["x0"].shape[0]
    zero = {k: np.zeros_like(v) for k, v in cases.items() if k not in ("x0", "Xref", "Uref")}
    ref_suite = dict(suite, cases=dict(cases, **zero))
    ref = run_cases_hip(ref_suite)                                    # warm path from an all-zero state
    for mode in (1, 2):
        s = make_batch(suite)
        rng = np.random.default_rng(mode)
        for f in ("vnew", "znew", "g", "y", "v", "z", "x", "u"):
            s.set(f, rng.normal(0, 7.0, cases[f].shape))              # garbage that must not be read
        s.set_x0(cases["x0"]); s.set("Xref", cases["Xref"]); s.set("Uref", cases["Uref"])
        s.set_option("one_shot", mode)
        s.solve()
        st = s.status()
        assert np.array_equal(st["iter"], ref["iter"].astype(int)) and np.array_equal(st["solved"], ref["sol_solved"].astype(int))
        assert np.array_equal(s.get("x"), ref["x"]) and np.array_equal(s.get("u"), ref["u"])
        if mode == 1:
            assert np.array_equal(s.get("vnew"), ref["vnew"]) and np.array_equal(s.get("znew"), ref["znew"])
        assert s.kernel_path() == "regs"
        s.close()


def test_million_instance_batch():
    """BASELINE config 5's batch size (2^20 instances, 1.3 GB per record family) in one launch: index arithmetic past
    2^31 bytes, grid of 262 144 workgroups; 20 fused MPC steps must reproduce the reference's iteration sequence in
    every instance and leave every instance bit-identical to instance 0."""
    suite, _ = sc.load_suite(os.path.join(GOLDEN, "hover_warm.npz"))
    prob, extra = sc.load_problem("quadrotor_20hz")
    B = 1 << 20
    s = make_batch(suite, batch=B)
    s.set_x_ref(np.tile(np.array(extra["hover"]["xref"], dtype=float).reshape(-1, 1), (1, prob["N"])), broadcast=True)
    s.set_x0(np.array(extra["hover"]["x0"], dtype=float), broadcast=True)
    s.set_option("steps_per_launch", 20)
    s.solve_async()
    st = s.reduce_stats()
    assert st[7] == float(suite["episode"]["iters"][:20].sum()) * B and st[2] == B
    it = s.status()["iter"]
    assert np.all(it == suite["episode"]["iters"][19])
    for k in ("u", "vnew", "g", "x0"):
        a = s.get(k)
        assert np.all(a == a[:1]), k
    s.close()


def test_device_pointer_set_get_roundtrip():
    """TINY_DEVICE flags: a host that already owns HBM buffers (here torch tensors) hands them over / receives
    results without a PCIe round trip; must equal the host-pointer path.  Runs in a fresh interpreter that imports
    torch FIRST (torch bundles its own HIP runtime; it cannot initialise after libtinympc_amd's is already live)."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "device_ptr_check.py")
    p = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "device pointer path ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_tiny_solve_batch_over_reference_structs():
    """tiny_solve_batch(TinySolver**, n): n ordinary reference-layout solvers (plain-data mirrors), one launch;
    every workspace field the reference's solve() writes must match the oracle."""
    import ctypes as C
    import pod
    import tinympc_amd as tm
    L = tm.lib()
    prob, extra = sc.load_problem("quadrotor_20hz")
    suite = sc.tracking_random_suite(B=5, seed=123)
    ref = sc.run_cases(OracleSolver, suite)
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    L.tiny_setup.argtypes = [C.POINTER(C.POINTER(pod.TinySolver))] + [C.POINTER(pod.Mat)] * 5 + [C.c_double] + [C.c_int] * 4
    L.tiny_set_bound_constraints.argtypes = [C.POINTER(pod.TinySolver)] + [C.POINTER(pod.Mat)] * 4
    L.tiny_solve_batch.argtypes = [C.POINTER(C.POINTER(pod.TinySolver)), C.c_int]
    L.tiny_destroy.argtypes = [C.POINTER(pod.TinySolver)]
    keep, solvers = [], (C.POINTER(pod.TinySolver) * 5)()
    for b in range(5):
        ms = [pod.mat(prob["A"]), pod.mat(prob["B"]), pod.mat(prob["f"]), pod.mat(np.diag(prob["Q"])), pod.mat(np.diag(prob["R"]))]
        keep.append(ms)
        sp = C.POINTER(pod.TinySolver)()
        assert L.tiny_setup(C.byref(sp), *[C.byref(m[0]) for m in ms], prob["rho"], nx, nu, N, 0) == 0
        cfg = suite["config"]
        bs = [pod.mat(cfg[k]) for k in ("x_min", "x_max", "u_min", "u_max")]
        keep.append(bs)
        assert L.tiny_set_bound_constraints(sp, *[C.byref(m[0]) for m in bs]) == 0
        sp.contents.settings.contents.max_iter = cfg["max_iter"]
        w = sp.contents.work.contents
        pod.to_np(w.Xref)[...] = suite["cases"]["Xref"][b]
        pod.to_np(w.Uref)[...] = suite["cases"]["Uref"][b]
        pod.to_np(w.x)[:, 0] = suite["cases"]["x0"][b]
        solvers[b] = sp
    rc = L.tiny_solve_batch(solvers, 5)
    assert rc == int(np.any(ref["sol_solved"] == 0))
    for b in range(5):
        s = solvers[b].contents
        w = s.work.contents
        assert w.iter == int(ref["iter"][b]) and w.status == int(ref["status"][b]) and s.solution.contents.solved == int(ref["sol_solved"][b])
        for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z", "q", "r", "p", "d"):
            assert rel_err(pod.to_np(getattr(w, k)), ref[k][b]) < RTOL, (b, k)
        assert rel_err(pod.to_np(s.solution.contents.x), ref["sol_x"][b]) < RTOL
        assert abs(w.primal_residual_input - ref["primal_residual_input"][b]) < 1e-11
        L.tiny_destroy(solvers[b])


def test_api_misuse_is_reported_not_ignored():
    """Out-of-scope or malformed requests fail loudly with the documented codes (no silent fallback)."""
    import ctypes as C
    import tinympc_amd as tm
    prob, _ = sc.load_problem("quadrotor_20hz")
    with pytest.raises(tm.TinyMPCError):                       # nx + nu > 32
        tm.TinyBatchSolver(np.eye(30), np.ones((30, 8)), None, np.ones(30), np.ones(8), 1.0, 30, 8, 10, 4)
    s = tm.TinyBatchSolver.from_problem(prob, 4)
    with pytest.raises(tm.TinyMPCError, match="cone dimension"):
        s.set_cone_constraints([0], [4], [0.5], [], [], [])   # the reference's project_soc only handles 3 (admm.cpp:53)
    with pytest.raises(tm.TinyMPCError):
        s.get("q")                                            # q/r/p/d need the debug option (or a solve on the coverage kernel)
    # overlapping cones are served (sequential projections, admm.cpp:111-135) -- by the coverage kernel, fused steps included since
    # round 5 (tests/test_gpu_fused_variants.py), per-instance data since round 6 (tests/test_gpu_hetero.py); what it cannot take is adaptive rho
    s.set_cone_constraints([0, 2], [3, 3], [0.5, 0.5], [], [], [])
    s.update_settings(en_state_soc=1)
    assert s.kernel_path() == "cover"
    s.set_option("steps_per_launch", 3)
    s.solve()
    s.set_option("steps_per_launch", 1)
    s.set_sensitivity(*[np.zeros((a, b_)) for a, b_ in ((4, 12), (12, 12), (4, 4), (12, 12))])
    s.set_adaptive_rho(1, 1.0, 100.0, 1)
    with pytest.raises(tm.TinyMPCError, match="overlapping cones"):
        s.solve()
    s.set_adaptive_rho(0, 1.0, 100.0, 1)
    s.update_settings(en_state_soc=0)                          # the family is switched off: the overlap is irrelevant again
    assert s.kernel_path() == "regs"
    s.set_cone_constraints([], [], [], [], [], [])
    with pytest.raises(tm.TinyMPCError, match="out of range"):
        s.set_cone_constraints([], [], [], [2], [3], [0.5])   # nu = 4: rows 2..4 do not exist
    s.close()
