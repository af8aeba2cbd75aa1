"""Shapes outside kernel_dims.txt: the one-row register kernel is instantiated at run time with hipRTC (csrc/jit.hip)
instead of falling back to the coverage kernel.  Same parity bar, same features (cone, half-spaces, fused steps,
heterogeneous data)."""
import os
import sys
import time

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
sys.path.insert(0, HERE)

import scenarios as sc  # noqa: E402
import tinympc_amd as tm  # noqa: E402
from cpu_solvers import OracleSolver  # noqa: E402
from hip_runner import make_batch, run_cases_hip  # noqa: E402

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.parametrize("dims", [(5, 3, 7), (9, 2, 12), (3, 1, 4), (7, 7, 5), (10, 6, 20)])
def test_uninstantiated_shape_runs_the_register_kernel(dims):
    suite = sc.sweep_suite(*dims, B=9, max_iter=120)
    assert dims not in tm.supported_dims()
    s = make_batch(suite)
    assert s.kernel_path() == "jit"
    s.set_option("no_jit", 1)
    assert s.kernel_path() == "cover"
    s.close()
    out, ref = run_cases_hip(suite), sc.run_cases(OracleSolver, suite)
    assert np.array_equal(out["iter"].astype(int), ref["iter"].astype(int))
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z"):
        assert rel_err(out[k], ref[k]) < RTOL, k
    cov = run_cases_hip(suite, options={"no_jit": 1})
    for k in ("x", "u", "vnew", "g", "v"):
        assert rel_err(cov[k], ref[k]) < RTOL, k


def test_jit_shape_with_cone_halfspaces_and_fused_steps():
    nx, nu, N = 7, 4, 9
    prob = sc.sweep_suite(nx, nu, N, B=1)["problem"]
    rng = np.random.default_rng(5)
    cfg = sc.default_config(prob, max_iter=40, en_state_soc=1, en_input_soc=1, u_min=-0.5, u_max=0.5,
                            state_cone=([2], [3], [0.7]), input_cone=([1], [3], [0.5]), en_state_linear=1, en_tv_input_linear=1,
                            linear=(rng.standard_normal((2, nx)), rng.uniform(0.2, 1.0, 2), np.zeros((0, nu)), np.zeros(0)),
                            tv_linear=(np.zeros((0, nx)), np.zeros((0, N)), rng.standard_normal((N - 1, nu)), rng.uniform(0.1, 0.5, (1, N - 1))))
    cases = sc.zero_cases(prob, 6)
    for k, v in cases.items():
        cases[k] = rng.normal(0, 0.3, v.shape)
    suite = dict(problem=prob, config=cfg, cases=cases)
    s = make_batch(suite)
    assert s.kernel_path() == "jit"
    s.close()
    out, ref = run_cases_hip(suite), sc.run_cases(OracleSolver, suite)
    assert np.array_equal(out["iter"].astype(int), ref["iter"].astype(int))
    for k in ("x", "u", "vnew", "znew", "g", "y", "vcnew", "zcnew", "gc", "yc", "vlnew", "gl", "zlnew_tv", "yl_tv"):
        assert rel_err(out[k], ref[k]) < RTOL, k
    # fused closed loop == separate launches, on the run-time instantiated kernel
    res = []
    for T in (1, 5):
        f = make_batch(suite)
        f.set_x0(cases["x0"]); f.set("Xref", cases["Xref"]); f.set("Uref", cases["Uref"])
        f.set_option("advance_x0", 1); f.set_option("steps_per_launch", T)
        for _ in range(5 // T):
            f.solve_async()
        res.append({k: f.get(k) for k in ("x", "u", "vnew", "g", "zcnew", "yc", "vlnew", "x0")})
        f.close()
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k


def test_jit_is_much_faster_than_the_coverage_kernel():
    suite = sc.sweep_suite(5, 3, 7, B=1)
    prob = suite["problem"]
    B = 32768
    rng = np.random.default_rng(1)
    t = {}
    for name, opt in (("jit", 0), ("cover", 1)):
        s = tm.TinyBatchSolver.from_problem(prob, B)
        s.set_bound_constraints(np.full((5, 1), -1e17), np.full((5, 1), 1e17), np.full((3, 1), -0.5), np.full((3, 1), 0.5))
        s.update_settings(max_iter=50, check_termination=0)
        s.set_option("no_jit", opt)
        s.set_x0(rng.uniform(-1, 1, (B, 5)))
        s.solve()                                   # includes the one-off compilation
        s.set_option("timing", 1)
        s.solve_async()
        t[name] = float(s.timing_ms()[0])
        assert s.kernel_path() == name
        s.close()
    print(f"(5,3,7) x {B}, 50 iterations: jit {t['jit']:.3f} ms, coverage {t['cover']:.3f} ms")
    assert t["jit"] * 10 < t["cover"]


@pytest.mark.parametrize("dims,path", [((16, 8, 6), "tile-jit"), ((6, 2, 60), "tile-jit"), ((20, 6, 24), "tile-jit"), ((10, 2, 40), "tile-jit")])
def test_uninstantiated_wide_or_long_shape_runs_the_tile_kernel(dims, path):
    suite = sc.sweep_suite(*dims, B=7, max_iter=150)
    s = make_batch(suite)
    assert s.kernel_path() == path
    s.close()
    out, ref = run_cases_hip(suite), sc.run_cases(OracleSolver, suite)
    assert np.array_equal(out["iter"].astype(int), ref["iter"].astype(int))
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z"):
        assert rel_err(out[k], ref[k]) < RTOL, k


@pytest.mark.parametrize("dims", [(16, 8, 6), (6, 2, 60)])
def test_uninstantiated_tile_shape_takes_the_dynamic_slot_form(dims):
    """Round 3: a run-time instantiated tile shape gets the dynamic slot form as well (persistent grid, device-wide work counter) --
    forced here on a small ragged batch (tile_dyn = 1); results = the static form's, bit for bit, and the oracle's."""
    suite = sc.sweep_suite(*dims, B=37, max_iter=150)
    ref = sc.run_cases(OracleSolver, suite)
    static = run_cases_hip(suite, options={"tile_dyn": 0})
    s = make_batch(suite)
    s.set_option("tile_dyn", 1)
    s.set_x0(suite["cases"]["x0"]); s.set("Xref", suite["cases"]["Xref"])
    s.solve()
    assert s.kernel_path() == "tile-jit" and s.get_option("last_tile_dyn") == 1
    assert np.array_equal(s.status()["iter"], ref["iter"].astype(int))
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z"):
        assert np.array_equal(s.get(k), static[k]), k
        assert rel_err(static[k], ref[k]) < RTOL, k
    s.close()


def test_many_halfspaces_stay_register_resident():
    """4 half-spaces per knot and family are compiled in; more get the KMAX = 8 / 16 / 32 variant at run time; 33 go to the
    coverage kernel.  Same results either way."""
    prob, _ = sc.load_problem("quadrotor_20hz")
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    rng = np.random.default_rng(12)
    for ns, want in ((4, "regs"), (7, "regs"), (9, "regs"), (20, "regs"), (33, "cover")):
        cfg = sc.default_config(prob, max_iter=30, en_state_linear=1, en_input_linear=1, u_min=-0.5, u_max=0.5,
                                linear=(rng.standard_normal((ns, nx)), rng.uniform(0.3, 1.0, ns), rng.standard_normal((2, nu)), rng.uniform(0.2, 0.6, 2)))
        cases = sc.zero_cases(prob, 5)
        for k in ("x0", "Xref", "Uref"):
            cases[k] = rng.normal(0, 0.3, cases[k].shape)
        suite = dict(problem=prob, config=cfg, cases=cases)
        s = make_batch(suite)
        assert s.kernel_path() == want, (ns, s.kernel_path())
        s.close()
        out, ref = run_cases_hip(suite), sc.run_cases(OracleSolver, suite)
        assert np.array_equal(out["iter"].astype(int), ref["iter"].astype(int)), ns
        for k in ("x", "u", "vnew", "vlnew", "zlnew", "gl", "yl"):
            assert rel_err(out[k], ref[k]) < RTOL, (ns, k)


@pytest.mark.parametrize("dims,cone_rows,path", [((20, 4, 10), 3, "tile"), ((16, 8, 6), 15, "tile-jit"), ((8, 3, 50), 2, "tile")])
def test_cones_on_wide_and_long_shapes_run_the_tile_kernel(dims, cone_rows, path):
    """The tile kernel's cone variant (run-time instantiated only): a state cone starting at row `cone_rows` (for
    (16,8,6) it straddles the two 16-lane rows of the tile: rows 15, 16, 17) and an input cone."""
    nx, nu, N = dims
    prob = sc.sweep_suite(*dims, B=1)["problem"]
    rng = np.random.default_rng(21)
    cfg = sc.default_config(prob, max_iter=60, en_state_soc=1, en_input_soc=1, u_min=-0.5, u_max=0.5,
                            state_cone=([cone_rows if cone_rows + 3 <= nx else 0], [3], [0.6]), input_cone=([0], [3], [0.8]))
    cases = sc.zero_cases(prob, 6)
    for k, v in cases.items():
        cases[k] = rng.normal(0, 0.3, v.shape)
    suite = dict(problem=prob, config=cfg, cases=cases)
    s = make_batch(suite)
    got = s.kernel_path()
    s.close()
    if dims == (8, 3, 50):
        assert got in ("tile", "tile-jit")
    else:
        assert got == path
    out, ref = run_cases_hip(suite), sc.run_cases(OracleSolver, suite)
    assert np.array_equal(out["iter"].astype(int), ref["iter"].astype(int))
    for k in ("x", "u", "vnew", "znew", "g", "y", "vcnew", "zcnew", "gc", "yc"):
        assert rel_err(out[k], ref[k]) < RTOL, k


@pytest.mark.parametrize("dims", [(20, 4, 10), (16, 8, 6), (8, 2, 50)])
def test_halfspaces_and_cones_on_wide_and_long_shapes_run_the_tile_kernel(dims):
    """Static + time-varying half-spaces (and a cone) on tile shapes: the tile kernel's LIN x SOC variants, run-time
    instantiated; 6 static state half-spaces exercise the KMAX = 8 tables."""
    nx, nu, N = dims
    prob = sc.sweep_suite(*dims, B=1)["problem"]
    rng = np.random.default_rng(33)
    tv_ok = N <= 20                                              # the per-knot table must fit the wave's LDS
    ns = 4 if dims == (20, 4, 10) else 6                         # (20,4,10): KMAX = 8 per-knot tables of a 32-lane tile exceed it
    cfg = sc.default_config(prob, max_iter=40, en_state_soc=1, u_min=-0.5, u_max=0.5, state_cone=([1], [3], [0.6]), input_cone=([], [], []),
                            en_state_linear=1, en_input_linear=1, en_tv_state_linear=int(tv_ok), en_tv_input_linear=int(tv_ok),
                            linear=(rng.standard_normal((ns, nx)), rng.uniform(0.3, 1.0, ns), rng.standard_normal((2, nu)), rng.uniform(0.2, 0.6, 2)),
                            tv_linear=(rng.standard_normal((2 * N, nx)), rng.uniform(0.3, 1.0, (2, N)), rng.standard_normal((N - 1, nu)), rng.uniform(0.2, 0.6, (1, N - 1))))
    cases = sc.zero_cases(prob, 5)
    for k, v in cases.items():
        cases[k] = rng.normal(0, 0.3, v.shape)
    suite = dict(problem=prob, config=cfg, cases=cases)
    s = make_batch(suite)
    assert s.kernel_path() in ("tile", "tile-jit"), s.kernel_path()
    s.close()
    out, ref = run_cases_hip(suite), sc.run_cases(OracleSolver, suite)
    assert np.array_equal(out["iter"].astype(int), ref["iter"].astype(int))
    fields = ["x", "u", "vnew", "znew", "g", "y", "vcnew", "gc", "vlnew", "zlnew", "gl", "yl"] + (["vlnew_tv", "zlnew_tv", "gl_tv", "yl_tv"] if tv_ok else [])
    for k in fields:
        assert rel_err(out[k], ref[k]) < RTOL, k


DISK_CHILD = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "oracle")); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, sys.argv[1])
import scenarios as sc
import tinympc_amd as tm
from hip_runner import run_cases_hip
suite = sc.sweep_suite(6, 3, 8, B=9, max_iter=60)
t = time.time()
out = run_cases_hip(suite)
print(json.dumps({"s": time.time() - t, "x": out["x"].ravel().tolist(), "iter": out["iter"].astype(int).tolist(), "used": tm.jit_used()}))
'''


def test_disk_cache_serves_a_second_process(tmp_path):
    """TINYMPC_AMD_JIT_CACHE: the second process loads the code object the first one compiled -- same results, no compile."""
    import json
    import subprocess
    env = dict(os.environ, TINYMPC_AMD_JIT_CACHE=str(tmp_path))
    runs = []
    for _ in range(2):
        p = subprocess.run([sys.executable, "-c", DISK_CHILD, os.path.join(HERE, "..")], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        runs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert len(os.listdir(tmp_path)) == 1 and len(runs[0]["used"]) == 1 and runs[0]["used"] == runs[1]["used"]
    assert runs[0]["x"] == runs[1]["x"] and runs[0]["iter"] == runs[1]["iter"]
    n, hit = None, None
    os.environ["TINYMPC_AMD_JIT_CACHE"] = str(tmp_path)
    try:
        n, hit = tm.jit_compile(runs[0]["used"][0])
    finally:
        del os.environ["TINYMPC_AMD_JIT_CACHE"]
    assert hit and n > 1000


@pytest.mark.parametrize("dims", [(20, 4, 10), (8, 3, 50)])
def test_tile_cone_variant_fused_steps_equal_single_step_launches(dims):
    """closed loop on the tile kernel's cone variant (the slack in LDS planes, transposed cone step): T fused MPC steps must leave
    what T single-step launches leave, bit for bit -- the planes are re-initialised from x|u between the steps (admm.cpp:352-357)
    inside the launch exactly as a new launch initialises them from the records"""
    nx, nu, N = dims
    prob = sc.sweep_suite(*dims, B=1)["problem"]
    rng = np.random.default_rng(5)
    cfg = sc.default_config(prob, max_iter=30, en_state_soc=1, en_input_soc=1, u_min=-0.5, u_max=0.5,
                            state_cone=([2], [3], [0.6]), input_cone=([0], [3], [0.8]))
    cases = sc.zero_cases(prob, 7)
    cases["x0"] = rng.normal(0, 0.4, cases["x0"].shape)
    cases["Xref"] = rng.normal(0, 0.2, cases["Xref"].shape)
    suite = dict(problem=prob, config=cfg, cases=cases)
    T, outs = 5, []
    for fused in (0, 1):
        s = make_batch(suite)
        assert s.kernel_path() in ("tile", "tile-jit")
        s.set_option("advance_x0", 1)
        s.set_x0(cases["x0"]); s.set("Xref", cases["Xref"]); s.set("Uref", cases["Uref"])
        if fused:
            s.set_option("steps_per_launch", T)
            s.solve_async()
        else:
            for _ in range(T):
                s.solve_async()
        outs.append({k: s.get(k) for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z", "vcnew", "zcnew", "gc", "yc", "x0")})
        outs[-1]["acc"] = s.reduce_stats()[7:9]
        s.close()
    assert outs[0]["acc"][0] > T
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("dims", [(8, 4, 10), (12, 4, 30)])
def test_lean_shape_without_hiprtc_falls_back_to_the_coverage_kernel(dims):
    """a LEAN shape of kernel_dims.txt carries only its box kernel compiled in.  A cone / debug-output launch whose
    variant hipRTC cannot make (option no_jit stands for a box without hipRTC or a compile error) must be served by the coverage
    kernel -- as for a shape outside kernel_dims.txt -- not refused; the box launch of the same batch stays on the one-row kernel."""
    nx, nu, N = dims
    assert dims in tm.supported_dims()
    suite = sc.sweep_suite(nx, nu, N, B=6, max_iter=60)
    cfg = suite["config"]
    cfg.update(en_input_soc=1, input_cone=([0], [3], [0.6]))
    ref = sc.run_cases(OracleSolver, suite)
    out = run_cases_hip(suite, options={"no_jit": 1})
    assert np.array_equal(out["iter"].astype(int), ref["iter"].astype(int))
    for k in ("x", "u", "vnew", "znew", "g", "y", "v", "z", "vcnew", "gc"):
        assert rel_err(out[k], ref[k]) < RTOL, k
    s = make_batch(suite)

    def settings(**kw):
        c = dict(cfg, **kw)
        s.update_settings(c["abs_pri_tol"], c["abs_dua_tol"], c["max_iter"], c["check_termination"], c["en_state_bound"], c["en_input_bound"],
                          c["en_state_soc"], c["en_input_soc"])
    s.set_option("no_jit", 1)
    s.set_x0(suite["cases"]["x0"])
    s.set("Xref", suite["cases"]["Xref"])
    s.solve()
    assert s.kernel_path() == "cover"                 # the cone launch went to the coverage kernel
    settings(en_input_soc=0)                          # ... the box launch of the same handle does not
    s.reset()
    s.set_x0(suite["cases"]["x0"])
    s.set("Xref", suite["cases"]["Xref"])
    s.solve()
    assert s.kernel_path() == "regs"
    s.set_option("debug", 1)                          # debug outputs of a lean shape: not compiled in either
    s.solve()
    assert s.kernel_path() == "cover"
    # fused steps of the cone launch: the coverage kernel runs them too since round 5 (a loop of single-step launches)
    s.set_option("debug", 0)
    settings(en_input_soc=1)
    s.set_option("steps_per_launch", 3)
    s.solve()
    assert s.kernel_path() == "cover"
    s.set_option("no_jit", 0)                         # with hipRTC back the variant is instantiated and the launch runs
    s.solve()
    assert s.kernel_path() == "regs"
    s.close()
