"""tiny_codegen (reference src/tinympc/codegen.hpp:9-18) for the batched solver: the generated plain-C project.

CPU: the six entry points write <dir>/tinympc/tiny_data.h, src/tiny_data.c, src/tiny_main.c and a Makefile; the data file
holds the solver's cache / dynamics / constraints as round-trip literals (parsed back here bit for bit), everything compiles
as C99 and links against libtinympc_amd.so; without a GPU the generated program reports that and exits non-zero.
GPU: the generated program is built with its own Makefile and must solve exactly like a batch configured by hand."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pod  # noqa: E402
import scenarios as sc  # noqa: E402
import tinympc_amd as tm  # noqa: E402


def _solver(name, adaptive=False):
    L = tm.lib()
    prob, extra = sc.load_problem(name)
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    keep = [pod.mat(prob["A"]), pod.mat(prob["B"]), pod.mat(prob["f"]), pod.mat(np.diag(prob["Q"])), pod.mat(np.diag(prob["R"]))]
    sp = C.POINTER(pod.TinySolver)()
    L.tiny_setup.argtypes = [C.POINTER(C.POINTER(pod.TinySolver))] + [C.POINTER(pod.Mat)] * 5 + [C.c_double] + [C.c_int] * 4
    assert L.tiny_setup(C.byref(sp), *[C.byref(k[0]) for k in keep], prob["rho"], nx, nu, N, 0) == 0
    s = sp.contents
    L.tiny_set_bound_constraints.argtypes = [C.POINTER(pod.TinySolver)] + [C.POINTER(pod.Mat)] * 4
    if name == "quadrotor_20hz":
        h = extra["hover"]
        b = [pod.mat(np.full(shp, v)) for shp, v in (((nx, N), h["x_min"]), ((nx, N), h["x_max"]), ((nu, N - 1), h["u_min"]), ((nu, N - 1), h["u_max"]))]
        assert L.tiny_set_bound_constraints(sp, *[C.byref(m[0]) for m in b]) == 0
        s.settings.contents.max_iter = h["max_iter"]
        L.tiny_set_x_ref.argtypes = [C.POINTER(pod.TinySolver), C.POINTER(pod.Mat)]
        xr = pod.mat(np.tile(np.array(h["xref"], dtype=float).reshape(nx, 1), (1, N)))
        assert L.tiny_set_x_ref(sp, C.byref(xr[0])) == 0
    else:
        m = extra["mpc"]
        b = [pod.mat(np.tile(np.array(m["x_min"], dtype=float).reshape(nx, 1), (1, N))), pod.mat(np.tile(np.array(m["x_max"], dtype=float).reshape(nx, 1), (1, N))),
             pod.mat(np.full((nu, N - 1), m["u_min"])), pod.mat(np.full((nu, N - 1), m["u_max"]))]
        assert L.tiny_set_bound_constraints(sp, *[C.byref(k[0]) for k in b]) == 0
        L.tiny_set_cone_constraints.argtypes = [C.POINTER(pod.TinySolver), C.POINTER(pod.VecXi), C.POINTER(pod.VecXi), C.POINTER(pod.Vec)] * 2 \
            if False else [C.POINTER(pod.TinySolver), C.POINTER(pod.VecXi), C.POINTER(pod.VecXi), C.POINTER(pod.Vec), C.POINTER(pod.VecXi), C.POINTER(pod.VecXi), C.POINTER(pod.Vec)]
        cs = [pod.veci(m["state_cone"]["A"]), pod.veci(m["state_cone"]["q"]), pod.vec(m["state_cone"]["c"]),
              pod.veci(m["input_cone"]["A"]), pod.veci(m["input_cone"]["q"]), pod.vec(m["input_cone"]["c"])]
        assert L.tiny_set_cone_constraints(sp, *[C.byref(k[0]) for k in cs]) == 0
        s.settings.contents.en_input_soc = 1
        s.settings.contents.max_iter = m["max_iter"]
        s.settings.contents.abs_pri_tol = m["abs_pri_tol"]
    if adaptive:
        s.settings.contents.adaptive_rho = 1
        s.settings.contents.adaptive_rho_min = 0.8
        L.tiny_initialize_sensitivity_matrices.argtypes = [C.POINTER(pod.TinySolver)]
        L.tiny_initialize_sensitivity_matrices.restype = None
        L.tiny_initialize_sensitivity_matrices(sp)
    return L, sp, prob, extra


def _codegen(L, sp, out):
    L.tiny_codegen.argtypes = [C.POINTER(pod.TinySolver), C.c_char_p, C.c_int]
    assert L.tiny_codegen(sp, str(out).encode(), 0) == 0
    for f in ("tinympc/tiny_data.h", "src/tiny_data.c", "src/tiny_main.c", "Makefile"):
        assert os.path.exists(os.path.join(out, f)), f


def _array(text, name):
    m = re.search(r"static const double %s\[(\d+)\] = \{(.*?)\};" % name, text, re.S)
    assert m, name
    v = np.array([float(t) for t in m.group(2).replace("\n", " ").split(",")])
    assert len(v) == int(m.group(1))
    return v


@pytest.mark.parametrize("name,adaptive", [("quadrotor_20hz", False), ("quadrotor_20hz", True), ("rocket_landing_20hz", False)])
def test_generated_project_round_trips_the_family_and_builds(tmp_path, name, adaptive):
    L, sp, prob, extra = _solver(name, adaptive)
    out = tmp_path / "gen"
    _codegen(L, sp, out)
    src = open(out / "src" / "tiny_data.c").read()
    c, w = sp.contents.cache.contents, sp.contents.work.contents
    for arr, m in (("Kinf_data", c.Kinf), ("Pinf_data", c.Pinf), ("Quu_inv_data", c.Quu_inv), ("AmBKt_data", c.AmBKt), ("Adyn_data", w.Adyn),
                   ("Bdyn_data", w.Bdyn), ("x_min_data", w.x_min), ("u_max_data", w.u_max)):
        assert np.array_equal(_array(src, arr), pod.to_np(m).T.ravel()), arr       # %.17g literals: bit-exact, column-major
    assert np.array_equal(_array(src, "Q_data"), pod.to_np(w.Q)) and np.array_equal(_array(src, "APf_data"), pod.to_np(c.APf))
    if adaptive:
        assert np.array_equal(_array(src, "dKinf_drho_data"), pod.to_np(c.dKinf_drho).T.ravel())
    else:
        assert "dKinf_drho_data: empty" in src
    if name.startswith("rocket"):
        assert "static const int Acu_data[1] = {0};" in src and _array(src, "cu_data")[0] == 0.5
    # plain C99, links against the library; no GPU here -> the generated program says so
    lib = os.path.join(ROOT, "tinympc_amd")
    p = subprocess.run(["make", "-C", str(out), f"TINYMPC_AMD_LIB={lib}", f"TINYMPC_AMD_INC={os.path.join(ROOT, 'include')}", "CC=gcc -Wall -Werror"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    if tm.device_count() == 0:
        r = subprocess.run([str(out / "tiny_main"), "64"], capture_output=True, text=True)
        assert r.returncode == 1 and "needs an MI355X" in r.stderr and "rho:" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name,adaptive", [("quadrotor_20hz", False), ("quadrotor_20hz", True), ("rocket_landing_20hz", False)])
def test_generated_program_solves_like_a_hand_configured_batch(tmp_path, name, adaptive):
    from cpu_solvers import OracleSolver
    L, sp, prob, extra = _solver(name, adaptive)
    out = tmp_path / "gen"
    _codegen(L, sp, out)
    lib = os.path.join(ROOT, "tinympc_amd")
    assert subprocess.run(["make", "-C", str(out), f"TINYMPC_AMD_LIB={lib}", f"TINYMPC_AMD_INC={os.path.join(ROOT, 'include')}"],
                          capture_output=True, text=True).returncode == 0
    B = 1000
    r = subprocess.run([str(out / "tiny_main"), str(B)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"(\d+) ADMM iterations over (\d+) instances, (\d+) converged", r.stdout)
    assert m and int(m.group(2)) == B
    # the same problem through the oracle: zero initial state, the solver's references, cold start
    w, st = sp.contents.work.contents, sp.contents.settings.contents
    cfg = sc.default_config(prob, max_iter=st.max_iter, abs_pri_tol=st.abs_pri_tol, x_min=pod.to_np(w.x_min), x_max=pod.to_np(w.x_max),
                            u_min=pod.to_np(w.u_min), u_max=pod.to_np(w.u_max), en_input_soc=st.en_input_soc)
    if name.startswith("rocket"):
        mm = extra["mpc"]
        cfg.update(state_cone=(mm["state_cone"]["A"], mm["state_cone"]["q"], mm["state_cone"]["c"]),
                   input_cone=(mm["input_cone"]["A"], mm["input_cone"]["q"], mm["input_cone"]["c"]))
    if adaptive:
        cfg = sc.adaptive_cfg(cfg, rho_min=0.8)
    o = sc.make_solver(OracleSolver, prob, cfg)
    o["Xref"] = pod.to_np(w.Xref)
    ret = o.solve()
    assert int(m.group(1)) == int(o.get("sol_iter")) * B and int(m.group(3)) == (B if ret == 0 else 0)
    assert ("Hooray" in r.stdout) == (ret == 0)
    o.close()
