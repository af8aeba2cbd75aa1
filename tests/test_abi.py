"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/tinympc_amd.h
declares, fails loudly without a GPU, and its HOST logic (tiny_setup / cache precompute / setters over the
plain-data struct mirrors) reproduces the reference.  No compute calls that need a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import pod
import scenarios as sc
import tinympc_amd as tm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
HEADER = os.path.join(ROOT, "include", "tinympc_amd.h")
REF = "/root/reference"


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    phases = "update_linear_cost|backward_pass_grad|forward_pass|update_slack|update_dual|termination_condition|project_soc|project_hyperplane"
    return sorted(set(re.findall(r"\b((?:tiny_\w+)|(?:codegen_\w+)|solve|" + phases + r")\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 43, names
    L = tm.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/tinympc_amd.h but not exported by libtinympc_amd.so"
    assert set(tm.BATCH_SYMBOLS) | set(tm.GROUP_SYMBOLS) | set(tm.REFERENCE_SYMBOLS) == set(names)


def test_struct_sizes_match_reference_layout():
    """sizes/offsets measured on the reference's types.hpp (Eigen 3.4.90, x86-64)."""
    assert C.sizeof(pod.Mat) == 24 and C.sizeof(pod.Vec) == 16
    assert C.sizeof(pod.TinySolution) == 56 and C.sizeof(pod.TinyCache) == 280
    assert C.sizeof(pod.TinySettings) == 88 and C.sizeof(pod.TinyWorkspace) == 1328 and C.sizeof(pod.TinySolver) == 32
    W = pod.TinyWorkspace
    for name, off in (("x", 16), ("x_min", 304), ("numStateCones", 400), ("cx", 408), ("vc", 504), ("Alin_x", 656),
                      ("vl", 736), ("numtvStateLinear", 880), ("vl_tv", 984), ("Q", 1128), ("Adyn", 1160),
                      ("Xref", 1224), ("Qu", 1272), ("primal_residual_state", 1288), ("status", 1320), ("iter", 1324)):
        assert getattr(W, name).offset == off, name


@pytest.mark.skipif(not os.path.isdir(REF + "/src/tinympc"), reason="needs the reference headers")
def test_layout_against_the_real_reference_header(tmp_path):
    """offsetof/sizeof from the reference's OWN types.hpp vs the plain-data mirrors of our header."""
    fields = ["x", "u", "vnew", "x_min", "numStateCones", "cx", "Acx", "vc", "numStateLinear", "Alin_x", "blin_x", "vl",
              "numtvStateLinear", "tv_Alin_x", "vl_tv", "Q", "R", "Adyn", "fdyn", "Xref", "Uref", "Qu",
              "primal_residual_state", "status", "iter"]
    body = "".join(f'printf("{f} %zu\\n", offsetof(TinyWorkspace, {f}));' for f in fields)
    prog = ("#include <cstdio>\n#include <cstddef>\n#include \"@INC@\"\nint main(){"
            "printf(\"sizes %zu %zu %zu %zu %zu\\n\", sizeof(TinySolution), sizeof(TinyCache), sizeof(TinySettings),"
            " sizeof(TinyWorkspace), sizeof(TinySolver));" + body + "return 0;}")
    outs = []
    for tag, inc, flags in (("ref", "tinympc/types.hpp", ["-I" + REF + "/src", "-I" + REF + "/include/Eigen", "-I" + REF + "/include"]),
                            ("ours", HEADER, [])):
        src = tmp_path / f"{tag}.cpp"
        src.write_text(prog.replace("@INC@", inc))
        exe = tmp_path / tag
        subprocess.check_call(["g++", "-std=c++17", "-w", "-Wno-invalid-offsetof", *flags, str(src), "-o", str(exe)])
        outs.append(subprocess.check_output([str(exe)]).decode())
    assert outs[0] == outs[1], f"layout drift:\n{outs[0]}\nvs\n{outs[1]}"


@pytest.mark.skipif(not os.path.isdir(REF + "/src/tinympc"), reason="needs the reference headers")
def test_rho_adapter_layout_against_the_real_reference_header(tmp_path):
    """csrc/rho_api.hip mirrors RhoAdapter / RhoBenchmarkResult (rho_benchmark.hpp:5-40) as plain data and static_asserts their
    sizes (304 / 56): the real header must give those sizes and the offsets the mirror implies."""
    prog = ('#include <cstdio>\n#include <cstddef>\n#include "tinympc/rho_benchmark.hpp"\nint main(){'
            'printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(RhoAdapter), sizeof(RhoBenchmarkResult), offsetof(RhoAdapter, clip),'
            ' offsetof(RhoAdapter, matrices_initialized), offsetof(RhoAdapter, A_matrix), offsetof(RhoAdapter, q_vector), offsetof(RhoAdapter, ATy_vector),'
            ' offsetof(RhoAdapter, format_nx), offsetof(RhoBenchmarkResult, initial_rho), offsetof(RhoBenchmarkResult, dual_norm)); return 0;}')
    src = tmp_path / "rho_layout.cpp"
    src.write_text(prog)
    subprocess.check_call(["g++", "-std=c++17", "-w", "-Wno-invalid-offsetof", "-I" + REF + "/src", "-I" + REF + "/include/Eigen", "-I" + REF + "/include",
                           str(src), "-o", str(tmp_path / "rho_layout")])
    got = [int(v) for v in subprocess.check_output([str(tmp_path / "rho_layout")]).split()]
    #        sizes      clip  init  A    q              ATy             nx    initial  dual_norm
    assert got == [304, 56, 16, 17, 24, 24 + 5 * 24, 24 + 10 * 24, 288, 8, 48], got
    src_text = open(os.path.join(ROOT, "tinympc_amd", "csrc", "rho_api.hip")).read()
    assert "sizeof(TinyRhoAdapterPOD) == 304 && sizeof(TinyRhoBenchmarkResultPOD) == 56" in src_text


@pytest.mark.skipif(not os.path.isdir(REF + "/src/tinympc"), reason="needs the reference headers")
def test_every_function_the_reference_headers_declare_is_exported_or_accounted_for():
    """tiny_api.hpp, admm.hpp, codegen.hpp, rho_benchmark.hpp of the reference against `nm -D libtinympc_amd.so`: the only names
    missing are the C++-mangled internals of rho_benchmark.cpp and the five functions upstream declares but never defines
    (INTEGRATION.md)."""
    declared = set()
    for h in ("tiny_api.hpp", "admm.hpp", "codegen.hpp", "rho_benchmark.hpp"):
        text = re.sub(r"//.*", "", open(os.path.join(REF, "src", "tinympc", h)).read())
        declared |= set(re.findall(r"^\s*(?:int|void|bool|tinytype)\s+\*?(\w+)\s*\(", text, flags=re.M))
    assert len(declared) >= 35, sorted(declared)
    nm = subprocess.run(["nm", "-D", "--defined-only", tm.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.strip()}
    rho_helpers = {"initialize_format_matrices": "_Z26initialize_format_matricesP10RhoAdapteriii", "predict_rho": "_Z11predict_rhoP10RhoAdapterddddd",
                   "compute_residuals": "_Z17compute_residualsP10RhoAdapterPdS1_S1_S1_", "update_matrices_with_derivatives": "_Z32update_matrices_with_derivativesP9TinyCached",
                   "format_matrices": "_Z15format_matricesP10RhoAdapterRKN5Eigen6MatrixIdLin1ELin1ELi0ELin1ELin1EEES5_S5_S5_S5_S5_P9TinyCacheP13TinyWorkspacei",
                   "benchmark_rho_adaptation": "_Z24benchmark_rho_adaptationP10RhoAdapterRKN5Eigen6MatrixIdLin1ELin1ELi0ELin1ELin1EEES5_S5_S5_S5_S5_P9TinyCacheP13TinyWorkspaceiP18RhoBenchmarkResult"}
    for name, mangled in rho_helpers.items():                 # C++ linkage upstream: exported under their Itanium names (csrc/rho_api.hip)
        assert mangled in exported, name
    assert "_Z6microsv" in exported
    rho_internals = set(rho_helpers)
    never_defined = {"update_primal", "compute_sensitivity_matrices", "tiny_update_matrices_with_derivatives",
                     "tiny_setup_state_soc_constraints", "tiny_setup_input_soc_constraints"}
    sources = "".join(open(os.path.join(REF, "src", "tinympc", f)).read() for f in ("admm.cpp", "tiny_api.cpp", "codegen.cpp", "rho_benchmark.cpp"))
    for f in never_defined:                                   # (declared, called by nobody, defined nowhere)
        assert not re.search(r"^[\w:<>\*& ]+\b%s\s*\([^;]*\)\s*\{" % f, sources, flags=re.M), f
    assert declared - exported == rho_internals | never_defined, sorted(declared - exported - rho_internals - never_defined)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libtinympc_ref.so")), reason="needs oracle/_ref (built where /root/reference exists)")
def test_every_symbol_the_reference_library_defines_is_exported():
    """`nm -D` of the real reference compiled by oracle/Makefile vs libtinympc_amd.so: every function the reference DEFINES
    (C names and C++-mangled ones; Eigen / libstdc++ instantiations and the shim's own ref_* hooks aside) is a symbol here."""
    def syms(path, only_text):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and (not only_text or ln.split()[1] == "T")}
    ref = {s for s in syms(os.path.join(ROOT, "oracle", "_ref", "libtinympc_ref.so"), True)
           if not s.startswith(("ref_", "_ZN5Eigen", "_ZNK5Eigen", "_ZSt", "_ZNS", "_ZNK", "_ZN9__gnu", "_init", "_fini"))}
    assert len(ref) >= 25, sorted(ref)
    assert not (ref - syms(tm.LIB_PATH, False)), sorted(ref - syms(tm.LIB_PATH, False))


def test_no_gpu_fails_loudly():
    if tm.device_count() > 0:
        pytest.skip("a GPU is present")
    prob, _ = tm.load_problem("cartpole")
    with pytest.raises(tm.TinyMPCError, match="no CPU fallback"):
        tm.TinyBatchSolver.from_problem(prob, 8)


def _setup(name, N=None, verbose=0):
    L = tm.lib()
    prob, extra = sc.load_problem(name)
    nx, nu, N = prob["nx"], prob["nu"], N or prob["N"]
    keep = []
    A, k = pod.mat(prob["A"]); keep.append(k)
    B, k = pod.mat(prob["B"]); keep.append(k)
    f, k = pod.mat(prob["f"]); keep.append(k)
    Q, k = pod.mat(np.diag(prob["Q"])); keep.append(k)
    R, k = pod.mat(np.diag(prob["R"])); keep.append(k)
    sp = C.POINTER(pod.TinySolver)()
    L.tiny_setup.argtypes = [C.POINTER(C.POINTER(pod.TinySolver))] + [C.POINTER(pod.Mat)] * 5 + [C.c_double] + [C.c_int] * 4
    rc = L.tiny_setup(C.byref(sp), C.byref(A), C.byref(B), C.byref(f), C.byref(Q), C.byref(R), prob["rho"], nx, nu, N, verbose)
    return L, sp, prob, extra, rc


@pytest.mark.parametrize("name", ["codegen_random", "cartpole", "quadrotor_20hz", "rocket_landing_20hz"])
def test_tiny_setup_host_logic_matches_reference_cache(name):
    """tiny_setup + tiny_precompute_and_set_cache (host, no GPU): cache vs the reference's (golden KAT)."""
    L, sp, prob, _, rc = _setup(name)
    assert rc == 0
    kat = np.load(os.path.join(GOLDEN, "cache_kat.npz"))
    s = sp.contents
    for k in ("Kinf", "Pinf", "Quu_inv", "AmBKt", "APf", "BPf"):
        got = pod.to_np(getattr(s.cache.contents, k)).reshape(kat[f"{name}.{k}"].shape)
        err = np.max(np.abs(got - kat[f"{name}.{k}"])) / max(np.max(np.abs(kat[f"{name}.{k}"])), 1e-300)
        assert err < 1e-12, (name, k, err)
    w = s.work.contents
    assert np.allclose(pod.to_np(w.Q), prob["Q"] + prob["rho"]) and np.allclose(pod.to_np(w.R), prob["R"] + prob["rho"])
    assert np.array_equal(pod.to_np(s.cache.contents.C1), pod.to_np(s.cache.contents.Quu_inv))     # tiny_api.cpp:375
    st = s.settings.contents
    assert (st.abs_pri_tol, st.max_iter, st.check_termination, st.en_state_bound, st.en_state_soc, st.adaptive_rho) == \
        (1e-3, 1000, 1, 1, 0, 0)
    assert (w.x.rows, w.x.cols, w.u.rows, w.u.cols, w.Xref.cols) == (prob["nx"], prob["N"], prob["nu"], prob["N"] - 1, prob["N"])
    # setters
    L.tiny_set_x0.argtypes = [C.POINTER(pod.TinySolver), C.POINTER(pod.Vec)]
    x0, k0 = pod.vec(np.arange(1, prob["nx"] + 1))
    assert L.tiny_set_x0(sp, C.byref(x0)) == 0
    assert np.array_equal(pod.to_np(w.x)[:, 0], np.arange(1, prob["nx"] + 1)) and np.all(pod.to_np(w.x)[:, 1:] == 0)
    L.tiny_set_bound_constraints.argtypes = [C.POINTER(pod.TinySolver)] + [C.POINTER(pod.Mat)] * 4
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    ms = [pod.mat(np.full(shp, v)) for shp, v in (((nx, N), -1.0), ((nx, N), 2.0), ((nu, N - 1), -3.0), ((nu, N - 1), 4.0))]
    assert L.tiny_set_bound_constraints(sp, *[C.byref(m[0]) for m in ms]) == 0
    assert np.all(pod.to_np(w.u_max) == 4.0) and w.x_min.rows == nx
    L.tiny_update_settings.argtypes = [C.POINTER(pod.TinySettings), C.c_double, C.c_double] + [C.c_int] * 10
    assert L.tiny_update_settings(s.settings, 2e-3, 1e-4, 77, 3, 0, 1, 0, 1, 0, 0, 0, 0) == 0
    assert (st.abs_pri_tol, st.abs_dua_tol, st.max_iter, st.check_termination, st.en_state_bound, st.en_input_soc) == \
        (2e-3, 1e-4, 77, 3, 0, 1)
    L.tiny_destroy.argtypes = [C.POINTER(pod.TinySolver)]
    assert L.tiny_destroy(sp) == 0


def test_tiny_setup_rejects_bad_dimensions(capfd):
    L = tm.lib()
    A, k1 = pod.mat(np.eye(3)); B, k2 = pod.mat(np.ones((4, 1))); f, k3 = pod.mat(np.zeros(4))
    Q, k4 = pod.mat(np.eye(4)); R, k5 = pod.mat(np.eye(1))
    sp = C.POINTER(pod.TinySolver)()
    L.tiny_setup.argtypes = [C.POINTER(C.POINTER(pod.TinySolver))] + [C.POINTER(pod.Mat)] * 5 + [C.c_double] + [C.c_int] * 4
    assert L.tiny_setup(C.byref(sp), C.byref(A), C.byref(B), C.byref(f), C.byref(Q), C.byref(R), 1.0, 4, 1, 10, 0) == 1
    assert "State transition matrix (A) has 3 rows. Expected 4." in capfd.readouterr().out     # tiny_api.cpp:13-19


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "tests", "dropin", "_build", "cartpole_example")),
                    reason="drop-in binaries not built (tests/dropin/build.sh needs /root/reference)")
@pytest.mark.parametrize("ex", ["cartpole_example", "quadrotor_hovering", "rocket_landing_mpc"])
def test_dropin_example_setup_output_matches_reference(ex):
    """The reference's own example main(), compiled against the reference's headers and linked against
    libtinympc_amd.so: everything it prints during tiny_setup (verbose cache dump, Eigen IOFormat) must equal
    the reference's stdout.  (The solves need a GPU: tests/test_gpu_dropin.py.)"""
    if tm.device_count() > 0:
        pytest.skip("full run covered by the GPU test")
    exe = os.path.join(ROOT, "tests", "dropin", "_build", ex)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    gold = open(os.path.join(GOLDEN, f"stdout_{ex}.txt")).read()
    marker = "Precomputation finished!\n\n"
    assert marker in p.stdout and marker in gold
    assert p.stdout.split(marker)[0] == gold.split(marker)[0]
    assert "no CPU path" in p.stderr or "has no CPU" in p.stderr       # the solves refuse to run without a GPU


def test_headline_kernel_register_budget():
    """The quadrotor instantiation of the one-row kernel must stay at two waves per SIMD without scratch (a feature
    added to the shared template once pushed it to 256 VGPRs + 236 B of scratch unnoticed)."""
    src = os.path.join(ROOT, "tinympc_amd", "csrc", "_gen", "u_12_4_10.hip")
    if not os.path.exists(src):
        pytest.skip("library not built through the Makefile")
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    blocks = re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)",
                        p.stderr, re.S)
    # <12,4,10, box only, dpp_mode 2, plain>: the per-knot-box form and the knot-invariant-box form (UB) the bench runs
    head = [b for b in blocks if "ILi12ELi4ELi10ELb0ELb0ELi2ELi0ELb0ELi4ELb0ELb0ELb0ELb0EE" in b[0] or "ILi12ELi4ELi10ELb0ELb0ELi2ELi0ELb0ELi4ELb0ELb1ELb0ELb0EE" in b[0]]
    assert len(head) == 2, [b[0] for b in blocks][:4]
    for _, vgpr, agpr, scratch, occ in head:
        assert int(scratch) == 0 and int(agpr) == 0 and int(vgpr) <= 256 and int(occ) == 2, head
    # round 6: their PREFETCH forms (template argument PF) -- two waves per SIMD; what little scratch the UB form has (two loop-invariant
    # addresses, 16-20 B) sits in front of the straight v|z reads of a tile's top, nowhere near the iteration loop
    pf = [b for b in blocks if "ILi12ELi4ELi10ELb0ELb0ELi2ELi0ELb0ELi4ELb0ELb0ELb0ELb1EE" in b[0] or "ILi12ELi4ELi10ELb0ELb0ELi2ELi0ELb0ELi4ELb0ELb1ELb0ELb1EE" in b[0]]
    assert len(pf) == 2, [b[0] for b in blocks][:4]
    for _, vgpr, agpr, scratch, occ in pf:
        assert int(scratch) <= 32 and int(agpr) == 0 and int(vgpr) <= 256 and int(occ) == 2, pf


def test_cone_kernel_register_budget():
    """the rocket instantiation of the cone variant -- its slack lives in LDS planes since round 4 -- runs two
    waves per SIMD WITHOUT scratch, in the per-knot-box form and in the knot-invariant one (UB) config 4 launches."""
    src = os.path.join(ROOT, "tinympc_amd", "csrc", "_gen", "u_6_3_10.hip")
    if not os.path.exists(src):
        pytest.skip("library not built through the Makefile")
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    blocks = re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)",
                        p.stderr, re.S)
    cone = [b for b in blocks if "ILi6ELi3ELi10ELb1ELb0ELi2ELi0ELb0ELi4ELb0ELb0ELb0ELb0EE" in b[0] or "ILi6ELi3ELi10ELb1ELb0ELi2ELi0ELb0ELi4ELb0ELb1ELb0ELb0EE" in b[0]]
    assert len(cone) == 2, [b[0] for b in blocks][:6]
    for _, vgpr, agpr, scratch, occ in cone:
        assert int(scratch) == 0 and int(agpr) == 0 and int(vgpr) <= 256 and int(occ) == 2, cone


def test_header_is_plain_c_and_links(tmp_path):
    """include/tinympc_amd.h is the C ABI: it must compile as C99 (no C++-isms) and a C program must link against the
    library and get the documented error codes from calls that need no GPU."""
    src = tmp_path / "c_abi.c"
    src.write_text('''#include <stdio.h>
#include "tinympc_amd.h"
int main(void) {
    TinySettings st;
    int dims[64 * 3];
    if (tiny_batch_solve(0) != TINY_ERR_NULL) return 1;
    if (tiny_set_default_settings(&st) != 0 || st.max_iter != 1000 || st.check_termination != 1) return 2;   /* tiny_api.cpp:413-441 */
    if (tiny_batch_supported_dims(dims, 64) < 19) return 3;
    printf("ok %d\\n", (int)sizeof(TinyWorkspace));
    return 0;
}
''')
    exe = tmp_path / "c_abi"
    libdir = os.path.dirname(tm.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.dirname(HEADER), str(src),
                           "-L", libdir, "-ltinympc_amd", f"-Wl,-rpath,{libdir}", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok 1328", (out.returncode, out.stdout, out.stderr)


def test_plan_struct_is_a_pointer_free_pod_of_the_size_the_python_mirror_assumes(tmp_path):
    """TinyBatchPlan (tiny_batch_get_plan / tiny_batch_set_plan) crosses processes as raw bytes: its size and the offsets the ctypes
    mirror unpacks must be what a C99 compiler lays out"""
    import tinympc_amd as tm
    src = tmp_path / "plan.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tinympc_amd.h"\nint main(void){printf("%zu %zu %zu %zu\\n", sizeof(TinyBatchPlan), '
                   'offsetof(TinyBatchPlan, auto_verdict), offsetof(TinyBatchPlan, auto_plain_rate), offsetof(TinyBatchPlan, hist));return 0;}\n')
    exe = tmp_path / "plan"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, off_verdict, off_rates, off_hist = map(int, subprocess.check_output([str(exe)]).split())
    assert size == tm.PLAN_BYTES and off_verdict == 40 and off_rates == 80 and off_hist == 120 and size == off_hist + 4096
