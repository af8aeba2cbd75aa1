"""GPU test of the drop-in boundary: the reference's own example programs (compiled against the reference's
headers, linked against libtinympc_amd.so by tests/dropin/build.sh) must print exactly what they print when
linked against the reference itself -- iteration counts per solve, 'Solver converged in N iterations',
6-digit tracking errors, totals (BASELINE config 1 = cartpole_example)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "dropin", "_build")


@pytest.mark.parametrize("ex", ["cartpole_example", "quadrotor_hovering", "quadrotor_tracking", "rocket_landing_mpc",
                                "quadrotor_linear_constraints", "quadrotor_tv_linear_constraints"])
def test_reference_example_stdout_identical(ex):
    exe = os.path.join(BUILD, ex)
    if not os.path.exists(exe):
        pytest.fail("tests/dropin/_build/%s is missing: the drop-in binaries are built by __graft_entry__.build() in the container that has "
                    "/root/reference and travel to the GPU box with the snapshot -- a box without them must not pass the (b) row by skipping it" % os.path.basename(exe))
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    gold = open(os.path.join(ROOT, "tests", "golden", f"stdout_{ex}.txt")).read()
    got = p.stdout
    if got != gold:
        gl, ol = gold.splitlines(), got.splitlines()
        diff = [(i, a, b) for i, (a, b) in enumerate(zip(gl, ol)) if a != b][:5]
        pytest.fail(f"{ex}: stdout differs from the reference's ({len(gl)} vs {len(ol)} lines); first diffs: {diff}")


def test_phase_functions_called_with_real_eigen_types():
    """tests/dropin/phase_driver.cpp (our caller, the reference's headers): update_linear_cost ... termination_condition,
    project_soc and project_hyperplane cross the library boundary with real Eigen objects by value / by reference and an
    Eigen return value.  Every printed number must equal what the same program prints when linked against the reference
    (1e-9 of the line's largest magnitude: FMA contraction differs, nothing else may)."""
    exe = os.path.join(BUILD, "phase_driver")
    if not os.path.exists(exe):
        pytest.fail("tests/dropin/_build/%s is missing: the drop-in binaries are built by __graft_entry__.build() in the container that has "
                    "/root/reference and travel to the GPU box with the snapshot -- a box without them must not pass the (b) row by skipping it" % os.path.basename(exe))
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    gold = open(os.path.join(ROOT, "tests", "golden", "stdout_phase_driver.txt")).read().splitlines()
    got = p.stdout.splitlines()
    assert len(got) == len(gold) == 78

    def split(line):
        words = line.replace("(input kept:", "").replace(")", "").split()
        nums = [i for i, w in enumerate(words) if w[0] in "-0123456789" and any(c.isdigit() for c in w) and i >= 1]
        return [w for i, w in enumerate(words) if i not in nums], [float(words[i]) for i in nums]

    for a, b in zip(got, gold):
        (ta, na), (tb, nb) = split(a), split(b)
        assert ta == tb and len(na) == len(nb), (a[:80], b[:80])
        if tb[0] == "project_hyperplane":
            na, nb = na[:-1], nb[:-1]                     # the trailing residual is rounding noise in both
        scale = max([abs(v) for v in nb] + [1e-300])
        assert max([abs(x - y) for x, y in zip(na, nb)] + [0.0]) <= 1e-9 * scale, (a[:100], b[:100])


def test_plain_c_example_of_the_batched_abi():
    """examples/batched_double_integrator.c (C99, built by __graft_entry__.build()): 4 096 double integrators, 60 fused
    closed-loop MPC steps; every instance ends at the origin."""
    exe = os.path.join(ROOT, "examples", "_build", "batched_double_integrator")
    if not os.path.exists(exe):
        pytest.fail("examples/_build/batched_double_integrator is missing: __graft_entry__.build() produces it")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, (p.stdout, p.stderr[-500:])
    assert "4096 instances x 60 MPC steps" in p.stdout and "kernel path 0" in p.stdout


def test_plain_c_example_of_per_instance_data_windows_and_the_plan():
    """examples/hetero_tracking.c (C99, built by __graft_entry__.build()): 2 048 different (18,6,10) families -- a wide shape outside tile_dims.txt:
    the tile kernel's per-instance form, instantiated at run time -- tracking a moving reference window for 40 fused MPC steps with the duals reset before
    every solve; then the launch plan exported and imported into a second handle.  Everything through the C ABI from plain C."""
    exe = os.path.join(ROOT, "examples", "_build", "hetero_tracking")
    if not os.path.exists(exe):
        pytest.fail("examples/_build/hetero_tracking is missing: __graft_entry__.build() produces it")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout, p.stderr[-800:])
    assert "kernel path 4" in p.stdout and "imported into a second handle" in p.stdout      # 4: the tile kernel, shape instantiated at run time


def test_adaptive_rho_through_the_reference_structs():
    """tests/dropin/adaptive_driver.cpp (our caller, the reference's headers): settings->adaptive_rho = 1 and the
    reference's own tiny_initialize_sensitivity_matrices, 60 closed-loop hover steps through tiny_solve(TinySolver*).  Per
    step the iteration count and solved flag must be identical to the real reference's, cache->rho and the moved
    Kinf / Pinf / C2 entries and the applied control within 1e-7 of the line's largest magnitude (they are printed after 6+
    Taylor steps of a closed loop; the batched tests hold single solves to 1e-9)."""
    exe = os.path.join(BUILD, "adaptive_driver")
    if not os.path.exists(exe):
        pytest.fail("tests/dropin/_build/%s is missing: the drop-in binaries are built by __graft_entry__.build() in the container that has "
                    "/root/reference and travel to the GPU box with the snapshot -- a box without them must not pass the (b) row by skipping it" % os.path.basename(exe))
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    gold = open(os.path.join(ROOT, "tests", "golden", "stdout_adaptive_driver.txt")).read().splitlines()
    got = p.stdout.splitlines()
    assert len(got) == len(gold) == 61 and got[-1] == gold[-1] and gold[-1] == "total iterations 735"
    for a, b in zip(got[:-1], gold[:-1]):
        wa, wb = a.split(), b.split()
        assert wa[:6] == wb[:6], (a[:60], b[:60])                  # step k iter n solved s
        na = [float(w) for w in wa[6:] if w[0] in "-0123456789"]
        nb = [float(w) for w in wb[6:] if w[0] in "-0123456789"]
        assert len(na) == len(nb) == 9
        assert abs(na[0] - nb[0]) <= 1e-9 * abs(nb[0]), (a[:80], b[:80])          # rho
        for x, y in zip(na[1:5], nb[1:5]):                                          # cache entries: relative
            assert abs(x - y) <= 1e-9 * max(abs(y), 1e-300), (a[:120], b[:120])
        scale = max(abs(v) for v in nb[5:])
        assert max(abs(x - y) for x, y in zip(na[5:], nb[5:])) <= 1e-7 * scale, (a[-80:], b[-80:])


@pytest.mark.parametrize("ex,dims,gen_dir", [("codegen_random", (2, 2, 3), "tinympc_generated_code_random_example"),
                                             ("codegen_cartpole", (4, 1, 10), "tinympc_generated_code_cartpole_example")])
def test_reference_codegen_examples_generate_a_project_that_solves(tmp_path, ex, dims, gen_dir):
    """The reference's two code-generation examples, unmodified (real Eigen types -> tiny_setup -> tiny_codegen): the project
    they leave behind is the plain-C one of csrc/codegen.hip -- it must hold the solver the example configured (dimensions,
    rho, tolerances, cache computed by tiny_setup) and, built with its own Makefile, solve a batch on this GPU."""
    import re
    exe = os.path.join(BUILD, ex)
    if not os.path.exists(exe):
        pytest.fail("tests/dropin/_build/%s is missing: the drop-in binaries are built by __graft_entry__.build() in the container that has "
                    "/root/reference and travel to the GPU box with the snapshot -- a box without them must not pass the (b) row by skipping it" % os.path.basename(exe))
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert p.returncode == 0 and p.stdout == "", p.stdout + p.stderr          # verbose = 0: the reference prints nothing either
    out = tmp_path / gen_dir
    for f in ("tinympc/tiny_data.h", "src/tiny_data.c", "src/tiny_main.c", "Makefile"):
        assert (out / f).exists(), f
    nx, nu, N = dims
    data = (out / "src" / "tiny_data.c").read_text()
    assert re.search(r"tiny_problem = \{\s*%d, %d, %d," % (nx, nu, N), data)
    lib = os.path.join(ROOT, "tinympc_amd")
    mk = subprocess.run(["make", "-C", str(out), f"TINYMPC_AMD_LIB={lib}", f"TINYMPC_AMD_INC={os.path.join(ROOT, 'include')}"], capture_output=True, text=True)
    assert mk.returncode == 0, mk.stdout + mk.stderr
    B = 512
    r = subprocess.run([str(out / "tiny_main"), str(B)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"(\d+) ADMM iterations over (\d+) instances, (\d+) converged", r.stdout)
    # zero state, zero references, cold start: the solution is the origin, every instance converges at its first check
    assert m and int(m.group(2)) == B and int(m.group(3)) == B and int(m.group(1)) == B
    assert f"nx {nx} nu {nu} N {N}, batch {B}" in r.stdout and "Hooray" in r.stdout


def test_rho_benchmark_helpers_called_through_their_cxx_names():
    """tests/dropin/rho_driver.cpp (our caller, the reference's rho_benchmark.hpp): the seven C++-mangled helpers of the
    adaptive-rho module resolve against libtinympc_amd.so (csrc/rho_api.hip) and print what they print when the program is
    linked against the reference -- dimensions and flags exactly, every number to 1e-9 relative (summation order of the
    dense products differs, nothing else may)."""
    import re
    exe = os.path.join(BUILD, "rho_driver")
    if not os.path.exists(exe):
        pytest.fail("tests/dropin/_build/%s is missing: the drop-in binaries are built by __graft_entry__.build() in the container that has "
                    "/root/reference and travel to the GPU box with the snapshot -- a box without them must not pass the (b) row by skipping it" % os.path.basename(exe))
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    gold = open(os.path.join(ROOT, "tests", "golden", "stdout_rho_driver.txt")).read().splitlines()
    got = p.stdout.splitlines()
    assert len(got) == len(gold) == 15
    num = re.compile(r"-?\d+\.\d+e[+-]\d+")
    for a, b in zip(got, gold):
        assert num.sub("#", a) == num.sub("#", b), (a[:120], b[:120])        # words, integers, dimensions, flags
        for x, y in zip(num.findall(a), num.findall(b)):
            assert abs(float(x) - float(y)) <= 1e-9 * max(abs(float(y)), 1e-300) + 1e-12, (a[:120], b[:120])


def test_tiny_solve_from_eight_threads_on_distinct_solvers():
    """the reference's tiny_solve on DISTINCT solvers is re-entrant (its only global is the print format,
    tiny_api.cpp:11).  examples/dropin_threads.c: eight host threads, each on its own TinySolver (two problem families), 200
    closed-loop steps through tiny_set_x0 / tiny_solve -- iteration totals and final states identical bit for bit to the same episodes
    run one after the other, no deadlock, and the throughput ratio printed (the drop-in contexts have their own locks since round 6)."""
    import re
    exe = os.path.join(ROOT, "examples", "_build", "dropin_threads")
    if not os.path.exists(exe):
        pytest.fail("examples/_build/dropin_threads is missing: __graft_entry__.build() produces it")
    p = subprocess.run([exe, "8", "200"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert sum("== serial" in ln for ln in lines) == 8 and not any("DIFFERS" in ln for ln in lines), p.stdout
    assert lines[-1] == "OK"
    m = re.search(r"throughput ratio ([0-9.]+)", p.stdout)
    assert m and float(m.group(1)) > 0.5, p.stdout                      # (concurrent calls must at least not collapse; the figure goes to profiles/)
    fams = {re.search(r"\(nx (\d+)", ln).group(1) for ln in lines if ln.startswith("solver")}
    assert fams == {"4", "6"}
