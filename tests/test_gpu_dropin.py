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
        pytest.skip("drop-in binaries are built in the container that has /root/reference")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    gold = open(os.path.join(ROOT, "tests", "golden", f"stdout_{ex}.txt")).read()
    got = p.stdout
    if got != gold:
        gl, ol = gold.splitlines(), got.splitlines()
        diff = [(i, a, b) for i, (a, b) in enumerate(zip(gl, ol)) if a != b][:5]
        pytest.fail(f"{ex}: stdout differs from the reference's ({len(gl)} vs {len(ol)} lines); first diffs: {diff}")
