#!/usr/bin/env python3
"""CPU only.  When the closed-loop fuzzer reports a deviation above its tolerance: is it the kernel or the problem?
Runs the ORACLE twice on the trial's instance -- once as drawn, once with the initial state perturbed by 1e-14 relative --
and prints how the difference between the two oracle runs grows from MPC step to MPC step.  A closed loop that amplifies
round-off by a factor per step (ADMM stopped at max_iter every step, cones and half-spaces active) does the same to the
1e-15 per-solve differences between the HIP path and the oracle.
    python tools/closed_loop_sensitivity.py <seed> [instance] [perturbation]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import scenarios as sc  # noqa: E402
from cpu_solvers import OracleSolver, build_oracle  # noqa: E402
from fuzz_closed_loop import draw, oracle_episode as run  # noqa: E402


if __name__ == "__main__":
    assert build_oracle()
    seed = int(sys.argv[1])
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-14
    d = draw(seed)
    print(f"seed {seed} shape {(d['nx'], d['nu'], d['N'])} instance {b} of {d['B']}, {d['T'] * d['launches']} MPC steps, max_iter {d['kw']['max_iter']}, perturbation {eps:g}")
    a, c = run(d, b, 0.0), run(d, b, eps)
    rel = lambda p, q: float(np.max(np.abs(p - q)) / max(np.max(np.abs(q)), 1e-300))
    for k, (p, q) in enumerate(zip(a, c)):
        print(f"  step {k:3d} it {p[2]:3d}/{q[2]:3d}: oracle vs perturbed oracle  x0 {rel(q[0], p[0]):.2e}  u {rel(q[1], p[1]):.2e}")
