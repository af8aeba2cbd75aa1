#!/usr/bin/env python3
"""CPU only, build container only (needs oracle/_ref built from /root/reference): the oracle against the REAL reference
on the same random configurations the GPU fuzzers use (tools/fuzz_parity.py random_suite) -- pins the checker itself
over the whole fuzzed space, not only on the committed golden suites.   python tools/fuzz_oracle_vs_reference.py [n] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import scenarios as sc
from cpu_solvers import OracleSolver, RefSolver, build_oracle, build_ref


from fuzz_parity import random_suite  # noqa: E402  (the generator only; nothing here touches the GPU)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    assert build_oracle() and build_ref() is not None
    bad = 0
    for seed in range(s0, s0 + n):
        suite, kw = random_suite(seed)
        if suite["config"]["check_termination"] == 0:
            continue                                    # the reference divides by zero (admm.cpp:312)
        a = sc.run_cases(RefSolver, suite)
        b = sc.run_cases(OracleSolver, suite)
        msg = None
        for k in ("iter", "sol_solved", "status"):
            if not np.array_equal(a[k].astype(int), b[k].astype(int)):
                msg = f"{k} {a[k].astype(int).tolist()} vs {b[k].astype(int).tolist()}"
        if msg is None:
            for k, v in a.items():
                if v.ndim >= 2 and k in b:
                    e = float(np.max(np.abs(b[k] - v)) / max(np.max(np.abs(v)), 1e-300))
                    if e > 1e-9:
                        msg = f"{k} off by {e:.2e}"
                        break
        if msg:
            bad += 1
            p = suite["problem"]
            print("MISMATCH seed", seed, (p["nx"], p["nu"], p["N"]), msg, flush=True)
    print(f"{n} trials, {bad} mismatches")
    sys.exit(1 if bad else 0)
