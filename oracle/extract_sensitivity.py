#!/usr/bin/env python3
"""TEST/DATA INFRASTRUCTURE: the adaptive-rho sensitivity tables of the reference as DATA.

tiny_initialize_sensitivity_matrices (src/tinympc/tiny_api.cpp:479-540) fills cache->dKinf_drho (4 x 12), dPinf_drho
(12 x 12), dC1_drho (4 x 4) and dC2_drho (12 x 12) from float literals through column-major Eigen maps over row-major
arrays (so the 4 x 12 table arrives re-shuffled, not transposed).  Rather than restating that by hand, this script runs
the REAL function (oracle/_ref/libtinympc_ref.so) on a quadrotor solver and stores the four matrices exactly as it
leaves them -- column-major doubles -- in tinympc_amd/data/sensitivity_quadrotor.json.  The library's own
tiny_initialize_sensitivity_matrices and the oracle's tests read that file (the Makefile turns it into an include)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scenarios as sc  # noqa: E402
from cpu_solvers import RefSolver, build_ref  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tinympc_amd", "data", "sensitivity_quadrotor.json")


def main():
    if build_ref() is None:
        sys.exit("needs /root/reference (the build container)")
    prob, _ = sc.load_problem("quadrotor_20hz")
    s = sc.make_solver(RefSolver, prob, sc.default_config(prob))
    assert s._f("init_sensitivity")(s.h) == 0
    out = {"nx": 12, "nu": 4, "source": "tiny_initialize_sensitivity_matrices, src/tinympc/tiny_api.cpp:479-540 (run, not restated)",
           "layout": "column-major, as Eigen stores them"}
    for k in ("dKinf_drho", "dPinf_drho", "dC1_drho", "dC2_drho"):
        m = s[k]                                    # numpy view (rows, cols)
        out[k] = {"rows": int(m.shape[0]), "cols": int(m.shape[1]), "data": np.asarray(m).T.ravel().tolist()}
    s.close()
    json.dump(out, open(OUT, "w"))
    print("wrote", os.path.normpath(OUT), {k: (v["rows"], v["cols"]) for k, v in out.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
