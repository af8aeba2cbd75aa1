#!/usr/bin/env python3
"""Extract the numeric problem definitions of the reference examples into DATA files.

Reads (never copies source from) /root/reference/examples/problem_data/*.hpp,
examples/trajectory_data/quadrotor_20hz_y_axis_line.hpp and the literal constants
of examples/cartpole_example.cpp:32-37 / examples/codegen_random.cpp:32-37, and
writes tinympc_amd/data/problems.json -- plain numbers (repr round-trips doubles
exactly).  /root/reference does not exist on the GPU box, so the JSON is committed.

Conventions:
  * matrices are stored row-major nested lists, A is (nx,nx), B is (nx,nu);
  * Q, R are the USER diagonals (rho not added);
  * the rocket header uses f-suffixed float literals
    (examples/problem_data/rocket_landing_params_20hz.hpp:7-29): values are rounded to
    float32 first and then widened, exactly what the C++ compiler does.
"""
import json
import os
import re
import sys

import numpy as np

REF = os.environ.get("TINYMPC_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tinympc_amd", "data", "problems.json")


def parse_arrays(path):
    """name -> list of python floats, honouring an 'f' suffix (float32 rounding)."""
    text = open(path).read()
    out = {}
    for m in re.finditer(r"tinytype\s+(\w+)\s*(?:\[[^\]]*\])?\s*=\s*\{?([^;]*?)\}?\s*;", text, re.S):
        name, body = m.group(1), m.group(2)
        vals = []
        for tok in re.findall(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?f?", body):
            if tok.endswith("f"):
                vals.append(float(np.float32(tok[:-1])))
            else:
                vals.append(float(tok))
        out[name] = vals
    return out


def main():
    probs = {}
    q = parse_arrays(f"{REF}/examples/problem_data/quadrotor_20hz_params.hpp")
    nx, nu = 12, 4
    probs["quadrotor_20hz"] = dict(
        nx=nx, nu=nu, N=10, rho=q["rho_value"][0],
        A=np.array(q["Adyn_data"]).reshape(nx, nx).tolist(),
        B=np.array(q["Bdyn_data"]).reshape(nx, nu).tolist(),
        f=[0.0] * nx, Q=q["Q_data"], R=q["R_data"],
        source="examples/problem_data/quadrotor_20hz_params.hpp:5-37",
        # examples/quadrotor_hovering.cpp:41-66
        hover=dict(x_min=-5.0, x_max=5.0, u_min=-0.5, u_max=0.5, max_iter=100,
                   x0=[0, 1, 0, 0.2, 0, 0, 0.1, 0, 0, 0, 0, 0], xref=[0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0], steps=100),
    )
    t = parse_arrays(f"{REF}/examples/trajectory_data/quadrotor_20hz_y_axis_line.hpp")
    xr = np.array(t["Xref_data"]).reshape(301, nx)   # [NTOTAL][NSTATES]; column of the Eigen map = time step
    probs["quadrotor_20hz"]["y_axis_line"] = xr.tolist()

    r = parse_arrays(f"{REF}/examples/problem_data/rocket_landing_params_20hz.hpp")
    nx, nu = 6, 3
    probs["rocket_landing_20hz"] = dict(
        nx=nx, nu=nu, N=10, rho=r["rho_value"][0],
        A=np.array(r["Adyn_data"]).reshape(nx, nx).tolist(),
        B=np.array(r["Bdyn_data"]).reshape(nx, nu).tolist(),
        f=r["fdyn_data"], Q=r["Q_data"], R=r["R_data"],
        source="examples/problem_data/rocket_landing_params_20hz.hpp:5-29",
        # examples/rocket_landing_mpc.cpp:59-64,69-98,104-123
        mpc=dict(x_min=[-5.0, -5.0, -0.5, -10.0, -10.0, -20.0], x_max=[5.0, 5.0, 100.0, 10.0, 10.0, 20.0],
                 u_min=-10.0, u_max=105.0, max_iter=100, abs_pri_tol=2e-3,
                 xinit=[4, 2, 20, -3, 2, -4.5], xg=[0, 0, 0, 0, 0, 0], uref_z=10.0, NTOTAL=100,
                 # effective cone parameters after the argument swap at rocket_landing_mpc.cpp:94
                 # (tiny_api.hpp:16-18 vs tiny_api.cpp:176-178): first triple binds to the STATE cone
                 state_cone=dict(A=[0], q=[3], c=[0.25]), input_cone=dict(A=[0], q=[3], c=[0.5])),
    )
    # examples/cartpole_example.cpp:32-37 (B is mapped column-major there; it is nx x 1 so no difference)
    probs["cartpole"] = dict(
        nx=4, nu=1, N=10, rho=1.0,
        A=[[1.0, 0.01, 0.0, 0.0], [0.0, 1.0, 0.039, 0.0], [0.0, 0.0, 1.002, 0.01], [0.0, 0.0, 0.458, 1.002]],
        B=[[0.0], [0.02], [0.0], [0.067]], f=[0.0] * 4, Q=[10.0, 1.0, 10.0, 1.0], R=[1.0],
        source="examples/cartpole_example.cpp:32-37",
        mpc=dict(x_min=-1e17, x_max=1e17, u_min=-1e17, u_max=1e17, max_iter=100,
                 x0=[0.5, 0, 0, 0], xref=[1.0, 0, 0, 0], steps=390),
    )
    # examples/codegen_random.cpp:32-37 (cache known-answer test of SURVEY.md section 8(c))
    probs["codegen_random"] = dict(
        nx=2, nu=2, N=3, rho=0.1, A=[[1.0, 5.0], [1.0, 2.0]], B=[[3.0, 3.0], [4.0, 1.0]], f=[0.0, 0.0],
        Q=[1.0, 1.0], R=[2.0, 2.0], source="examples/codegen_random.cpp:32-37")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as fh:
        json.dump(probs, fh)
    print("wrote", os.path.normpath(OUT), {k: (v["nx"], v["nu"], v["N"]) for k, v in probs.items()})


if __name__ == "__main__":
    sys.exit(main())
