#!/usr/bin/env python3
"""Is the reference's adaptive-rho path (admm.cpp:339-345, 397-423; rho_benchmark.cpp) deterministic?  

solve() declares `RhoAdapter adapter;` (admm.cpp:339) without initialising `matrices_initialized`; format_matrices
(rho_benchmark.cpp:57-59) sizes the adapter's Eigen matrices only when that indeterminate bool reads false.  This probe
runs the 100-step quadrotor hover episode with adaptive_rho = 1 on the REAL reference (oracle/_ref) in child processes:
  A. 20 times with the stack region of solve()'s frame painted with zeros before every tiny_solve (ref_stack_fill(0));
  B. with that region painted with 0xFF / 0x01 before every solve (the flag then reads true);
  C. unpainted (whatever the previous calls left), a few times;
and compares the per-step iteration sequences, final rho and exit codes.  It also runs the same episode on the plain-C
restatement (oracle/liboracle.so)."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))


def episode(kind, fill, steps=100):
    import scenarios as sc
    from cpu_solvers import OracleSolver, RefSolver
    cls = RefSolver if kind == "ref" else OracleSolver
    prob, extra = sc.load_problem("quadrotor_20hz")
    cfg = sc._hover_cfg(prob, extra)
    h = extra["hover"]
    s = sc.make_solver(cls, prob, cfg)
    tables = json.load(open(os.path.join(HERE, "..", "tinympc_amd", "data", "sensitivity_quadrotor.json")))
    if kind == "ref":
        s.init_sensitivity()
    else:
        s.set_sensitivity({k: np.array(tables[k]["data"]).reshape(tables[k]["cols"], tables[k]["rows"]).T for k in
                           ("dKinf_drho", "dPinf_drho", "dC1_drho", "dC2_drho")})
    s.set_adaptive_rho(1, 1.0, 100.0, 1)
    s["Xref"] = np.tile(np.array(h["xref"], dtype=np.float64).reshape(-1, 1), (1, prob["N"]))
    x0 = np.array(h["x0"], dtype=np.float64)
    its, rhos = [], []
    A, B = np.array(prob["A"]), np.array(prob["B"])
    for _ in range(steps):
        s["x"][:, 0] = x0
        if kind == "ref" and fill is not None:
            s._f("solve_fill")(s.h, fill)           # paint + solve in one C call: nothing runs in between
        else:
            s.solve()
        its.append(int(s.get("sol_iter")) * (1 if s.get("sol_solved") else -1))
        rhos.append(float(s.get("rho")))
        x0 = A @ x0 + B @ s["u"][:, 0]
    return dict(iters=its, total=int(np.abs(its).sum()), rho_final=rhos[-1], rho_first10=rhos[:10],
                K00=float(s["Kinf"][0, 0]), P22=float(s["Pinf"][2, 2]), x_final=x0.tolist())


def child(kind, fill):
    p = subprocess.run([sys.executable, __file__, "--child", kind, str(fill)], capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        return dict(exit=p.returncode, stderr=p.stderr[-300:])
    return json.loads(p.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        fill = None if sys.argv[3] == "None" else int(sys.argv[3])
        print(json.dumps(episode(sys.argv[2], fill)))
        sys.exit(0)
    runs = [child("ref", 0) for _ in range(20)]
    same = all(r == runs[0] for r in runs)
    print("A. stack painted with zeros before every solve, 20 runs: identical =", same)
    if "exit" in runs[0]:
        print("   child failed:", runs[0])
    print("   total iterations", runs[0].get("total"), " final rho", runs[0].get("rho_final"), " Kinf[0,0]", runs[0].get("K00"))
    print("   iterations per step:", runs[0].get("iters"))
    print("   rho after steps 0..9:", runs[0].get("rho_first10"))
    for fill in (255, 1):
        r = child("ref", fill)
        print(f"B. stack painted with 0x{fill:02X}:", {k: r[k] for k in r if k in ("exit", "stderr", "total", "rho_final")} if "exit" in r else
              ("same as A" if r == runs[0] else dict(total=r["total"], rho_final=r["rho_final"])))
    raw = [child("ref", -1) for _ in range(5)]
    print("C. unpainted stack, 5 runs:", ["same as A" if r == runs[0] else (r.get("exit"), r.get("total")) for r in raw])
    o = child("oracle", None)
    print("oracle restatement: iterations identical to A =", o.get("iters") == runs[0].get("iters"), " total", o.get("total"),
          " final rho", o.get("rho_final"), " |rho diff|", abs(o.get("rho_final", 0) - runs[0].get("rho_final", 0)),
          " max |x_final diff|", float(np.max(np.abs(np.array(o["x_final"]) - np.array(runs[0]["x_final"])))))
