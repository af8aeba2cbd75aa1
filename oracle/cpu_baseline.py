"""TEST/BENCH INFRASTRUCTURE -- times the CPU path on the hover workload (bench.py's cpu_baseline leg).

One PROCESS per core (threads of one process contend on the reference's std::cout and malloc,
SURVEY.md section 6), each running whole 100-step quadrotor-hover closed-loop episodes
(examples/quadrotor_hovering.cpp) for a bounded wall time.  kind = "reference" runs the real TinyMPC
(oracle/_ref/libtinympc_ref.so, stdout muted) when that library exists, else "port" runs the C
restatement (oracle/liboracle.so).
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)


def _worker(args):
    kind, seconds, steps = args
    import scenarios as sc
    from cpu_solvers import OracleSolver, RefSolver
    cls = RefSolver if kind == "reference" else OracleSolver
    prob, extra = sc.load_problem("quadrotor_20hz")
    cfg = sc._hover_cfg(prob, extra)
    h = extra["hover"]
    solves = iters = 0
    t0 = time.perf_counter()
    while True:
        s = sc.make_solver(cls, prob, cfg)          # fresh cold workspace per episode, like the example's main()
        s["Xref"] = np.tile(np.array(h["xref"], dtype=np.float64).reshape(-1, 1), (1, prob["N"]))
        total, _, _, _ = s.closed_loop(h["x0"], steps)
        s.close()
        solves += steps
        iters += total
        if time.perf_counter() - t0 >= seconds:
            break
    return solves, iters, time.perf_counter() - t0


def run(seconds=8.0, steps=100, cores=None):
    from cpu_solvers import have_ref, build_oracle
    kind = "reference" if have_ref() else "port"
    if kind == "port":
        build_oracle()
    cores = cores or (os.cpu_count() or 1)
    ctx = mp.get_context("spawn")                    # never fork a process that may hold a HIP context
    with ctx.Pool(cores) as pool:
        res = pool.map(_worker, [(kind, seconds, steps)] * cores)
    wall = max(r[2] for r in res)
    solves = sum(r[0] for r in res)
    iters = sum(r[1] for r in res)
    single = max(r[0] / r[2] for r in res)
    return dict(value=solves / wall, unit="QP solves/s", cores=cores, kind=kind,
                sample=f"{solves // steps} closed-loop hover episodes x {steps} MPC steps "
                       f"(quadrotor nx=12 nu=4 N=10, {iters / max(solves, 1):.2f} ADMM iters/solve) over {wall:.1f} s, "
                       f"one process per core",
                admm_iters_per_s=iters / wall, best_single_core_solves_per_s=single)


if __name__ == "__main__":
    import json
    print(json.dumps(run(float(sys.argv[1]) if len(sys.argv) > 1 else 3.0)))
