"""TEST/BENCH INFRASTRUCTURE -- times the CPU path on the hover workload (bench.py's cpu_baseline leg).

One PROCESS per core (threads of one process contend on the reference's std::cout and malloc,
SURVEY.md section 6), each running `steps`-step quadrotor-hover closed-loop episodes from the cold
state of tiny_setup (examples/quadrotor_hovering.cpp) -- the SAME first-K-steps workload bench.py
times on the GPU, so ADMM iterations per solve are equal on both sides -- for a bounded time.  Like the
GPU side's cold start, building the fresh solver of an episode is outside the clock: only the
closed-loop solves are timed.  kind = "reference" runs the real TinyMPC
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
    busy = 0.0
    while True:
        s = sc.make_solver(cls, prob, cfg)          # fresh cold workspace per episode, like the example's main()
        s["Xref"] = np.tile(np.array(h["xref"], dtype=np.float64).reshape(-1, 1), (1, prob["N"]))
        t0 = time.perf_counter()
        total, _, _, _ = s.closed_loop(h["x0"], steps)
        busy += time.perf_counter() - t0
        s.close()
        solves += steps
        iters += total
        if busy >= seconds:
            break
    return solves, iters, busy


def physical_cores():
    """distinct (package, core) pairs of /proc/cpuinfo: os.cpu_count() counts hardware THREADS (SMT siblings share a core)"""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except OSError:
        return None


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
    rate = sum(r[0] / r[2] for r in res)             # every core runs concurrently: the rates add
    phys = physical_cores()
    return dict(value=rate, unit="QP solves/s", cores=cores, physical_cores=phys, kind=kind,
                sample=f"{solves // steps} closed-loop hover episodes x the first {steps} MPC steps from a cold start "
                       f"(quadrotor nx=12 nu=4 N=10), {wall:.1f} s of solve time per process, one process per hardware thread "
                       f"({cores} threads on {phys if phys else '?'} physical cores)",
                admm_iters_per_solve=iters / max(solves, 1),
                admm_iters_per_s=sum(r[1] / r[2] for r in res), best_single_core_solves_per_s=single)


if __name__ == "__main__":
    import json
    print(json.dumps(run(float(sys.argv[1]) if len(sys.argv) > 1 else 3.0)))
