"""TEST/BENCH INFRASTRUCTURE -- the checker behind every `configs` entry of bench.py's line.

bench.py's GPU leg (tools/bench_configs.py) leaves, per entry, a SAMPLE of its own input records (a few hundred instances of
the very batch it timed) together with what the GPU computed for them.  After the GPU legs, in processes of its own (one per
host core, spawn context: no HIP context is ever forked), this module

  * solves every sampled record with the ORACLE (oracle/liboracle.so, the pinned C restatement) and compares: per-instance
    iteration counts (exact) and the applied control u[:,0] (max relative error)          -> `parity_sample`
  * times the REAL reference (oracle/_ref/libtinympc_ref.so; the oracle when it is absent: kind "port") on the same
    records, about `seconds` of solve time per core and entry, setup / state reset outside the clock -> `cpu_baseline`

Entry kinds: "single" = one cold solve per record (configs 3 and 5; with `problems`: every record under its OWN problem data); "episode" =
a closed loop of `steps` MPC steps with a moving state-reference window (config 4: examples/rocket_landing_mpc.cpp:120-135); "tracking"
= the same with the duals reset before every solve (examples/quadrotor_tracking.cpp:77-106).
"""
import multiprocessing as mp
import os
import pickle
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)


def _pick(a, r, shared_ndim):
    a = np.asarray(a)
    return a if a.ndim == shared_ndim else a[r]


def _prepare(s, e, r):
    s["Xref"] = _pick(e["Xref"], r, 2)
    s["Uref"] = _pick(e["Uref"], r, 2)
    s["x"][:, 0] = e["x0"][r]


def _run_one(s, e, r):
    """-> (iterations [signed per step for an episode], u0 [nu] or [steps, nu], solves, total iterations)"""
    if e["kind"] == "episode":
        tot, it, u0, _ = s.closed_loop_traj(e["x0"][r], e["steps"], e["traj"])
        return it, u0, e["steps"], tot
    if e["kind"] == "tracking":
        # examples/quadrotor_tracking.cpp:77-106, step by step: work->Xref = the window k ... k + N - 1 (clamped at the trajectory's end),
        # work->y = 0, work->g = 0, tiny_set_x0, tiny_solve, x0 <- A x0 + B u[:,0] (+ f)
        prob, traj, T = e["problem"], np.asarray(e["traj"], dtype=np.float64), e["steps"]
        A, Bm, f = np.asarray(prob["A"]), np.asarray(prob["B"]), np.asarray(prob["f"]).reshape(-1)
        N = prob["N"]
        x = np.array(e["x0"][r], dtype=np.float64)
        its = np.zeros(T, dtype=np.int32)
        u0 = np.zeros((T, prob["nu"]))
        tot = 0
        for k in range(T):
            win = np.minimum(np.arange(k, k + N), len(traj) - 1)
            s["Xref"] = traj[win].T
            s["g"] = np.zeros_like(s["g"]); s["y"] = np.zeros_like(s["y"])
            s["x"][:, 0] = x
            s.solve()
            it = int(s.get("sol_iter"))
            its[k] = it if int(s.get("sol_solved")) else -it
            tot += it
            u0[k] = s["u"][:, 0]
            x = A @ x + Bm @ u0[k] + f
        return its, u0, T, tot
    s.solve()
    it = int(s.get("sol_iter"))
    return np.array([it if int(s.get("sol_solved")) else -it], dtype=np.int32), s["u"][:, 0].copy(), 1, it


def _worker(args):
    spec_path, wi, nw, seconds = args
    import scenarios as sc
    from cpu_solvers import OracleSolver, RefSolver, have_ref
    spec = pickle.load(open(spec_path, "rb"))
    out = {}
    for e in spec:
        n = len(e["x0"])
        mine = list(range(wi, n, nw))
        res = dict(records=mine, iters=[], u0=[], solves=0, total_iters=0, busy=0.0)
        if mine:
            # parity: the oracle on this worker's records, each from the cold state of tiny_setup
            cfg = sc.default_config(e["problem"], **e["cfg_kw"])
            own = e.get("problems")                      # per-instance problem data: every record has its own family (its own tiny_setup)
            s = None if own else sc.make_solver(OracleSolver, e["problem"], cfg)
            cold = None if own else s.snapshot()
            for r in mine:
                if own:
                    s = sc.make_solver(OracleSolver, own[r], cfg)
                else:
                    s.restore(cold)
                _prepare(s, e, r)
                it, u0, _, _ = _run_one(s, e, r)
                res["iters"].append(it)
                res["u0"].append(u0)
                if own:
                    s.close()
            if not own:
                s.close()
            # CPU baseline: the real reference on the same records, round and round until `seconds` of solve time
            cls = RefSolver if have_ref() else OracleSolver
            s = sc.make_solver(cls, own[mine[0]] if own else e["problem"], cfg)
            cold = s.snapshot()
            k = 0
            while res["busy"] < seconds:
                r = mine[0] if own else mine[k % len(mine)]      # (per-instance data: the setup is outside the clock, so one record's family is timed)
                k += 1
                s.restore(cold)
                _prepare(s, e, r)
                t0 = time.perf_counter()
                _, _, ns, ni = _run_one(s, e, r)
                res["busy"] += time.perf_counter() - t0
                res["solves"] += ns
                res["total_iters"] += ni
            s.close()
        out[e["name"]] = res
    return out


def run(spec_path, seconds=1.0, cores=None):
    from cpu_solvers import have_ref, build_oracle
    build_oracle()
    spec = pickle.load(open(spec_path, "rb"))
    cores = cores or (os.cpu_count() or 1)
    nmax = max(len(e["x0"]) for e in spec)
    nw = min(cores, nmax)
    ctx = mp.get_context("spawn")
    with ctx.Pool(nw) as pool:
        res = pool.map(_worker, [(spec_path, i, nw, seconds) for i in range(nw)])
    kind = "reference" if have_ref() else "port"
    out = {}
    for e in spec:
        name, n = e["name"], len(e["x0"])
        it = [None] * n
        u0 = [None] * n
        rate = irate = 0.0
        solves = iters = 0
        active = 0
        busy_max = 0.0
        for w in res:
            r = w[name]
            for j, rec in enumerate(r["records"]):
                it[rec], u0[rec] = r["iters"][j], r["u0"][j]
            if r["busy"] > 0:
                rate += r["solves"] / r["busy"]
                irate += r["total_iters"] / r["busy"]
                solves += r["solves"]; iters += r["total_iters"]; active += 1
                busy_max = max(busy_max, r["busy"])
        it = np.array(it); u0 = np.array(u0)
        g_it = np.asarray(e["gpu_iter"]); g_u0 = np.asarray(e["gpu_u0"])
        if e["kind"] in ("episode", "tracking"):         # GPU logs are [steps, n(, nu)]
            g_it = g_it.T; g_u0 = np.transpose(g_u0, (1, 0, 2))
        else:
            g_it = g_it.reshape(n, 1)
        scale = np.maximum(np.abs(u0).max(axis=-1, keepdims=True), 1e-12)
        rel = float((np.abs(g_u0 - u0) / scale).max())
        out[name] = dict(
            parity_sample=dict(instances=n, checker="oracle/liboracle.so (pinned to the reference by tests/test_oracle_golden.py)",
                               iter_sum_gpu=int(np.abs(g_it).sum()), iter_sum_oracle=int(np.abs(it).sum()),
                               iteration_count_mismatches=int((g_it != it).sum()), solves_compared=int(it.size),
                               max_rel_err_u0=rel,
                               note="per solve: |iterations| and the solved flag (sign) must be equal; u[:,0] relative to the solve's largest |u0| entry"),
            cpu_baseline=None if seconds <= 0 else dict(value=rate, unit="QP solves/s", cores=active, kind=kind, admm_iters_per_s=irate,
                              admm_iters_per_solve=iters / max(solves, 1),
                              sample="%d of this entry's own input records (%s), %d solves in %.1f s of solve time per core, one process per core, "
                                     "cold state restored and inputs set outside the clock" % (n, e["kind"], solves, busy_max)))
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(run(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)))
