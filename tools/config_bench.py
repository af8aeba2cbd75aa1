#!/usr/bin/env python3
"""Throughput of BASELINE configs 3 and 4 at full size (config 2 = bench.py, config 5 = sweep_bench.py), on one GPU or
sharded over the GPUs of a node:

    python tools/config_bench.py out.json [config3,config4,...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
           tools/config_bench.py out.json config3,config4

Under torch.distributed.run the TOTAL batch of the config is sharded round-robin by instance index
(tinympc_amd.distributed.shard_indices(interleaved=True): iteration counts diverge) -- every rank draws
the same seeded inputs and keeps its own instances --, there is no data-path collective, the timed region is closed by
the one 64-byte statistics exchange (RCCL) and the slowest rank's clock counts.

  config 3: quadrotor_tracking (12,4,10), 262 144 instances, per-instance random references around the y-axis line
            (SURVEY.md section 8(d) recipe), duals zeroed, ONE cold solve per instance (divergent iteration counts).
  config 4: rocket_landing (6,3,10), second-order-cone thrust constraint ON (en_input_soc = 1), 65 536 instances with
            perturbed initial states, the 90-step closed loop of examples/rocket_landing_mpc.cpp (plant on device,
            reference window refreshed per step as the example does), fused 1 launch per step.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm  # noqa: E402
from tinympc_amd.distributed import shard_indices  # noqa: E402

RANK = int(os.environ.get("RANK", "0"))
LOCAL_RANK = int(os.environ.get("LOCAL_RANK", "0"))
WORLD = int(os.environ.get("WORLD_SIZE", "1"))
_dist = None


def dist_init():
    """one process per GPU under torch.distributed.run; a no-op for a plain `python tools/config_bench.py`"""
    global _dist
    if WORLD > 1 or os.environ.get("TINYMPC_FORCE_DIST"):
        from tinympc_amd.distributed import init_process_group
        global LOCAL_RANK
        _dist, LOCAL_RANK = init_process_group(LOCAL_RANK)        # (LOCAL_RANK becomes the device index: smoke mode shares GPUs)
    return _dist


class Job:
    """The sharded form of one config: this rank's instance ids, a barrier, the statistics exchange and the slowest
    rank's wall clock -- all of them trivial when there is one rank."""

    def __init__(self, total):
        self.total = total
        self.idx = np.array(shard_indices(total, RANK, WORLD, interleaved=True))
        self.exchange = None
        if _dist is not None:
            import torch
            self.one = torch.zeros(1, device=f"cuda:{LOCAL_RANK}")

    def attach(self, solver):
        if _dist is not None:
            from tinympc_amd.distributed import StatsExchange
            self.exchange = StatsExchange(solver, _dist, LOCAL_RANK, total_batch=self.total)
        self.solver = solver

    def barrier(self):
        self.solver.synchronize()
        if _dist is not None:
            import torch
            _dist.all_reduce(self.one)
            torch.cuda.synchronize()

    def stats(self):
        """job-wide statistics (the one exchange of the path when sharded)"""
        return self.exchange().numpy() if self.exchange is not None else self.solver.reduce_stats()

    def slowest(self, seconds):
        if _dist is None:
            return seconds
        import torch
        t = torch.tensor([seconds], dtype=torch.float64, device=f"cuda:{LOCAL_RANK}")
        _dist.all_reduce(t, op=_dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.exchange is not None:
            self.exchange.close()


def apply_opts(s):
    """TINYMPC_OPTS="one_shot=2,grid_waves_per_cu=8": solver options for an experiment, applied to every solver here."""
    for kv in filter(None, os.environ.get("TINYMPC_OPTS", "").split(",")):
        k, v = kv.split("=")
        s.set_option(k, int(v))


def config3(B=262144, reps=3):
    prob, extra = tm.load_problem("quadrotor_20hz")
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    traj = np.array(extra["y_axis_line"])
    rng = np.random.default_rng(20260923)
    k = rng.integers(0, 291, B)
    Xref = traj[k[:, None] + np.arange(N)[None, :]].transpose(0, 2, 1) + rng.normal(0, 0.05, (B, nx, N))
    Uref = rng.normal(0, 0.05, (B, nu, N - 1))
    x0 = Xref[:, :, 0].copy()
    x0[:, :3] += rng.normal(0, 0.1, (B, 3))
    job = Job(B)
    if WORLD > 1:
        return config3_sharded(job, prob, Xref[job.idx], Uref[job.idx], x0[job.idx], reps)
    s = tm.TinyBatchSolver.from_problem(prob, B)
    apply_opts(s)
    s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=100)
    s.set_x_ref(Xref)
    s.set_u_ref(Uref)
    best, one_shot = None, {}
    s.set_option("repack_after", 0)           # the plain launch (the default is the automatic split, measured below)
    for _ in range(reps):
        s.reset()
        s.set_x0(x0)
        s.set_option("timing", 1)
        s.solve_async()
        ms = float(s.timing_ms()[0])
        best = ms if best is None else min(best, ms)
    st = s.reduce_stats()
    it = s.status()["iter"]
    alg = s.algorithmic_bytes()
    for mode in (1, 2):                       # the same solves as one-shot launches: nothing of the zero state is read
        s.set_option("one_shot", mode)
        bm = None
        for _ in range(reps):
            s.reset()                         # (only so that the accumulated statistics restart)
            s.set_x0(x0)
            s.set_option("timing", 1)
            s.solve_async()
            ms = float(s.timing_ms()[0])
            bm = ms if bm is None else min(bm, ms)
        so = s.reduce_stats()
        assert so[0] == st[0], "one-shot iteration total differs"
        one_shot[f"one_shot={mode}"] = dict(kernel_ms=bm, solves_per_s=B / (bm * 1e-3), admm_iters_per_s=so[0] / (bm * 1e-3),
                                            bytes_per_solve=s.algorithmic_bytes(cold=(2 if mode == 1 else 1)))
    s.set_option("one_shot", 0)
    repack = {}
    for cap in (9, 10, 11, 12, 16, 24):      # split solves: stop at `cap`, compact the open instances, finish them densely packed
        s.set_option("repack_after", cap)
        bm = None
        for _ in range(reps):
            s.reset()
            s.set_x0(x0)
            s.set_option("timing", 1)
            s.solve_async()
            ms = float(s.timing_ms()[0])
            bm = ms if bm is None else min(bm, ms)
        so = s.reduce_stats()
        assert so[0] == st[0] and so[1] == st[1] and np.array_equal(s.status()["iter"], it), "split solve differs"
        repack[f"repack_after={cap}"] = dict(kernel_ms=bm, solves_per_s=B / (bm * 1e-3), admm_iters_per_s=so[0] / (bm * 1e-3))
    # the default: automatic split -- plain solves first (timed, histogram), then the proposed K, kept if the clock confirms it
    s.set_option("repack_after", -1)
    auto_ms = []
    for _ in range(10):
        s.reset()
        s.set_x0(x0)
        s.set_option("timing", 1)
        s.solve_async()
        auto_ms.append(float(s.timing_ms()[0]))
    so = s.reduce_stats()
    assert so[0] == st[0] and np.array_equal(s.status()["iter"], it), "automatic split differs"
    auto = dict(kernel_ms_per_solve=auto_ms, kernel_ms=min(auto_ms[4:]), admm_iters_per_s=so[0] / (min(auto_ms[4:]) * 1e-3),
                K=s.get_option("auto_split_k"), predicted=s.get_option("auto_split_permille") / 1000.0,
                measured=s.get_option("auto_split_measured_permille") / 1000.0, verdict=s.get_option("auto_split_verdict"))
    s.close()
    t = best * 1e-3
    return dict(one_shot=one_shot, repack=repack, automatic_split=auto, config="quadrotor_tracking x262144, per-instance random refs, one cold solve (kernel_ms: plain launch, repack_after = 0)", batch=B, kernel_ms=best,
                solves_per_s=B / t, admm_iters_per_s=st[0] / t, iters_per_solve=st[0] / B, solved_fraction=st[1] / B,
                iter_histogram={int(v): int(c) for v, c in zip(*np.unique(it, return_counts=True))},
                hbm_frac=alg * B / t / 8e12, fp64_frac=st[0] * tm.flops_per_iter(nx, nu, N) / t / 78.6e12)


def config3_sharded(job, prob, Xref, Uref, x0, reps):
    """config 3 on WORLD GPUs: the total batch sharded round-robin, one cold solve per instance, wall clock of the slowest rank
    around launch + statistics exchange"""
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    s = tm.TinyBatchSolver.from_problem(prob, len(job.idx), device=LOCAL_RANK)
    apply_opts(s)
    s.set_bound_constraints(np.full((nx, 1), -5.0), np.full((nx, 1), 5.0), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=100)
    s.set_x_ref(Xref)
    s.set_u_ref(Uref)
    job.attach(s)
    best, st = None, None
    for _ in range(reps + 1):
        s.reset()
        s.set_x0(x0)
        job.barrier()
        t0 = time.perf_counter()
        s.solve_async()
        st = job.stats()
        job.barrier()
        dt = job.slowest(time.perf_counter() - t0)
        best = dt if best is None else min(best, dt)
    job.close()
    s.close()
    return dict(config=f"quadrotor_tracking x{job.total}, per-instance random refs, one cold solve, sharded round-robin over {WORLD} GPUs",
                batch=job.total, n_gpus=WORLD, batch_this_rank=len(job.idx), seconds=best, solves_per_s=job.total / best,
                admm_iters_per_s=st[0] / best, iters_per_solve=st[0] / job.total, solved_fraction=st[1] / job.total,
                fp64_frac_per_gpu=st[0] * tm.flops_per_iter(nx, nu, N) / best / 78.6e12 / WORLD)


def config4(B=65536):
    prob, extra = tm.load_problem("rocket_landing_20hz")
    m = extra["mpc"]
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    rng = np.random.default_rng(20260923)
    x0 = 1.1 * np.array(m["xinit"]) * (1 + 0.05 * rng.uniform(-1, 1, (B, nx)))
    xinit, xg = np.array(m["xinit"], dtype=float), np.array(m["xg"], dtype=float)
    traj = np.stack([xinit + (xg - xinit) * float(i) / (m["NTOTAL"] - 1) for i in range(m["NTOTAL"])])   # Xref window source
    job = Job(B)                                  # BASELINE: "batch 64k, sharded 8 x MI355X" = 8 192 instances per GPU
    x0 = x0[job.idx]
    s = tm.TinyBatchSolver.from_problem(prob, len(job.idx), device=LOCAL_RANK)
    job.attach(s)
    apply_opts(s)
    s.set_bound_constraints(np.array(m["x_min"]), np.array(m["x_max"]), np.full((nu, 1), m["u_min"]), np.full((nu, 1), m["u_max"]))
    s.set_cone_constraints(m["state_cone"]["A"], m["state_cone"]["q"], m["state_cone"]["c"],
                           m["input_cone"]["A"], m["input_cone"]["q"], m["input_cone"]["c"])
    s.update_settings(abs_pri_tol=m["abs_pri_tol"], max_iter=m["max_iter"], en_input_soc=1)
    uref = np.zeros((nu, N - 1)); uref[2, :] = m["uref_z"]
    steps = m["NTOTAL"] - N
    out = {}
    for T in (1, steps):
        s.reset()
        s.set_u_ref(uref, broadcast=True)
        s.set_reference_trajectory(traj)          # examples/rocket_landing_mpc.cpp:111-113 (window k .. k+N-1)
        s.set_x0(x0)
        s.set_option("advance_x0", 1)
        s.set_option("steps_per_launch", T)
        job.barrier()
        t0 = time.perf_counter()
        for _ in range(steps // T):
            s.solve_async()
        st = job.stats()                          # (sharded: the one 64-byte exchange closes the timed region)
        job.barrier()
        dt = job.slowest(time.perf_counter() - t0)
        out[f"steps_per_launch={T}"] = dict(seconds=dt, solves_per_s=B * steps / dt, admm_iters_per_s=st[7] / dt,
                                            iters_per_solve=st[7] / (B * steps), solved_fraction=st[8] / (B * steps))
    job.close()
    s.close()
    return dict(config=f"rocket_landing x{B}, input SOC on, 90-step closed loop (wall clock incl. launches), sharded round-robin over {WORLD} GPU(s)",
                batch=B, n_gpus=WORLD, batch_this_rank=len(job.idx), **out)


def linear_example(B=65536, tv=False):
    """examples/quadrotor_(tv_)linear_constraints.cpp: altitude ceiling + total-thrust half-spaces, boxes off; one
    cold solve of B perturbed instances on the register-resident LIN kernel and on the coverage kernel."""
    prob, _ = tm.load_problem("quadrotor_20hz")
    nx, nu, N = prob["nx"], prob["nu"], prob["N"]
    rng = np.random.default_rng(1)
    x0 = np.array([-2.0, -2.0, 1.0] + [0.0] * 9) + rng.normal(0, 0.05, (B, nx))
    xg = np.array([2.0, 2.0, 4.0] + [0.0] * 9)
    Xref = np.stack([(1 - i / 49.0) * x0 + (i / 49.0) * xg for i in range(N)], axis=2)
    ax = np.zeros((1, nx)); ax[0, 2] = 1.0
    out = {}
    for name, force in (("regs", 0), ("cover", 1)):
        Bk = B if not force else B // 8
        s = tm.TinyBatchSolver.from_problem(prob, Bk)
        if tv:
            s.set_tv_linear_constraints(np.tile(ax, (N, 1)), np.linspace(1.1, 3.0, N).reshape(1, N), np.ones((N - 1, nu)), np.full((1, N - 1), 6.0))
            s.update_settings(max_iter=100, en_state_bound=0, en_input_bound=0, en_tv_state_linear=1, en_tv_input_linear=1)
        else:
            s.set_linear_constraints(ax, [3.0], np.ones((1, nu)), [6.0])
            s.update_settings(max_iter=100, en_state_bound=0, en_input_bound=0, en_state_linear=1, en_input_linear=1)
        s.set_option("force_general", force)
        s.set_x_ref(Xref[:Bk])
        ms = None
        for _ in range(2):                       # the first launch of a kernel also loads its code object
            s.reset()
            s.set_x0(x0[:Bk])
            s.set_option("timing", 1)
            s.solve_async()
            t_ = float(s.timing_ms()[0])
            ms = t_ if ms is None else min(ms, t_)
        st = s.reduce_stats()
        out[name] = dict(kernel=s.kernel_path(), batch=Bk, kernel_ms=ms, solves_per_s=Bk / (ms * 1e-3),
                         admm_iters_per_s=st[0] / (ms * 1e-3), iters_per_solve=st[0] / Bk)
        s.close()
    return dict(config=f"quadrotor {'time-varying ' if tv else ''}linear constraints example, one cold solve", **out)


if __name__ == "__main__":
    dist_init()
    runs = {"config3": config3, "config4": config4, "linear_static": lambda: linear_example(tv=False),
            "linear_tv": lambda: linear_example(tv=True)}
    default = ["config3", "config4"] if WORLD > 1 else list(runs)      # the linear examples are single-GPU experiments
    pick = sys.argv[2].split(",") if len(sys.argv) > 2 else default
    res = {k: runs[k]() for k in pick}
    if RANK == 0:
        print(json.dumps(res, indent=1))
        if len(sys.argv) > 1:
            json.dump(res, open(sys.argv[1], "w"), indent=1)
    if _dist is not None:
        _dist.destroy_process_group()
