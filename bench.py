#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on its quoted config: QP solves/s (+ ADMM iterations/s) for
65 536 batched quadrotor-hover instances (nx=12, nu=4, N=10) per MI355X.

A "step" = ONE batched tiny_solve over the whole per-GPU batch (one MPC step of the closed loop of
examples/quadrotor_hovering.cpp, warm-started from the previous step, plant advanced on device).
The timed region always starts from the cold state of tiny_setup, so K steps = the first K steps of
the reference's 100-step episode (K = 100 -> 882 ADMM iterations per instance).
State is resident in HBM before timing starts.  The K-step region is repeated (cold start, barrier,
K steps, statistics exchange, barrier) until at least --min-seconds of timed work has run; `value`
comes from the MEDIAN repetition, min / max are reported beside it.  Multi-GPU: one process per GPU,
batch sharded with no data-path collective (weak scaling: 65 536 instances per GPU); ONE RCCL
all-gather of the 64-byte statistics message (tiny_batch_allreduce_stats, the library's native exchange, on a
communicator bootstrapped over torch.distributed) closes every timed repetition.

Rooflines (all in the one JSON line):
  roofline         the bound that BINDS the timed launches: FP64 VALU issue when the launches carry many
                   ADMM iterations (cold / fused steps), HBM when they are single warm steps; `traffic` = HBM
                   bytes per launch measured with rocprofv3 PMC passes (profiles/traffic.json)
  roofline_hbm / roofline_fp64   both fractions of the timed launches, whichever binds
  regimes          an untimed replay of the reference episode with ONE launch per MPC step after the
                   timed region: cold steps 0-4 (FP64 fraction) and steady state steps 70-99 (HBM fraction
                   of real, algorithmic bytes -- every launch loads and stores the records)

  python bench.py --gpus 1 --steps 100 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29500 bench.py --gpus 8 --steps 100 --warmup 10
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable copy rate)
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector peak (spec)


def flops_per_iter(nx, nu, N):
    """SURVEY.md section 8 footnote 1 (box constraints only)."""
    S = nx * N + nu * (N - 1)
    return (4 * S + 2 * nx * nx + 3 * nx + (N - 1) * (2 * nx * nx + 4 * nx * nu + 2 * nu * nu + 2 * nu + 3 * nx)
            + (N - 1) * (2 * nx * nx + 4 * nx * nu + 2 * nx + 2 * nu) + 11 * S)


def steps_per_launch(steps, warmup=0, requested=0):
    """MPC steps fused into one launch of the TIMED region: `requested`, or (0 = auto) the largest divisor of --steps that is
    <= 100 (the reference episode length), so that the timed region is exactly --steps steps in whole launches.  The warm-up
    launches are cut the same way from --warmup on their own (it is untimed and ends in a cold start anyway), so `warmup` no
    longer constrains the choice.  None if a requested value does not divide --steps."""
    T = requested
    if T <= 0:
        T = max(d for d in range(1, 101) if steps % d == 0)
    T = max(1, T)
    if steps % T:
        return None
    return T


def traffic_per_launch(T, B):
    """PMC-measured HBM bytes per launch of the solve kernel (rocprofv3 FETCH_SIZE / WRITE_SIZE passes,
    tools/collect_profiles.py).  A launch loads and stores the records once however many MPC steps it fuses."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath) or B != 65536:
        return None
    try:
        t = json.load(open(tpath))
        for key in (("fused_launch", "fused_100_steps_launch") if T > 1 else ("per_step_launch",)):
            if key in t:
                return t[key]["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def main():
    # stdout carries exactly ONE line, the result JSON: libraries that chat on fd 1 (RCCL prints a version banner at
    # communicator creation) are pointed at stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=65536, help="instances PER GPU")
    ap.add_argument("--grid-waves-per-cu", type=int, default=int(os.environ.get("TINYMPC_GRID_WAVES_PER_CU", "0")))
    ap.add_argument("--dpp-mode", type=int, default=int(os.environ.get("TINYMPC_DPP_MODE", "2")))
    ap.add_argument("--steps-per-launch", type=int, default=int(os.environ.get("TINYMPC_STEPS_PER_LAUNCH", "0")),
                    help="closed-loop MPC steps fused into one kernel launch (ADMM state stays in registers); "
                         "0 = auto: the largest divisor of --steps and --warmup that is <= 100; 1 = one launch per step")
    ap.add_argument("--opt", action="append", default=[], help="solver option name=value (experiments)")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="repeat the timed K-step region until this much timed work has run (median reported)")
    ap.add_argument("--max-repeats", type=int, default=2000)
    ap.add_argument("--no-regimes", action="store_true", help="skip the untimed one-launch-per-step replay")
    ap.add_argument("--regimes", action="store_true", help="(kept for old command lines: the replay is always on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only supports dmabuf IPC (RCCL across ranks)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus > 1 and world == 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    # CPU baseline first (rank 0, N=1 only), before any HIP context exists in this process: the real reference on
    # every host core, the SAME workload as the timed region (the first --steps MPC steps of the episode from cold)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import cpu_baseline
        cpu = cpu_baseline.run(seconds=args.cpu_seconds, steps=args.steps)

    import numpy as np
    import torch
    import tinympc_amd as tm
    from tinympc_amd.distributed import StatsExchange

    if not torch.cuda.is_available() or tm.device_count() == 0:
        sys.exit("bench.py needs an MI355X: tinympc_amd has no CPU fallback")
    # TINYMPC_BENCH_SHARE_GPU=1: smoke test of the N > 1 control flow on a box with fewer GPUs than ranks -- the ranks share
    # the devices round-robin and talk over gloo (RCCL refuses two ranks on one device); the line it prints is not a measurement
    share_gpu = bool(os.environ.get("TINYMPC_BENCH_SHARE_GPU"))
    if share_gpu:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("TINYMPC_FORCE_DIST"):      # FORCE_DIST: exercise the RCCL path on one GPU
        import torch.distributed as dist
        if share_gpu:
            os.environ["TINYMPC_EXCHANGE"] = "torch"
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm

    prob, extra = tm.load_problem("quadrotor_20hz")
    h = extra["hover"]
    nx, nu, N, B = prob["nx"], prob["nu"], prob["N"], args.batch
    s = tm.TinyBatchSolver.from_problem(prob, B, device=local_rank)
    s.set_bound_constraints(np.full((nx, 1), h["x_min"]), np.full((nx, 1), h["x_max"]),
                            np.full((nu, 1), h["u_min"]), np.full((nu, 1), h["u_max"]))
    s.update_settings(max_iter=h["max_iter"])
    s.set_option("advance_x0", 1)
    s.set_option("grid_waves_per_cu", args.grid_waves_per_cu)
    s.set_option("dpp_mode", args.dpp_mode)
    for kv in args.opt:                              # experiments: --opt grid_waves_per_cu=8
        k, v = kv.split("=")
        s.set_option(k, int(v))
    T = steps_per_launch(args.steps, args.warmup, args.steps_per_launch)
    if T is None:
        sys.exit("--steps must be a multiple of --steps-per-launch")
    Tw = steps_per_launch(args.warmup) if args.warmup > 0 else 1      # the untimed warm-up steps, fused the same way on their own
    s.set_option("steps_per_launch", T)
    launches = args.steps // T
    stream = torch.cuda.Stream(device=local_rank)
    s.set_stream(stream.cuda_stream)                 # kernels, events and the RCCL collective share one stream
    xref = np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N))
    x0 = np.array(h["x0"], dtype=np.float64)
    dev = f"cuda:{local_rank}"
    stats = torch.zeros(10, dtype=torch.float64, device=dev)

    def cold_start():
        s.reset()
        s.set_x_ref(xref, broadcast=True)
        s.set_x0(x0, broadcast=True)

    one = torch.zeros(1, device=dev)
    exchange = StatsExchange(s, dist, local_rank, total_batch=world * B) if dist is not None else None

    def barrier():
        # no rank leaves before every rank has arrived: a one-element all-reduce between two device synchronisations
        # (dist.barrier() itself costs 0.5 ms on this stack, 25x the all-reduce: tools/dist_exchange_cost.py)
        torch.cuda.synchronize()
        if dist is not None:
            dist.all_reduce(one)
        torch.cuda.synchronize()

    def max_over_ranks(values):
        if dist is None:
            return list(values)
        t = torch.tensor(list(values), dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    def timed_repetition():
        """cold start (untimed) -> barrier -> EXACTLY --steps MPC steps + the statistics exchange -> barrier."""
        cold_start()
        s.set_option("timing", launches)             # HIP events around every timed solve kernel, on `stream`
        barrier()
        t0 = time.perf_counter()
        for _ in range(launches):
            s.solve_async()
        if dist is not None:                         # the one exchange of the path: a 64-byte message per rank, RCCL over xGMI
            st = exchange()
        else:
            s.reduce_stats_async(stats.data_ptr())
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is None:
            st = stats.to("cpu")
        return elapsed, st.tolist(), s.timing_ms()

    with torch.cuda.stream(stream):
        cold_start()
        s.set_option("steps_per_launch", Tw)
        for _ in range(args.warmup // Tw):
            s.solve_async()
        s.synchronize()
        s.set_option("steps_per_launch", T)
        if dist is not None:                         # first-use costs of the collective stay out of the timed region
            exchange()
            barrier()
        e0, st, km = timed_repetition()
        e0 = max_over_ranks([e0])[0]
        repeats = int(min(max(3, -(-args.min_seconds // max(e0, 1e-6))), args.max_repeats))
        rep_s, rep_kernel_ms = [e0], [km]
        for _ in range(repeats - 1):
            e, st, km = timed_repetition()
            rep_s.append(e)
            rep_kernel_ms.append(km)
    rep_s = np.array(max_over_ranks(rep_s))          # every repetition: the slowest rank's clock
    elapsed = float(np.median(rep_s))
    acc_iters, acc_solved, max_resid = st[7], st[8], st[3:7]        # job-wide, one repetition (each starts with a reset)
    kern_ms = np.concatenate(rep_kernel_ms)          # this rank's solve-kernel durations, all repetitions
    kern_first = np.array([k[0] for k in rep_kernel_ms])
    kern_sum_rep = float(np.median([k.sum() for k in rep_kernel_ms]))

    # Untimed replay (rank 0): the same episode with ONE launch per MPC step, so that the two regimes SURVEY.md 8(d) asks
    # for are visible in every line -- cold steps (100 ADMM iterations each, FP64 bound) and steady state (1-2 iterations:
    # every launch loads and stores the records, the real HBM roofline of this path).  Median over five replays.
    regimes = None
    fl = flops_per_iter(nx, nu, N)
    bytes_warm = s.algorithmic_bytes(cold=False)
    if rank == 0 and not args.no_regimes:
        def replay(store_primal):
            runs = []
            s.set_option("store_primal", store_primal)
            for _ in range(5):
                cold_start()
                s.set_option("timing", 100)
                for _ in range(100):
                    s.solve_async()
                s.synchronize()
                runs.append(s.timing_ms())
            s.set_option("store_primal", 1)
            return np.median(np.array(runs), axis=0)
        with torch.cuda.stream(stream):
            s.set_option("steps_per_launch", 1)
            ms = replay(1)
            ms_lean = replay(0)
            ms_u0 = replay(2)
            s.set_option("steps_per_launch", T)
        S = nx * N + nu * (N - 1)
        cold = float(ms[:5].mean()) * 1e-3
        warm = float(ms[70:].mean()) * 1e-3
        lean = float(ms_lean[70:].mean()) * 1e-3
        first = float(ms_u0[70:].mean()) * 1e-3
        regimes = {
            "cold": {"steps": "0-4", "admm_iters_per_solve": 100, "ms_per_launch": cold * 1e3,
                     "fp64_tflops": 100 * B * fl / cold / 1e12, "fp64_frac": 100 * B * fl / cold / 1e12 / FP64_PEAK_TFLOPS},
            "steady_state": {"steps": "70-99", "ms_per_launch": warm * 1e3, "ms_per_launch_min": float(ms[70:].min()),
                             "algorithmic_bytes_per_solve": bytes_warm,
                             "hbm_gbs": bytes_warm * B / warm / 1e9, "hbm_frac": bytes_warm * B / warm / 1e9 / HBM_PEAK_GBS,
                             "note": "one launch per MPC step; bytes_warm = 8(nx+8S)+44 per solve (SURVEY.md 8(d)) / launch time; a "
                                     "solve that converges at its first check does not store v|z again (admm.cpp:431-441 returns "
                                     "before v = vnew), so the bytes really moved are up to 8S = %d B per solve lower" % (8 * S)},
            "steady_state_lean": {"steps": "70-99", "ms_per_launch": lean * 1e3, "ms_per_launch_min": float(ms_lean[70:].min()),
                                  "hbm_gbs": bytes_warm * B / lean / 1e9, "hbm_frac": bytes_warm * B / lean / 1e9 / HBM_PEAK_GBS,
                                  "bytes_moved_per_solve_max": bytes_warm - 8 * S,
                                  "first_knot_only": {"ms_per_launch": first * 1e3, "hbm_frac": bytes_warm * B / first / 1e9 / HBM_PEAK_GBS,
                                                      "note": "store_primal = 2: x[:,0], x[:,1], u[:,0] are still written (the control a caller applies)"},
                                  "note": "option store_primal = 0: x|u is not written back either (no consumer between steps: the "
                                          "plant step runs on the device, solution->x|u = vnew|znew is still stored); the figure is "
                                          "still bytes_warm / launch time, i.e. solves/s in the formula's units"}}
    solves = float(world) * B * args.steps
    value = solves / elapsed
    avg_kernel_s = float(kern_ms.mean()) * 1e-3
    # real HBM bytes of one launch: the records are loaded and stored once however many MPC steps it fuses
    hbm_gbs = bytes_warm * B / avg_kernel_s / 1e9
    iters_local = acc_iters / world                  # one repetition, this GPU
    fp64_tflops = iters_local * fl / (kern_sum_rep * 1e-3) / 1e12
    traffic = traffic_per_launch(T, B)
    common = {"traffic": traffic, "kernel": "admm_solve_kernel<12,4,10>", "avg_launch_ms": avg_kernel_s * 1e3,
              "launches_per_repetition": launches, "mpc_steps_per_launch": T}
    roofline_hbm = dict({"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": bytes_warm * B,
                         "note": "bytes_warm = 8(nx+8S)+44 = %d B per instance and LAUNCH (the ADMM state stays in registers "
                                 "between the MPC steps a launch fuses) / average launch time" % bytes_warm}, **common)
    roofline_fp64 = dict({"bound": "fp64-valu", "achieved": fp64_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": fp64_tflops / FP64_PEAK_TFLOPS, "flops_per_admm_iter": fl,
                          "admm_iters_per_launch_and_instance": iters_local / B / launches,
                          "note": "ADMM iterations x %d FLOP (SURVEY.md 8 footnote 1) / summed kernel time of one repetition" % fl},
                         **common)
    binding = roofline_fp64 if roofline_fp64["frac"] >= roofline_hbm["frac"] else roofline_hbm
    if rank == 0:
        out = {
            "metric": "QP solves/sec (+ ADMM iters/sec), 64k-batch quadrotor hover",
            "value": value, "unit": "QP solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "quadrotor_hovering (nx=12, nu=4, N=10), 65536 identical instances per GPU, "
                                   "closed-loop MPC steps from a cold start (BASELINE configs[1])",
                       "batch_per_gpu": B, "parallelism": f"batch-sharded x{world}",
                       "grid_waves_per_cu": args.grid_waves_per_cu, "dpp_mode": args.dpp_mode,
                       "mpc_steps_per_launch": T,
                       "stats_exchange": (exchange.kind if exchange is not None else "none (one rank)")},
            "timed_region": {"repeats": int(repeats), "seconds_total": float(rep_s.sum()),
                             "ms": {"median": elapsed * 1e3, "min": float(rep_s.min()) * 1e3, "max": float(rep_s.max()) * 1e3},
                             "value_min": solves / float(rep_s.max()), "value_max": solves / float(rep_s.min()),
                             "note": "each repetition = cold start (untimed), barrier, exactly --steps MPC steps + the statistics "
                                     "exchange, barrier; value and ms_per_step come from the median repetition"},
            "admm_iters_per_s": acc_iters / elapsed,
            "admm_iters_per_solve": acc_iters / solves,
            "solved_fraction": acc_solved / solves,
            "max_residuals": max_resid,
            "roofline": binding,
            "roofline_hbm": roofline_hbm,
            "roofline_fp64": roofline_fp64,
            "kernel_ms": {"first_launch_median": float(np.median(kern_first)), "sum_per_repetition_median": kern_sum_rep,
                          "min": float(kern_ms.min()), "max": float(kern_ms.max()), "count": int(kern_ms.size)},
        }
        if share_gpu:
            out["data"] = "synthetic; SMOKE RUN: %d ranks share %d GPU(s) over gloo -- not a measurement" % (world, torch.cuda.device_count())
        if regimes is not None:
            out["regimes"] = regimes
        if cpu is not None:
            out["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if exchange is not None:
        exchange.close()
    s.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
