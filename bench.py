#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on its quoted config: QP solves/s (+ ADMM iterations/s) for
65 536 batched quadrotor-hover instances (nx=12, nu=4, N=10).

A "step" = ONE batched tiny_solve over the whole per-GPU batch (one MPC step of the closed loop of
examples/quadrotor_hovering.cpp, warm-started from the previous step, plant advanced on device).
The timed region always starts from the cold state of tiny_setup, so K steps = the first K steps of
the reference's 100-step episode (K = 100 -> 882 ADMM iterations per instance).
State is resident in HBM before timing starts.  The K-step region is repeated (cold start, barrier,
K steps, statistics exchange, barrier) until at least --min-seconds of timed work has run; `value`
comes from the MEDIAN repetition, min / max are reported beside it.

How it starts -- all three forms run the same code:
  python bench.py --gpus 1 ...                             one process, one GPU
  python bench.py --gpus N ...                             a PLAIN process: it spawns N ranks of itself with
                                                           torch.distributed.run on 127.0.0.1 (one process per GPU) and
                                                           passes rank 0's one JSON line through
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N ...                             the ranks are already there (RANK / WORLD_SIZE in the env)

Multi-GPU: one process per GPU, the batch sharded with NO data-path collective; ONE RCCL all-gather of the 64-byte
statistics message (tiny_batch_allreduce_stats, the library's native exchange, on a communicator bootstrapped over
torch.distributed) closes every timed repetition.  --scaling weak (default): 65 536 instances PER GPU;
--scaling strong: 65 536 instances in TOTAL, sharded (at 8 GPUs: 8 192 per GPU = exactly one resident wave set).
Under weak scaling with N > 1 the strong-scaling form is measured as well, after the headline, and reported as
`strong_scaling` in the same line.

Output: stdout carries exactly ONE line, compact (compact_line: <= 4 KiB of strict JSON -- the contract's keys, `roofline`,
`cpu_baseline`, a few numbers per regime and per `configs` entry); the full record described below goes to
gpurun_out/bench_details.json (the line names it under `details`).

Rooflines (full record; the line keeps frac / achieved / peak / kernel / avg_launch_ms of each):
  roofline         the bound that BINDS the timed launches: FP64 VALU issue when the launches carry many
                   ADMM iterations (cold / fused steps), HBM when they are single warm steps; `traffic` = HBM
                   bytes per launch measured with rocprofv3 PMC passes (profiles/traffic.json, `traffic_source`)
  roofline_hbm / roofline_fp64   both fractions of the timed launches, whichever binds
  regimes          an untimed replay of the reference episode with ONE launch per MPC step after the
                   timed region: cold steps 0-4 (FP64 fraction) and steady state steps 70-99 -- `hbm_frac` counts the
                   bytes the launch form really moves, `hbm_frac_formula` SURVEY.md 8(d)'s bytes_warm
                   `regimes.beyond_l3`: the same replay at batch 262 144 (working set ~4x the 256 MiB Infinity Cache), the
                   line's `roofline_hbm` -- the warm regime where the records really come from HBM
  configs          (1 GPU) BASELINE configs 3, 4 (input cone / state cone / both) and six cells of config 5, each with its own
                   roofline (median), a parity sample against the oracle and the reference timed on the same records
  cpu_baseline     (1 GPU) the real reference on the host cores, AFTER the GPU legs, in processes of its own
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable copy rate)
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector peak (spec)
TOTAL_BATCH = 65536          # BASELINE configs[1]


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


def shard_size(total, rank, world):
    """contiguous shards of a `total`-instance batch (identical instances: nothing to balance)"""
    base, rem = divmod(total, world)
    return base + (1 if rank < rem else 0)


def moved_bytes_per_solve(bytes_warm, S, nx, nu, iters, ref_shared, store_primal):
    """Bytes one warm solve of the default launch form REALLY moves: SURVEY.md 8(d)'s bytes_warm minus what the form provably
    skips -- the Xref|Uref read when all instances share one L2-resident record (-8S), the v|z store of a solve that converges
    at its first check (admm.cpp:431-441 returns before v = vnew: -8S), the x|u store with store_primal = 0 (-8S) or all of it
    but the first knot with store_primal = 2."""
    b = bytes_warm
    if ref_shared:
        b -= 8 * S
    if iters == 1:
        b -= 8 * S
    if store_primal == 0:
        b -= 8 * S
    elif store_primal == 2:
        b -= 8 * (S - (2 * nx + nu))
    return b


def traffic_per_launch(T, B):
    """PMC-measured HBM bytes per launch of the solve kernel (rocprofv3 FETCH_SIZE / WRITE_SIZE passes,
    tools/collect_profiles.py), read from profiles/traffic.json -- a citation of an earlier profiling run, not something this
    run measures (`traffic_source` says which).  A launch loads and stores the records once however many MPC steps it fuses."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath) or B != 65536:
        return None, None
    try:
        t = json.load(open(tpath))
        # the PMC entry whose launches fuse the SAME number of MPC steps as this run's (the 100-step entry was cited for
        # the driver's 20-step launches); a launch form without an entry of its own has no citation
        keys = {1: ("per_step_launch",), 20: ("fused_launch_driver_flags",), 100: ("fused_launch", "fused_100_steps_launch")}.get(T, ())
        for key in keys:
            if key in t:
                return t[key]["hbm_bytes_per_launch"], "profiles/traffic.json[%s] (%s)" % (key, t[key].get("round", "r02 PMC run"))
    except Exception:
        pass
    return None, None


COMPACT_LINE_LIMIT = 4096    # bytes: the driver keeps a bounded tail of stdout (r04's 32 992-byte line was dropped)
DETAILS_PATH = os.path.join("gpurun_out", "bench_details.json")


def _r(x, digits=5):
    """a float with `digits` significant digits (None / non-finite -> None: the line must be strict JSON)"""
    if x is None:
        return None
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, x))


def _config_summary(e):
    """one `configs` entry of the details file -> the few numbers of it the line carries"""
    if not isinstance(e, dict):
        return None
    if "error" in e or "skipped" in e:
        return {"error": str(e.get("error", e.get("skipped")))[:80]}
    c = {"ms": _r(e.get("ms")), "frac": _r((e.get("roofline") or {}).get("frac"), 4), "path": e.get("kernel")}
    tr = e.get("traffic")
    if isinstance(tr, dict) and tr.get("ratio_traffic_over_algorithmic") is not None:
        c["traffic_ratio"] = _r(tr["ratio_traffic_over_algorithmic"], 3)
    ps = e.get("parity_sample")
    if isinstance(ps, dict):
        c["mismatches"] = ps.get("iteration_count_mismatches", ps.get("error"))
    if e.get("first_call_ms") is not None:
        c["first_call_ms"] = _r(e["first_call_ms"])
    if e.get("planned_first_call_ms") is not None:
        c["planned_first_call_ms"] = _r(e["planned_first_call_ms"])
    if e.get("plan_shipped"):
        c["plan"] = "shipped"                            # the first call already ran under tinympc_amd/data/plans.txt's entry
    return c


def compact_line(out, details_path=None):
    """The ONE stdout line: the contract's keys + `roofline` + `cpu_baseline` and a handful of numbers per regime / config, at
    most COMPACT_LINE_LIMIT bytes of strict JSON.  Everything else the run measured (timed_region, regimes, full `configs`
    entries, notes) goes to the details file this line names.  Pure function of `out` (tests/test_bench_cpu.py feeds it canned
    input): if the line still comes out too long, the optional blocks are dropped one by one, never the contract's keys."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(out.get("value"), 7), _r(out.get("ms_per_step"), 6)
    cfg = out.get("config") or {}
    line["config"] = {k: cfg.get(k) for k in ("workload", "batch_per_gpu", "total_batch", "parallelism", "mpc_steps_per_launch", "stats_exchange", "launcher")
                      if k in cfg}
    if "error" in out:
        line["error"] = str(out["error"])[:600]
    for k in ("rccl_ranks", "solved_fraction"):
        if k in out:
            line[k] = out[k]
    for k in ("admm_iters_per_s", "admm_iters_per_solve"):
        if k in out:
            line[k] = _r(out[k], 6)
    rf = out.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = {"bound": rf.get("bound"), "achieved": _r(rf.get("achieved")), "peak": rf.get("peak"), "unit": rf.get("unit"),
                            "frac": _r(rf.get("frac"), 4), "traffic": rf.get("traffic"),
                            "traffic_source": "cited" if rf.get("traffic") is not None else None,
                            "kernel": rf.get("kernel"), "avg_launch_ms": _r(rf.get("avg_launch_ms")),
                            "mpc_steps_per_launch": rf.get("mpc_steps_per_launch")}
    optional = []
    rh, rq = out.get("roofline_hbm"), out.get("roofline_fp64")
    if isinstance(rh, dict) and isinstance(rq, dict):
        line["roofline_timed"] = {"hbm_frac": _r(rh.get("frac"), 4), "fp64_frac": _r(rq.get("frac"), 4)}
        optional.append("roofline_timed")
    reg = out.get("regimes")
    if isinstance(reg, dict):
        w = {}
        for key, short in (("steady_state", "shared_ref"), ("steady_state_per_instance_refs", "own_refs")):
            r_ = reg.get(key)
            if isinstance(r_, dict):
                w[short] = {"hbm_frac": _r(r_.get("hbm_frac"), 4), "gbs": _r(r_.get("hbm_gbs"), 4), "ms": _r(r_.get("ms_per_launch"), 4)}
        if w:
            w["batch"] = (out.get("config") or {}).get("batch_per_gpu")
            w["beyond_L3"] = False
            line["warm_regime"] = w
            optional.append("warm_regime")
        big = reg.get("beyond_l3")
        if isinstance(big, dict) and "error" not in big:
            b_ = {"batch": big.get("batch"), "beyond_L3": True}
            for key, short in (("steady_state", "shared_ref"), ("steady_state_per_instance_refs", "own_refs")):
                r_ = big.get(key)
                if isinstance(r_, dict):
                    # (hbm_frac: the 30 launches' bytes over their time; min / med / max: the launches one by one)
                    b_[short] = {"hbm_frac": _r(r_.get("hbm_frac"), 4), "min": _r(r_.get("hbm_frac_min"), 3), "med": _r(r_.get("hbm_frac_median"), 3),
                                 "max": _r(r_.get("hbm_frac_max"), 3), "gbs": _r(r_.get("hbm_gbs"), 4), "ms": _r(r_.get("ms_per_launch"), 4)}
            line["roofline_hbm"] = b_
            optional.append("roofline_hbm")
        cold = reg.get("cold")
        if isinstance(cold, dict):
            line["cold_regime"] = {"fp64_frac": _r(cold.get("fp64_frac"), 4), "ms": _r(cold.get("ms_per_launch"), 4)}
            optional.append("cold_regime")
    cpu = out.get("cpu_baseline")
    if isinstance(cpu, dict):
        if "error" in cpu:
            line["cpu_baseline"] = {"error": str(cpu["error"])[:200]}
        else:
            line["cpu_baseline"] = {"value": _r(cpu.get("value"), 6), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "physical_cores": cpu.get("physical_cores"),
                                    "kind": cpu.get("kind"),
                                    "sample": str(cpu.get("sample", ""))[:200]}
    cfgs = out.get("configs")
    if isinstance(cfgs, dict):
        line["configs"] = {k: _config_summary(v) for k, v in cfgs.items()}
        mism = [v.get("mismatches") for v in line["configs"].values() if isinstance(v, dict) and "mismatches" in v]
        line["parity"] = {"entries_checked": len(mism), "mismatches": (sum(m for m in mism if isinstance(m, int)) if all(isinstance(m, int) for m in mism)
                                                                       else "see details")}
        optional[:0] = ["configs"]
    if "strong_scaling" in out and isinstance(out["strong_scaling"], dict):
        ss = out["strong_scaling"]
        line["strong_scaling"] = {"total_batch": ss.get("total_batch"), "batch_this_rank": ss.get("batch_this_rank"), "value": _r(ss.get("value"), 7),
                                  "ms_per_step": _r(ss.get("ms_per_step"), 6)}
        optional.append("strong_scaling")
    if "wall_seconds" in out:
        line["wall_seconds"] = out["wall_seconds"]
    if details_path:
        line["details"] = details_path

    def dumps(d):
        return json.dumps(d, allow_nan=False, separators=(",", ":"))
    text = dumps(line)
    while len(text) > COMPACT_LINE_LIMIT and optional:
        line.pop(optional.pop(0), None)
        line["truncated"] = True
        text = dumps(line)
    return text


def write_details(out, path=None):
    """the full record of the run (every regime, every `configs` entry with its notes) -> gpurun_out/bench_details.json; returns
    the path written, or None (an unwritable directory must not cost the run its line)"""
    path = path or os.path.join(ROOT, DETAILS_PATH)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1, default=lambda o: repr(o))
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=TOTAL_BATCH, help="instances PER GPU (weak scaling) / in TOTAL (strong scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--grid-waves-per-cu", type=int, default=int(os.environ.get("TINYMPC_GRID_WAVES_PER_CU", "0")))
    ap.add_argument("--dpp-mode", type=int, default=int(os.environ.get("TINYMPC_DPP_MODE", "2")))
    ap.add_argument("--steps-per-launch", type=int, default=int(os.environ.get("TINYMPC_STEPS_PER_LAUNCH", "0")),
                    help="closed-loop MPC steps fused into one kernel launch (ADMM state stays in registers); "
                         "0 = auto: the largest divisor of --steps that is <= 100; 1 = one launch per step")
    ap.add_argument("--opt", action="append", default=[], help="solver option name=value (experiments)")
    ap.add_argument("--min-seconds", type=float, default=5.0,
                    help="repeat the timed K-step region until this much timed work has run (median reported)")
    ap.add_argument("--max-repeats", type=int, default=5000)
    ap.add_argument("--no-regimes", action="store_true", help="skip the untimed one-launch-per-step replay")
    ap.add_argument("--regimes", action="store_true", help="(kept for old command lines: the replay is always on)")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 3 / 4 / 5-slice (1-GPU runs carry them by default)")
    ap.add_argument("--configs-budget", type=float, default=60.0, help="seconds after which no further config entry is started")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="N > 1, weak scaling: skip the additional strong-scaling measurement")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="seconds of reference solve time per host core (the cpu_baseline leg)")
    ap.add_argument("--configs-cpu-seconds", type=float, default=0.25,
                    help="seconds of reference solve time per checker process and `configs` entry (oracle/config_check.py); 0 = parity sample only")
    ap.add_argument("--configs-cpu-cores", type=int, default=64, help="checker processes of the `configs` leg (a bounded sample of the host's cores)")
    ap.add_argument("--beyond-l3-batch", type=int, default=262144,
                    help="1 GPU: batch of the second warm-regime replay, whose working set is several times the 256 MiB Infinity Cache (0 = skip)")
    ap.add_argument("--details", default=None, help="where the full record goes (default gpurun_out/bench_details.json)")
    ap.add_argument("--dist-timeout", type=float, default=120.0,
                    help="N > 1: seconds a rendezvous / process-group collective / communicator setup may take before the rank gives up")
    ap.add_argument("--run-timeout", type=float, default=900.0,
                    help="N > 1, plain process: seconds the self-spawned ranks get for the measurement itself, on top of --dist-timeout")
    return ap.parse_args(argv)


def error_line(args, world, message, **extra):
    """The ONE JSON line of a run that could not produce a measurement: same envelope (metric, n_gpus, steps, ...), `value` null,
    `error` says why, `preflight` what this process can see of the node (GPU count, RCCL loadable) -- so that a failed N > 1 run
    leaves a record instead of silence."""
    out = {"metric": "QP solves/sec (+ ADMM iters/sec), 64k-batch quadrotor hover", "value": None, "unit": "QP solves/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
           "dtype": "f64", "data": "synthetic", "config": {"workload": "quadrotor_hovering (nx=12, nu=4, N=10) x %d (BASELINE configs[1])" % args.batch},
           "error": message}
    pre = {}
    try:
        import tinympc_amd as tm
        pre["gpus_visible"] = tm.device_count()
        pre["rccl_loadable"] = bool(tm.rccl_available())
    except Exception as e:                               # noqa: BLE001
        pre["library_error"] = repr(e)
    pre["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    out["preflight"] = pre
    out["error"] = str(message)[:1200]
    for k, v in extra.items():                           # launcher_rc, partial_lines ...: bounded, the line must stay parseable
        out[k] = [str(x)[:300] for x in v[:2]] if isinstance(v, (list, tuple)) else v
    return json.dumps(out, allow_nan=False, default=repr)


def spawn_ranks(args):
    """`python bench.py --gpus N` as a PLAIN process: start N ranks of this very script under torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1) and hand rank 0's one JSON line through.  Same code path as the launcher form.  The ranks run
    in a process group of their own under a deadline (--dist-timeout for the rendezvous / communicator setup + --run-timeout for the
    measurement): whatever happens to them -- a rank that dies, a rendezvous or ncclCommInitRank that never returns -- this process
    still prints exactly one JSON line (error_line) and returns non-zero."""
    import signal
    import tinympc_amd as tm
    have = tm.device_count()
    if have < args.gpus and not os.environ.get("TINYMPC_BENCH_SHARE_GPU"):
        print(error_line(args, args.gpus, "bench.py --gpus %d: this node shows %d GPU(s) (TINYMPC_BENCH_SHARE_GPU=1 lets the ranks share devices "
                                           "over gloo: a smoke run of the control flow, not a measurement)" % (args.gpus, have)), flush=True)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), TINYMPC_BENCH_SELF_SPAWNED="1",
               TINYMPC_DIST_TIMEOUT=str(args.dist_timeout))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    deadline = args.dist_timeout + args.run_timeout
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    timed_out = False
    try:
        out, _ = p.communicate(timeout=deadline)
    except subprocess.TimeoutExpired:
        timed_out = True
        for sig in (signal.SIGTERM, signal.SIGKILL):     # the launcher AND its ranks: they share the session started above
            try:
                os.killpg(p.pid, sig)
            except ProcessLookupError:
                break
            try:
                p.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        out = ""
        try:
            out = p.stdout.read() or ""
        except Exception:                                # noqa: BLE001
            pass
    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
    if not timed_out and len(lines) == 1:                # rank 0's line: the result, or its own account of what stopped the job
        print(lines[0], flush=True)
        return p.returncode if p.returncode else (1 if '"error"' in lines[0] else 0)
    why = ("the ranks did not finish within %.0f s (--dist-timeout %.0f + --run-timeout %.0f): killed" % (deadline, args.dist_timeout, args.run_timeout)
           if timed_out else "the ranks ended with rc %s and %d JSON line(s) on stdout" % (p.returncode, len(lines)))
    print(error_line(args, args.gpus, "self-spawned torch.distributed.run: " + why, launcher_rc=p.returncode,
                     launcher_seconds=time.perf_counter() - t0, partial_lines=lines[:2]), flush=True)
    return 1 if (timed_out or p.returncode == 0) else (p.returncode if 0 < p.returncode < 256 else 1)


_PHASE = ["start", 0.0]      # what the rank is doing and since when (the N > 1 watchdog reads it)


def enter_phase(name):
    _PHASE[0], _PHASE[1] = name, time.perf_counter()


class Job:
    """One rank's view of the job: device, process group, barrier, max over ranks."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # TINYMPC_BENCH_SHARE_GPU=1: smoke test of the N > 1 control flow on a box with fewer GPUs than ranks -- the ranks share
        # the devices round-robin and talk over gloo (RCCL refuses two ranks on one device); the line it prints is not a measurement
        self.share_gpu = bool(os.environ.get("TINYMPC_BENCH_SHARE_GPU"))
        if self.share_gpu:
            self.local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(self.local_rank)
        self.dev = f"cuda:{self.local_rank}"
        self.dist = None
        if self.world > 1 or os.environ.get("TINYMPC_FORCE_DIST"):      # FORCE_DIST: exercise the RCCL path on one GPU
            import datetime
            import torch.distributed as dist
            self.dist = dist
            # test hook (tests/test_gpu_sharding.py): TINYMPC_BENCH_FAULT="<rank>:exit" | "<rank>:hang" makes that rank die / stall
            # in front of the rendezvous -- what the deadline handling in spawn_ranks and the collective timeout here are for
            fault = os.environ.get("TINYMPC_BENCH_FAULT", "")
            if fault and int(fault.split(":")[0]) == self.rank:
                if fault.endswith("hang"):
                    time.sleep(1e6)
                os._exit(17)
            to = datetime.timedelta(seconds=float(os.environ.get("TINYMPC_DIST_TIMEOUT", args.dist_timeout)))
            enter_phase("process_group")
            if self.share_gpu:
                os.environ["TINYMPC_EXCHANGE"] = "torch"
                dist.init_process_group("gloo", timeout=to)
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank), timeout=to)   # "nccl" is RCCL on ROCm
        enter_phase("solver_setup")
        self.one = torch.zeros(1, device=self.dev)
        self.stream = torch.cuda.Stream(device=self.local_rank)

    def barrier(self):
        # no rank leaves before every rank has arrived: a one-element all-reduce between two device synchronisations
        # (dist.barrier() itself costs 0.5 ms on this stack, 25x the all-reduce: tools/dist_exchange_cost.py)
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.all_reduce(self.one)
        self.torch.cuda.synchronize()

    def max_over_ranks(self, values):
        if self.dist is None:
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.tolist()

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


class Hover:
    """BASELINE configs[1] on this rank's shard: the solver, its cold start, the timed repetition."""

    def __init__(self, job, args, B, total_batch, exchange=True):
        import numpy as np
        import tinympc_amd as tm
        from tinympc_amd.distributed import StatsExchange
        self.np, self.job, self.args, self.B, self.total = np, job, args, B, total_batch
        prob, extra = tm.load_problem("quadrotor_20hz")
        h = extra["hover"]
        self.nx, self.nu, self.N = prob["nx"], prob["nu"], prob["N"]
        s = self.s = tm.TinyBatchSolver.from_problem(prob, B, device=job.local_rank)
        nx, nu, N = self.nx, self.nu, self.N
        s.set_bound_constraints(np.full((nx, 1), h["x_min"]), np.full((nx, 1), h["x_max"]),
                                np.full((nu, 1), h["u_min"]), np.full((nu, 1), h["u_max"]))
        s.update_settings(max_iter=h["max_iter"])
        s.set_option("advance_x0", 1)
        s.set_option("grid_waves_per_cu", args.grid_waves_per_cu)
        s.set_option("dpp_mode", args.dpp_mode)
        for kv in args.opt:                              # experiments: --opt grid_waves_per_cu=8
            k, v = kv.split("=")
            s.set_option(k, int(v))
        self.T = steps_per_launch(args.steps, args.warmup, args.steps_per_launch)
        if self.T is None:
            sys.exit("--steps must be a multiple of --steps-per-launch")
        self.Tw = steps_per_launch(args.warmup) if args.warmup > 0 else 1      # the untimed warm-up steps, fused the same way on their own
        self.launches = args.steps // self.T
        s.set_stream(job.stream.cuda_stream)             # kernels, events and the RCCL collective share one stream
        self.xref = np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N))
        self.x0 = np.array(h["x0"], dtype=np.float64)
        self.stats = job.torch.zeros(10, dtype=job.torch.float64, device=job.dev)
        self.exchange = None
        if job.dist is not None and exchange:
            enter_phase("communicator")
            self.exchange = StatsExchange(s, job.dist, job.local_rank, total_batch=total_batch)
            enter_phase("solver_setup")

    def cold_start(self):
        self.s.reset()
        self.s.set_x_ref(self.xref, broadcast=True)
        self.s.set_x0(self.x0, broadcast=True)

    def timed_repetition(self):
        """cold start (untimed) -> barrier -> EXACTLY --steps MPC steps + the statistics exchange -> barrier."""
        s, job = self.s, self.job
        self.cold_start()
        s.set_option("timing", self.launches)            # HIP events around every timed solve kernel, on the solver's stream
        job.barrier()
        t0 = time.perf_counter()
        for _ in range(self.launches):
            s.solve_async()
        if job.dist is not None:                         # the one exchange of the path: a 64-byte message per rank, RCCL over xGMI
            st = self.exchange()
        else:
            s.reduce_stats_async(self.stats.data_ptr())
        job.barrier()
        elapsed = time.perf_counter() - t0
        if job.dist is None:
            st = self.stats.to("cpu")
        return elapsed, st.tolist(), s.timing_ms()

    def measure(self, min_seconds, warmup=True):
        """-> dict(elapsed median, rep_s, kernel ms arrays, job-wide statistics of one repetition)"""
        np, s, job, args = self.np, self.s, self.job, self.args
        with job.torch.cuda.stream(job.stream):
            if warmup:
                self.cold_start()
                s.set_option("steps_per_launch", self.Tw)
                for _ in range(args.warmup // self.Tw):
                    s.solve_async()
                s.synchronize()
            s.set_option("steps_per_launch", self.T)
            if job.dist is not None:                     # first-use costs of the collective stay out of the timed region
                self.exchange()
                job.barrier()
            e0, st, km = self.timed_repetition()
            e0 = job.max_over_ranks([e0])[0]
            repeats = int(min(max(3, -(-min_seconds // max(e0, 1e-6))), args.max_repeats))
            rep_s, rep_kernel_ms = [e0], [km]
            for _ in range(repeats - 1):
                e, st, km = self.timed_repetition()
                rep_s.append(e)
                rep_kernel_ms.append(km)
        rep_s = np.array(job.max_over_ranks(rep_s))      # every repetition: the slowest rank's clock
        return dict(elapsed=float(np.median(rep_s)), rep_s=rep_s, repeats=repeats, st=st,
                    kern_ms=np.concatenate(rep_kernel_ms), kern_first=np.array([k[0] for k in rep_kernel_ms]),
                    kern_sum_rep=float(np.median([k.sum() for k in rep_kernel_ms])))

    def close(self):
        if self.exchange is not None:
            self.exchange.close()
        self.s.close()


def regimes_replay(hv, fl, bytes_warm, full=True, reps=5):
    """Untimed replay (rank 0): the same episode with ONE launch per MPC step, so that the two regimes SURVEY.md 8(d) asks
    for are visible in every line -- cold steps (100 ADMM iterations each, FP64 bound) and steady state (1-2 iterations:
    every launch loads and stores the records, the real HBM roofline of this path).  Median over five replays."""
    np, s, job, B = hv.np, hv.s, hv.job, hv.B
    nx, nu, N = hv.nx, hv.nu, hv.N
    S = nx * N + nu * (N - 1)

    def replay(store_primal, share_ref=1):
        runs = []
        s.set_option("store_primal", store_primal)
        s.set_option("share_ref", share_ref)
        for _ in range(reps):
            hv.cold_start()
            s.set_option("timing", 100)
            for _ in range(100):
                s.solve_async()
            s.synchronize()
            runs.append(s.timing_ms())
        s.set_option("store_primal", 1)
        s.set_option("share_ref", 1)
        return np.median(np.array(runs), axis=0)

    with job.torch.cuda.stream(job.stream):
        s.set_option("steps_per_launch", 1)
        ms = replay(1)
        ms_own = replay(1, share_ref=0)
        ms_lean = replay(0) if full else None
        ms_u0 = replay(2) if full else None
        # the iteration count of every step of the episode (all instances are identical): which launches skip the v|z store
        hv.cold_start()
        iters = []
        for _ in range(100):
            s.solve_async()
            s.synchronize()
            iters.append(int(s.status()["iter"][0]))
        s.set_option("steps_per_launch", hv.T)
    iters = np.array(iters)
    warm_steps = slice(70, 100)

    def steady(msv, ref_shared, store_primal, note):
        t = float(msv[warm_steps].mean()) * 1e-3
        per_step = np.array([moved_bytes_per_solve(bytes_warm, S, nx, nu, it, ref_shared, store_primal) for it in iters[warm_steps]])
        moved = float(per_step.mean())
        fr = per_step * B / (msv[warm_steps] * 1e-3) / 1e9 / HBM_PEAK_GBS          # every launch of the 30 on its own bytes and its own time
        return {"steps": "70-99", "ms_per_launch": t * 1e3, "ms_per_launch_min": float(msv[warm_steps].min()),
                "ms_per_launch_median": float(np.median(msv[warm_steps])), "ms_per_launch_max": float(msv[warm_steps].max()),
                "hbm_frac_min": float(fr.min()), "hbm_frac_median": float(np.median(fr)), "hbm_frac_max": float(fr.max()),
                "admm_iters_per_solve": float(iters[warm_steps].mean()),
                "bytes_moved_per_solve": moved, "algorithmic_bytes_per_solve": bytes_warm,
                "hbm_gbs": moved * B / t / 1e9, "hbm_frac": moved * B / t / 1e9 / HBM_PEAK_GBS,
                "hbm_frac_formula": bytes_warm * B / t / 1e9 / HBM_PEAK_GBS, "note": note}

    cold = float(ms[:5].mean()) * 1e-3
    out = {
        "batch": B,
        "cold": {"steps": "0-4", "admm_iters_per_solve": float(iters[:5].mean()), "ms_per_launch": cold * 1e3,
                 "fp64_tflops": float(iters[:5].mean()) * B * fl / cold / 1e12,
                 "fp64_frac": float(iters[:5].mean()) * B * fl / cold / 1e12 / FP64_PEAK_TFLOPS},
        "steady_state": steady(ms, True, 1,
                               "one launch per MPC step, the default form.  hbm_frac counts the bytes this form moves: bytes_warm = "
                               "8(nx+8S)+44 (SURVEY.md 8(d)) minus the Xref|Uref read (config 2's instances share ONE L2-resident "
                               "reference record) and minus the v|z store of a solve that converges at its first check (admm.cpp:431-441 "
                               "returns before v = vnew); hbm_frac_formula divides the full bytes_warm by the same time"),
        "steady_state_per_instance_refs": steady(ms_own, False, 1,
                                                 "option share_ref = 0: every instance reads its OWN reference record, what any "
                                                 "non-identical batch needs -- the general warm-step figure of this path"),
        "iters_per_step": iters.tolist(),
    }
    if full:
        out["steady_state_no_primal_store"] = steady(ms_lean, True, 0,
                                                     "option store_primal = 0: x|u is not written back either (no consumer between steps when "
                                                     "the plant step runs on the device; solution->x|u = vnew|znew is still stored)")
        out["steady_state_first_knot_store"] = steady(ms_u0, True, 2,
                                                      "option store_primal = 2: of x|u only x[:,0], x[:,1], u[:,0] are written (the control a "
                                                      "closed-loop caller applies)")
    return out


def beyond_l3_replay(job, args, fl, batch):
    """The warm regime where the records really come from HBM: the same one-launch-per-step replay over a batch
    whose working set is several times the 256 MiB Infinity Cache (262 144 instances x ~3.7 KB of records per warm solve ~ 1 GB),
    shared and per-instance reference records."""
    try:
        hb = Hover(job, args, batch, batch, exchange=False)
        with job.torch.cuda.stream(job.stream):          # one untimed episode: first-use costs of this batch size
            hb.cold_start()
            hb.s.set_option("steps_per_launch", hb.Tw)
            hb.s.solve_async()
            hb.s.synchronize()
        r = regimes_replay(hb, fl, hb.s.algorithmic_bytes(cold=False), full=False, reps=3)
        hb.close()
        r.pop("iters_per_step", None)
        r["working_set_bytes"] = int(batch * r["steady_state_per_instance_refs"]["bytes_moved_per_solve"])
        return r
    except Exception as e:                               # noqa: BLE001
        return {"error": repr(e)}


def cpu_baseline_subprocess(seconds, steps):
    """The real reference on every host core (oracle/cpu_baseline.py), in processes of its own AFTER the GPU legs: the GPU
    phase then sits at the START of the run (the driver's sparse gpu_busy sampler sees it) and no HIP context is forked."""
    code = ("import sys, json; sys.path.insert(0, %r); import cpu_baseline; "
            "print('@@CPU@@' + json.dumps(cpu_baseline.run(seconds=%r, steps=%r)))" % (os.path.join(ROOT, "oracle"), seconds, steps))
    try:
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=max(120.0, 20 * seconds))
        for line in p.stdout.splitlines():
            if line.startswith("@@CPU@@"):
                return json.loads(line[7:])
        return {"error": (p.stderr or p.stdout)[-400:]}
    except Exception as e:                               # noqa: BLE001
        return {"error": repr(e)}


def config_check_subprocess(spec_path, seconds, cores=None):
    code = ("import sys, json; sys.path.insert(0, %r); import config_check; "
            "print('@@CHK@@' + json.dumps(config_check.run(%r, seconds=%r, cores=%r)))" % (os.path.join(ROOT, "oracle"), spec_path, seconds, cores))
    try:
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=max(300.0, 200 * seconds))
        for line in p.stdout.splitlines():
            if line.startswith("@@CHK@@"):
                return json.loads(line[7:])
        return {"error": (p.stderr or p.stdout)[-600:]}
    except Exception as e:                               # noqa: BLE001
        return {"error": repr(e)}


def main():
    args = parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only supports dmabuf IPC (RCCL across ranks)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    # stdout carries exactly ONE line, the result JSON: libraries that chat on fd 1 (RCCL prints a version banner at
    # communicator creation) are pointed at stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    env_rank, env_world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # A rank of an N > 1 job must never end in silence: whatever stops it -- an exception out of a collective that timed out
    # (--dist-timeout), a setup phase that never returns (watchdog below: ncclCommInitRank has no timeout of its own) -- rank 0
    # still writes ONE JSON line with "error" before it leaves.
    setup_done = _PHASE

    def give_up(message):
        if env_rank == 0:
            os.write(result_fd, (error_line(args, env_world, message) + "\n").encode())
        os._exit(3)

    if env_world > 1:
        import threading

        def watchdog():
            # only the phases that can hang without a timeout of their own are on the clock (rendezvous / init_process_group,
            # ncclCommInitRank inside StatsExchange): a slow cold start of the solver itself (library load, first-use hipRTC) is not
            while _PHASE[0] != "finished":               # (for the whole run: the strong-scaling leg sets up a communicator of its own)
                time.sleep(1.0)
                if _PHASE[0] in ("process_group", "communicator") and time.perf_counter() - _PHASE[1] > args.dist_timeout:
                    give_up("rank %d: %s setup did not finish within --dist-timeout %.0f s" % (env_rank, _PHASE[0], args.dist_timeout))
        threading.Thread(target=watchdog, daemon=True).start()
    try:
        run(args, result_fd, setup_done)
    except SystemExit:
        raise
    except BaseException as e:                           # noqa: BLE001
        import traceback
        traceback.print_exc()
        give_up("rank %d: %r" % (env_rank, e))


def run(args, result_fd, setup_done):
    t_start = time.perf_counter()

    import numpy as np
    import torch
    import tinympc_amd as tm

    if not torch.cuda.is_available() or tm.device_count() == 0:
        sys.exit("bench.py needs an MI355X: tinympc_amd has no CPU fallback")
    job = Job(args)
    rank, world = job.rank, job.world
    args.gpus = world
    strong = args.scaling == "strong"
    B = shard_size(args.batch, rank, world) if strong else args.batch
    total = args.batch if strong else world * args.batch
    hv = Hover(job, args, B, total)
    if job.dist is not None:                             # the communicator has carried one exchange: setup is over
        with job.torch.cuda.stream(job.stream):
            hv.exchange()
            job.barrier()
    enter_phase("measuring")
    nx, nu, N, T, launches = hv.nx, hv.nu, hv.N, hv.T, hv.launches
    m = hv.measure(args.min_seconds)
    elapsed, rep_s, st = m["elapsed"], m["rep_s"], m["st"]
    acc_iters, acc_solved, max_resid = st[7], st[8], st[3:7]        # job-wide, one repetition (each starts with a reset)
    kern_ms, kern_sum_rep = m["kern_ms"], m["kern_sum_rep"]
    fl = flops_per_iter(nx, nu, N)
    bytes_warm = hv.s.algorithmic_bytes(cold=False)

    regimes = None
    if rank == 0 and not args.no_regimes:
        regimes = regimes_replay(hv, fl, bytes_warm)
        regimes["traffic_note"] = "roofline.traffic is a citation of a committed rocprofv3 PMC run (profiles/traffic.json), not of this run"
    exchange_kind = hv.exchange.kind if hv.exchange is not None else "none (one rank)"
    # ranks of the communicator the exchange really ran on: asked of RCCL (ncclCommCount) on the native path, of the process group
    # when the exchange went through torch.distributed (`stats_exchange` says which)
    if hv.exchange is None:
        rccl_ranks = 1
    elif hv.exchange.kind == "native":
        rccl_ranks = hv.exchange.comm_ranks
    else:
        rccl_ranks = job.dist.get_world_size()
    hv.close()
    if regimes is not None and world == 1 and args.beyond_l3_batch > 0:
        regimes["beyond_l3"] = beyond_l3_replay(job, args, fl, args.beyond_l3_batch)

    # N > 1 under weak scaling: the strong-scaling form of the metric as well (65 536 instances in TOTAL, sharded)
    strong_extra = None
    if world > 1 and not strong and not args.no_strong:
        hs = Hover(job, args, shard_size(args.batch, rank, world), args.batch)
        ms_ = hs.measure(min(args.min_seconds, 2.0))
        sst = ms_["st"]
        strong_extra = {"total_batch": args.batch, "batch_this_rank": hs.B, "value": args.batch * args.steps / ms_["elapsed"], "unit": "QP solves/s",
                        "ms_per_step": ms_["elapsed"] / args.steps * 1e3, "admm_iters_per_s": sst[7] / ms_["elapsed"], "repeats": ms_["repeats"],
                        "kernel_ms_sum_per_repetition": ms_["kern_sum_rep"],
                        "note": "the same timed region with 65 536 instances in TOTAL, sharded over the ranks (BASELINE: '64k-batch ... 1/2/4/8 GPU')"}
        hs.close()

    configs, spec_path = None, None
    if rank == 0 and world == 1 and not args.no_configs:
        import tempfile
        import bench_configs
        spec_path = os.path.join(tempfile.mkdtemp(prefix="tinympc_bench_"), "config_samples.pkl")
        configs = bench_configs.run_all(device=job.local_rank, budget_s=args.configs_budget,
                                        log=lambda msg: print(msg, file=sys.stderr, flush=True), spec_out=spec_path)
    gpu_phase_s = time.perf_counter() - t_start

    # CPU baseline LAST (rank 0, N = 1 only): the real reference on every host core, the SAME workload as the timed region
    cpu = None
    cpu_phase_s = check_phase_s = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        t_cpu = time.perf_counter()
        cpu = cpu_baseline_subprocess(args.cpu_seconds, args.steps)
        cpu_phase_s = time.perf_counter() - t_cpu
        t_cpu = time.perf_counter()
        # ... and the checker of the `configs` entries: the oracle on a sample of each entry's own records (`parity_sample`), the
        # real reference timed on the same records (`cpu_baseline` of the entry) -- oracle/config_check.py, processes of its own
        if configs is not None and spec_path and os.path.exists(spec_path):
            chk = config_check_subprocess(spec_path, args.configs_cpu_seconds, args.configs_cpu_cores)
            for name, e in configs.items():
                if isinstance(chk.get(name), dict) and "error" not in e and "skipped" not in e:
                    e.update(chk[name])
                elif "error" in chk and "error" not in e and "skipped" not in e:
                    e["parity_sample"] = {"error": chk["error"]}
            check_phase_s = time.perf_counter() - t_cpu

    solves = float(total) * args.steps
    value = solves / elapsed
    avg_kernel_s = float(kern_ms.mean()) * 1e-3
    # real HBM bytes of one launch: the records are loaded and stored once however many MPC steps it fuses
    hbm_gbs = bytes_warm * B / avg_kernel_s / 1e9
    iters_local = acc_iters * (B / float(total))     # one repetition, this GPU (identical instances)
    fp64_tflops = iters_local * fl / (kern_sum_rep * 1e-3) / 1e12
    traffic, traffic_source = traffic_per_launch(T, B)
    common = {"traffic": traffic, "traffic_source": traffic_source, "kernel": "admm_solve_kernel<12,4,10>", "avg_launch_ms": avg_kernel_s * 1e3,
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
        per = "%d identical instances %s" % (args.batch, "in total, sharded" if strong else "per GPU")
        out = {
            "metric": "QP solves/sec (+ ADMM iters/sec), 64k-batch quadrotor hover",
            "value": value, "unit": "QP solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "quadrotor_hovering (nx=12, nu=4, N=10), %s, closed-loop MPC steps from a cold start "
                                   "(BASELINE configs[1])" % per,
                       "batch_per_gpu": B, "total_batch": total, "parallelism": f"batch-sharded x{world}",
                       "grid_waves_per_cu": args.grid_waves_per_cu, "dpp_mode": args.dpp_mode,
                       "mpc_steps_per_launch": T, "stats_exchange": exchange_kind,
                       "launcher": ("self-spawned torch.distributed.run" if os.environ.get("TINYMPC_BENCH_SELF_SPAWNED") else
                                    ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "plain process"))},
            "rccl_ranks": rccl_ranks,
            "timed_region": {"repeats": int(m["repeats"]), "seconds_total": float(rep_s.sum()),
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
            "kernel_ms": {"first_launch_median": float(np.median(m["kern_first"])), "sum_per_repetition_median": kern_sum_rep,
                          "min": float(kern_ms.min()), "max": float(kern_ms.max()), "count": int(kern_ms.size)},
            "gpu_phase_seconds": gpu_phase_s, "cpu_baseline_phase_seconds": cpu_phase_s, "configs_checker_phase_seconds": check_phase_s,
        }
        if job.share_gpu:
            out["data"] = "synthetic; SMOKE RUN: %d ranks share %d GPU(s) over gloo -- not a measurement" % (world, torch.cuda.device_count())
        if strong_extra is not None:
            out["strong_scaling"] = strong_extra
        if regimes is not None:
            out["regimes"] = regimes
        if configs is not None:
            out["configs"] = configs
        if cpu is not None:
            out["cpu_baseline"] = cpu
        out["wall_seconds"] = _r(time.perf_counter() - t_start, 4)
        details = write_details(out, args.details)
        sys.stdout.flush()
        os.write(result_fd, (compact_line(out, details) + "\n").encode())
    enter_phase("finished")
    job.close()


if __name__ == "__main__":
    main()
