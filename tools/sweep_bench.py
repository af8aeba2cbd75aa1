#!/usr/bin/env python3
"""BASELINE configs[4]: the random (nx, nu, N) sweep -- one cold batched solve per cell, per-instance random
x0 / Xref, max_iter 500, u in [-0.5, 0.5].  Prints a markdown roofline table and writes JSON.

    python tools/sweep_bench.py --batch 131072 --out profiles/r01_sweep.json
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
           tools/sweep_bench.py --batch 1048576 --out sweep_8gpu.json        # BASELINE: batch 1M over 8 x MI355X

Under torch.distributed.run --batch is the TOTAL batch of a cell: it is sharded round-robin by instance index over the
ranks (every rank draws the same seeded inputs and keeps its own instances; iteration counts diverge), each
cell's timed region is one launch per GPU closed by the one 64-byte statistics exchange (RCCL), the slowest rank's wall
clock counts.  On one GPU the kernel time of the launch (HIP events) is reported instead.

Which kernel serves a cell is reported: "regs" = one-row register-resident kernel (admm_kernel.hip.h), "tile" =
W x R-row tile kernel (tile_kernel.hip.h), "cover" = coverage kernel (general_kernel.hip.h).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm  # noqa: E402
from tinympc_amd.distributed import shard_indices  # noqa: E402

PARITY = None                                         # --parity: list of samples (one per cell)
UNIFORM = 0                                           # --uniform K: see run_cell
RANK = int(os.environ.get("RANK", "0"))
LOCAL_RANK = int(os.environ.get("LOCAL_RANK", "0"))
WORLD = int(os.environ.get("WORLD_SIZE", "1"))
_dist = None


def run_cell_sharded(nx, nu, N, B, reps):
    """one cell on WORLD GPUs: total batch B, this rank's round-robin shard, wall clock of the slowest rank"""
    import time
    import torch
    from tinympc_amd.distributed import StatsExchange
    prob, rng = tm.random_problem(nx, nu, N)
    idx = np.array(shard_indices(B, RANK, WORLD, interleaved=True))
    s = tm.TinyBatchSolver.from_problem(prob, len(idx), device=LOCAL_RANK)
    for kv in filter(None, os.environ.get("TINYMPC_OPTS", "").split(",")):
        s.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=500)
    x0 = rng.uniform(-1, 1, (B, nx))[idx]                       # the same draws as the unsharded cell, this rank's rows
    xr = np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1))[idx], N, axis=2)
    exchange = StatsExchange(s, _dist, LOCAL_RANK, total_batch=B)
    one = torch.zeros(1, device=f"cuda:{LOCAL_RANK}")

    def barrier():
        s.synchronize()
        _dist.all_reduce(one)
        torch.cuda.synchronize()
    best, st = None, None
    for _ in range(reps + 1):
        s.reset()
        s.set_x0(x0)
        s.set_x_ref(xr)
        barrier()
        t0 = time.perf_counter()
        s.solve_async()
        st = exchange().numpy()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=f"cuda:{LOCAL_RANK}")
        _dist.all_reduce(t, op=_dist.ReduceOp.MAX)
        best = float(t.item()) if best is None else min(best, float(t.item()))
    path = s.kernel_path()
    alg = s.algorithmic_bytes()
    exchange.close()
    s.close()
    fl = tm.flops_per_iter(nx, nu, N)
    iters, solved = st[0], st[1]
    return dict(nx=nx, nu=nu, N=N, batch=B, n_gpus=WORLD, kernel=path, ms=best * 1e3, solves_per_s=B / best, iters_per_s=iters / best,
                iters_per_solve=iters / B, solved_fraction=solved / B, fp64_tflops=iters * fl / best / 1e12,
                fp64_frac=iters * fl / best / 78.6e12 / WORLD, hbm_gbs=alg * B / best / 1e9, hbm_frac=alg * B / best / 8e12 / WORLD,
                bytes_per_solve=alg, flops_per_iter=fl)


def run_cell(nx, nu, N, B, reps):
    if _dist is not None:
        return run_cell_sharded(nx, nu, N, B, reps)
    prob, rng = tm.random_problem(nx, nu, N)
    s = tm.TinyBatchSolver.from_problem(prob, B)
    for kv in filter(None, os.environ.get("TINYMPC_OPTS", "").split(",")):       # experiments: prefer_tile=1 ...
        s.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    s.set_bound_constraints(np.full((nx, 1), -1e17), np.full((nx, 1), 1e17), np.full((nu, 1), -0.5), np.full((nu, 1), 0.5))
    s.update_settings(max_iter=500)
    x0 = rng.uniform(-1, 1, (B, nx))
    xr = np.repeat(rng.uniform(-0.2, 0.2, (B, nx, 1)), N, axis=2)
    if UNIFORM > 0:
        # --uniform K: the instruction path of ONE iteration, measured -- tolerances 0 (no solve ever passes its test: every instance runs
        # exactly K iterations, every row of every wave is busy, no lock step, no split, no tails) -- for tools/sweep_counters.py /
        # sweep_ceiling.py under rocprofv3 --pmc SQ_INSTS_VALU: instructions per wave-iteration = the kernel's real loop length
        s.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=UNIFORM)
        s.set_option("repack_after", 0)
        s.set_option("plan", 0)
    best = None
    s.set_x0(x0)                                      # inputs resident before the timed solves (reset() keeps x0 and the references):
    s.set_x_ref(xr)                                   # a 200 MB upload in front of every solve lets the GPU clock down first
    for _ in range(reps + 1):
        s.reset()
        s.set_option("timing", 1)
        s.solve_async()
        ms = float(s.timing_ms()[0])
        best = ms if best is None else min(best, ms)
    path = s.kernel_path()
    st = s.reduce_stats()
    iters, solved = st[0], st[1]
    alg = s.algorithmic_bytes()
    auto_k, auto_pm = s.get_option("auto_split_k"), s.get_option("auto_split_permille")
    stt = s.status()
    hv, hc = np.unique(stt["iter"], return_counts=True)
    if PARITY is not None:                            # --parity: a sample of THIS batch for oracle/config_check.py (run after the GPU work)
        idx = np.unique(np.linspace(0, B - 1, 256).astype(np.int64))
        it = np.where(stt["solved"][idx] != 0, stt["iter"][idx], -stt["iter"][idx])
        PARITY.append(dict(name="sweep_%d_%d_%d" % (nx, nu, N), kind="single", problem={k: v for k, v in prob.items()},
                           cfg_kw=dict(max_iter=500, u_min=np.full((nu, 1), -0.5), u_max=np.full((nu, 1), 0.5)),
                           x0=x0[idx], Xref=xr[idx], Uref=np.zeros((nu, N - 1)), gpu_iter=it.astype(np.int32), gpu_u0=s.get("u")[idx][:, :, 0]))
    s.close()
    fl = tm.flops_per_iter(nx, nu, N)
    t = best * 1e-3
    return dict(nx=nx, nu=nu, N=N, batch=B, kernel=path,
                ms=best, solves_per_s=B / t, iters_per_s=iters / t, iters_per_solve=iters / B, solved_fraction=solved / B,
                fp64_tflops=iters * fl / t / 1e12, fp64_frac=iters * fl / t / 78.6e12,
                hbm_gbs=alg * B / t / 1e9, hbm_frac=alg * B / t / 8e12, bytes_per_solve=alg, flops_per_iter=fl,
                auto_split_k=auto_k, auto_split_predicted=auto_pm / 1000.0, iter_histogram={int(v): int(c) for v, c in zip(hv, hc)})


def hetero_parity(args):
    """full-batch evidence for the round-5 per-instance-data forms of the tile kernel (tiny_api.cpp:307-381 once per
    INSTANCE).  tools/bench_configs.py hetero_cell builds the batch (A x (1 + N(0, 1e-3)), B x (1 + N(0, 0.05)), rho x U(0.8, 1.2) per
    instance) and the sample; oracle/config_check.py solves every sampled instance under its own problem data."""
    import pickle
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench_configs as bc
    cells = ([tuple(int(v) for v in c.split(",")) for c in args.cells.split(";")] if args.cells else
             [(nx, nu, N) for nx in (4, 8, 12, 20) for nu in (2, 4, 8) for N in (10, 30, 50) if nx + nu > 16 or N == 50])
    rows, specs = [], []
    print("| nx | nu | N | kernel | ms | it/solve | solved | FP64 frac |")
    print("|---|---|---|---|---|---|---|---|")
    for nx, nu, N in cells:
        try:
            e, spec = bc.hetero_cell(nx, nu, N, B=args.batch)
        except Exception as ex:                        # noqa: BLE001
            rows.append(dict(nx=nx, nu=nu, N=N, error=repr(ex)))
            print("| %d | %d | %d | error: %s |" % (nx, nu, N, repr(ex)[:120]), flush=True)
            continue
        rows.append(dict(nx=nx, nu=nu, N=N, kernel=e["kernel"], ms=e["ms"], iters_per_solve=e["iters_per_solve"], solved_fraction=e.get("solved_fraction"),
                         fp64_frac=e["roofline"]["frac"], name=spec["name"]))
        specs.append(spec)
        print("| %d | %d | %d | %s | %.3f | %.1f | %.3f | %.3f |" % (nx, nu, N, e["kernel"], e["ms"], e["iters_per_solve"], e.get("solved_fraction") or 0.0, e["roofline"]["frac"]), flush=True)
    spec_path = os.path.join(tempfile.mkdtemp(prefix="tinympc_sweep_het_"), "samples.pkl")
    pickle.dump(specs, open(spec_path, "wb"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json; sys.path.insert(0, %r); import config_check; print('@@CHK@@' + json.dumps(config_check.run(%r, seconds=0.02)))"
            % (os.path.join(root, "oracle"), spec_path))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=3000)
    chk = [json.loads(ln[7:]) for ln in p.stdout.splitlines() if ln.startswith("@@CHK@@")]
    with open(args.hetero, "w") as f:
        f.write("The tile kernel's per-instance-data form (round 5) on every config-5 cell the tile kernel serves, at the FULL batch (%d instances, every one its own A, B,\n"
                "rho and cache -- tiny_api.cpp:307-381 per instance, computed by the batched Riccati kernel --, one cold solve, max_iter 500): 256 instances evenly spaced\n"
                "through the batch, each solved again by ITS OWN oracle (oracle/liboracle.so: tiny_setup with that instance's data); iteration counts and solved flags must be\n"
                "equal, u[:,0] relative to the solve's largest entry.  (tools/sweep_bench.py --hetero)\n\n" % args.batch)
        f.write("| cell | kernel | ms | FP64 frac | instances | iteration sum GPU | oracle | count mismatches | max rel err u0 |\n|---|---|---|---|---|---|---|---|---|\n")
        bad = 0
        for r in rows:
            if "error" in r:
                f.write("| (%d,%d,%d) | launch failed: %s |\n" % (r["nx"], r["nu"], r["N"], r["error"][:200]))
                bad += 1
                continue
            ps = chk[0][r["name"]]["parity_sample"] if chk and r["name"] in chk[0] else None
            if ps is None:
                f.write("| (%d,%d,%d) | %s | checker failed: %s |\n" % (r["nx"], r["nu"], r["N"], r["kernel"], (p.stderr or "")[-200:].replace("\n", " ")))
                bad += 1
                continue
            bad += ps["iteration_count_mismatches"]
            f.write("| (%d,%d,%d) | %s | %.3f | %.3f | %d | %d | %d | %d | %.1e |\n" % (r["nx"], r["nu"], r["N"], r["kernel"], r["ms"], r["fp64_frac"], ps["instances"],
                                                                                      ps["iter_sum_gpu"], ps["iter_sum_oracle"], ps["iteration_count_mismatches"], ps["max_rel_err_u0"]))
        f.write("\ntotal iteration-count mismatches: %d\n" % bad)
    print("hetero parity table ->", args.hetero)
    print("@@JIT@@" + json.dumps(tm.jit_used()))          # (the run-time instantiated forms these launches took: csrc/jit_prebuilt.txt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=131072)
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--cells", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--parity", default="", help="(1 GPU) also check 256 instances of every cell's batch against the oracle, iteration counts "
                                                 "and u[:,0] (oracle/config_check.py, processes of its own AFTER the GPU work); markdown table to this file")
    ap.add_argument("--hetero", default="", help="(1 GPU, with --parity's checker) the tile kernel's per-instance-data form at every cell's FULL batch: every instance "
                                                 "its own A, B, rho (its own cache from the batched Riccati kernel), one cold solve; 256 instances evenly spaced "
                                                 "through the batch against EACH ONE'S OWN oracle (tiny_setup per instance); markdown table to this file")
    ap.add_argument("--uniform", type=int, default=0, help="K > 0: every instance runs exactly K iterations (tolerances 0): the per-iteration instruction path, for counters")
    args = ap.parse_args()
    global UNIFORM
    UNIFORM = args.uniform
    if args.hetero and WORLD == 1:
        return hetero_parity(args)
    global PARITY
    if args.parity and WORLD == 1:
        PARITY = []
    global _dist, LOCAL_RANK
    if WORLD > 1 or os.environ.get("TINYMPC_FORCE_DIST"):
        from tinympc_amd.distributed import init_process_group
        _dist, LOCAL_RANK = init_process_group(LOCAL_RANK)        # (LOCAL_RANK becomes the device index: smoke mode shares GPUs)
    say = print if RANK == 0 else (lambda *a, **k: None)
    cells = ([tuple(int(v) for v in c.split(",")) for c in args.cells.split(";")] if args.cells else
             [(nx, nu, N) for nx in (4, 8, 12, 20) for nu in (2, 4, 8) for N in (10, 30, 50)])
    rows = []
    say("| nx | nu | N | kernel | ms | solves/s | ADMM it/s | it/solve | solved | FP64 frac | HBM frac |")
    say("|---|---|---|---|---|---|---|---|---|---|---|")
    for nx, nu, N in cells:
        r = run_cell(nx, nu, N, args.batch, args.reps)
        rows.append(r)
        say(f"| {nx} | {nu} | {N} | {r['kernel']} | {r['ms']:.3f} | {r['solves_per_s']:.3e} | {r['iters_per_s']:.3e} | "
              f"{r['iters_per_solve']:.1f} | {r['solved_fraction']:.3f} | {r['fp64_frac']:.3f} | {r['hbm_frac']:.4f} |", flush=True)
    if args.out and RANK == 0:
        json.dump(rows, open(args.out, "w"), indent=1)
    if PARITY:
        import pickle, subprocess, tempfile
        spec = os.path.join(tempfile.mkdtemp(prefix="tinympc_sweep_"), "samples.pkl")
        pickle.dump(PARITY, open(spec, "wb"))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        code = ("import sys, json; sys.path.insert(0, %r); import config_check; print('@@CHK@@' + json.dumps(config_check.run(%r, seconds=0.05)))"
                % (os.path.join(root, "oracle"), spec))
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1800)
        chk = [json.loads(ln[7:]) for ln in p.stdout.splitlines() if ln.startswith("@@CHK@@")]
        with open(args.parity, "w") as f:
            f.write("Every cell of the config-5 sweep at its FULL batch (%d instances, one cold solve): 256 instances evenly spaced through the batch, solved again by\n"
                    "the oracle (oracle/liboracle.so) -- iteration counts and solved flags must be equal, u[:,0] relative to the solve's largest entry.\n\n" % args.batch)
            f.write("| cell | kernel | instances | iteration sum GPU | oracle | count mismatches | max rel err u0 |\n|---|---|---|---|---|---|---|\n")
            bad = 0
            for r in rows:
                name = "sweep_%d_%d_%d" % (r["nx"], r["nu"], r["N"])
                ps = chk[0][name]["parity_sample"] if chk and name in chk[0] else None
                if ps is None:
                    f.write("| (%d,%d,%d) | %s | checker failed: %s |\n" % (r["nx"], r["nu"], r["N"], r["kernel"], (p.stderr or "")[-200:].replace("\n", " ")))
                    bad += 1
                    continue
                bad += ps["iteration_count_mismatches"]
                f.write("| (%d,%d,%d) | %s | %d | %d | %d | %d | %.1e |\n" % (r["nx"], r["nu"], r["N"], r["kernel"], ps["instances"], ps["iter_sum_gpu"],
                                                                             ps["iter_sum_oracle"], ps["iteration_count_mismatches"], ps["max_rel_err_u0"]))
            f.write("\ntotal iteration-count mismatches: %d\n" % bad)
        say("parity table ->", args.parity)
    if _dist is not None:
        _dist.destroy_process_group()


if __name__ == "__main__":
    main()
