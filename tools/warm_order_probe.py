#!/usr/bin/env python3
"""Warm regime (one launch per MPC step, 1-2 ADMM iterations per solve) of the hover workload under the launch-order option:
successive launches walk the batch in the same direction (launch_order 0) or in alternating directions (1), so that a launch begins
with the records the one before it touched last (the 256 MiB Infinity Cache).  Per form: median over replays of the per-step kernel
time (HIP events), steps 70-99, and the bytes the form moves per solve."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinympc_amd as tm

prob, extra = tm.load_problem("quadrotor_20hz")
h = extra["hover"]
nx, nu, N = prob["nx"], prob["nu"], prob["N"]
S = nx * N + nu * (N - 1)
for B in [int(b) for b in os.environ.get("BATCHES", "65536").split(",")]:
    s = tm.TinyBatchSolver.from_problem(prob, B)
    s.set_bound_constraints(np.full((nx, 1), h["x_min"]), np.full((nx, 1), h["x_max"]), np.full((nu, 1), h["u_min"]), np.full((nu, 1), h["u_max"]))
    s.update_settings(max_iter=h["max_iter"])
    s.set_option("advance_x0", 1)
    if os.environ.get("TORCH_STREAM"):                   # as bench.py runs it: the solver on a torch-created stream
        import torch
        _stream = torch.cuda.Stream(device=0)
        s.set_stream(_stream.cuda_stream)
    xref = np.tile(np.array(h["xref"], dtype=np.float64).reshape(nx, 1), (1, N))
    x0 = np.array(h["x0"], dtype=np.float64)
    bw = s.algorithmic_bytes()
    print(f"batch {B}; bytes_warm = {bw} B per solve; records {B * N * (nx + nu) * 8 / 2**20:.0f} MiB each")
    print("| store_primal | share_ref | grid waves/CU | launch_order | steps 70-99: us mean | us min | moved B/solve | moved TB/s | formula TB/s |")
    print("|---|---|---|---|---|---|---|---|---|")
    its = None
    quick = bool(os.environ.get("QUICK"))
    for sp in ((1,) if quick else (1, 0)):
        for sr in (1, 0):
            for g in ((0,) if quick else (0, 8)):
                for lo in (0, 1):
                    s.set_option("store_primal", sp); s.set_option("share_ref", sr); s.set_option("grid_waves_per_cu", g); s.set_option("launch_order", lo)
                    runs = []
                    for _ in range(6):
                        s.reset()
                        s.set_x_ref(xref, broadcast=True)
                        s.set_x0(x0, broadcast=True)
                        s.set_option("timing", 100)
                        it = []
                        for k in range(100):
                            s.solve_async()
                            if its is None:
                                it.append(int(s.status()["iter"][0]))
                        if its is None:
                            its = np.array(it)
                        runs.append(s.timing_ms()[70:])
                    ms = np.median(np.array(runs[1:]), axis=0)
                    moved = np.mean([bw - (8 * S if sr else 0) - (8 * S if i == 1 else 0) - (8 * S if sp == 0 else 0) for i in its[70:]])
                    t = ms.mean() * 1e-3
                    print(f"| {sp} | {sr} | {g} | {lo} | {ms.mean()*1e3:.1f} | {ms.min()*1e3:.1f} | {moved:.0f} | {moved*B/t/1e12:.2f} | {bw*B/t/1e12:.2f} |", flush=True)
                    if quick:
                        print("    per step us:", " ".join(f"{v*1e3:.0f}" for v in ms), " iters:", " ".join(str(i) for i in its[70:]), flush=True)
    s.close()
