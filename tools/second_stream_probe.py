#!/usr/bin/env python3
"""Do other HIP streams in the process cost the library's launch-bound paths anything?  (Asked when a build that created the second
stream of "step_regroup_streams" = 2 inside a split solve ran (4,4,10) x 131 072 10 % slower -- which turned out to be the stream's
CREATION landing in the clock-checked probe of the split solve, whose verdict then went the wrong way; the buffers are now made
before the probe.)  Answer: no (profiles/r04_second_stream_probe.md).  Conditions, each in a process of its own: (a) the batch's
stream alone, (b) torch initialised and one kernel run on ITS current stream (what bench.py's process holds), (c) one more idle
stream created through torch, (d) as (b) but the batch launched on torch's stream.  Per condition: best of 10 of the split solve
[ms] (one-row form), and the median warm one-launch-per-step solve of 65 536 quadrotors [us].
    python tools/second_stream_probe.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(cond):
    keep = []
    if cond in "bcd":
        import torch
        keep.append(torch.zeros(1024, device="cuda:0") + 1)
        torch.cuda.synchronize()
        if cond == "c":
            keep.append(torch.cuda.Stream())
    import tinympc_amd as tm
    prob, rng = tm.random_problem(4, 4, 10)
    B = 131072
    s = tm.TinyBatchSolver.from_problem(prob, B)
    if cond == "d":
        import torch
        s.set_stream(torch.cuda.current_stream().cuda_stream)
    s.set_bound_constraints(np.full((4, 1), -1e17), np.full((4, 1), 1e17), np.full((4, 1), -0.5), np.full((4, 1), 0.5))
    s.update_settings(max_iter=500)
    s.set_option("half_rows", 0)
    s.set_x0(rng.uniform(-1, 1, (B, 4))); s.set_x_ref(np.repeat(rng.uniform(-0.2, 0.2, (B, 4, 1)), 10, axis=2))
    ms = []
    for _ in range(14):
        s.reset(); s.set_option("timing", 1); s.solve_async(); ms.append(float(np.sum(s.timing_ms())))
    split = min(ms[4:])
    s.close()
    # warm quadrotor steps, one launch each
    prob, extra = tm.load_problem("quadrotor_20hz")
    B = 65536
    q = tm.TinyBatchSolver.from_problem(prob, B)
    if cond == "d":
        import torch
        q.set_stream(torch.cuda.current_stream().cuda_stream)
    h = extra["hover"]
    q.set_bound_constraints(np.full((12, 1), h["x_min"]), np.full((12, 1), h["x_max"]), np.full((4, 1), h["u_min"]), np.full((4, 1), h["u_max"]))
    q.update_settings(max_iter=h["max_iter"])
    q.set_x_ref(np.tile(np.array(h["xref"], dtype=float).reshape(-1, 1), (1, prob["N"])), broadcast=True)
    q.set_x0(np.array(h["x0"], dtype=float), broadcast=True)
    q.set_option("advance_x0", 1)
    for _ in range(70):
        q.solve_async()
    q.set_option("timing", 30)
    for _ in range(30):
        q.solve_async()
    us = 1e3 * float(np.median(q.timing_ms()))
    q.close()
    print("%s %.3f %.1f" % (cond, split, us))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        names = dict(a="the batch's own stream alone", b="+ torch initialised, one kernel on its current stream", c="+ one more idle stream (torch.cuda.Stream())",
                     d="torch initialised, the batch ON torch's current stream")
        print("| streams in the process | (4,4,10) x 131 072 split solve, ms | warm quadrotor step x 65 536, us |\n|---|---|---|")
        for c in "abcd":
            out = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True)
            try:
                _, a, b = out.stdout.strip().split("\n")[-1].split()
                print("| %s | %s | %s |" % (names[c], a, b), flush=True)
            except Exception:                             # noqa: BLE001
                print("| %s | failed: %s | |" % (names[c], (out.stderr or out.stdout)[-300:].replace("\n", " ")), flush=True)
