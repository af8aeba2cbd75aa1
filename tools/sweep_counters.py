#!/usr/bin/env python3
"""CPU: the SQ counters of one profiled sweep run, per cell.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d <dir> -o sweep -- \
        python tools/sweep_bench.py --out <dir>/sweep_pmc.json                       (GPU; tools/sweep_evidence.sh)
    python tools/sweep_counters.py <dir> [solves per cell] [counters JSON of a `sweep_bench.py --uniform K` run] > profiles/r06_sweep_counters.json

Every dispatch of a solve kernel (admm_solve_kernel / admm_tile_kernel / admm_general_kernel <nx, nu, N, ...>) belongs to the cell of
its first three template arguments; a cell's solves (sweep_bench.py: reps + 1 cold solves of the whole batch, each the same work) are
summed: VALU wave-instructions issued, wave-cycles, kernel time (the dispatches' own Start / End timestamps).  With the run's own
sweep_pmc.json (iterations per solve) that gives, per cell,

    instr_per_wave_iter   VALU instructions issued per IDEAL wave-iteration (iterations of all instances / instances per wave): what one
                          iteration of a wave costs in issue slots INCLUDING lock step (rows that idle still issue), load / store code,
                          probes and launch tails
    issue_util            4 x instructions / (1024 SIMDs x kernel time x 2.4 GHz): the share of the chip's VALU issue slots taken
    fp64_frac_profiled    algorithmic FLOPs / (kernel time x 78.6 TFLOP/s) of the profiled run = flops_per_instr / 128 x issue_util
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main():
    root = sys.argv[1]
    sweep = {(r["nx"], r["nu"], r["N"]): r for r in json.load(open(os.path.join(root, "sweep_pmc.json")))}
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # solves per cell in that run (sweep_bench.py --reps 1: two)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"admm_(solve|tile|general)_kernel<\s*(\d+),\s*(\d+),\s*(\d+)([^>]*)>", r["Kernel_Name"])
            if not m:
                continue
            cell = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
            a = acc[cell]
            a[r["Counter_Name"]] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen[cell]:
                seen[cell].add(key)
                a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                a["dispatches"] += 1
                a["vgpr"] = max(a["vgpr"], float(r["VGPR_Count"])); a["agpr"] = max(a["agpr"], float(r["Accum_VGPR_Count"]))
                a["scratch"] = max(a["scratch"], float(r["Scratch_Size"])); a["lds"] = max(a["lds"], float(r["LDS_Block_Size"]))
                a["kernel_" + m.group(1)] += 1
    out = []
    uni = {}
    if len(sys.argv) > 3:                                        # <dir of the --uniform K run>: instructions per wave-iteration with every row busy
        for r in json.load(open(sys.argv[3])):
            uni[(r["nx"], r["nu"], r["N"])] = r
    for cell, a in sorted(acc.items()):
        s = sweep.get(cell)
        if not s:
            continue
        iters = s["iters_per_solve"] * s["batch"] * reps
        fl = s["flops_per_iter"]
        t = a["ns"] * 1e-9
        I = a.get("SQ_INSTS_VALU", 0.0)
        out.append(dict(nx=cell[0], nu=cell[1], N=cell[2], kernel=s["kernel"], dispatches=int(a["dispatches"]), kernel_ms_total=t * 1e3, solves_in_run=reps,
                        valu_insts=I, wave_cycles=a.get("SQ_WAVE_CYCLES", 0.0), busy_cycles=a.get("SQ_BUSY_CYCLES", 0.0), waves=a.get("SQ_WAVES", 0.0),
                        vgpr=int(a["vgpr"]), agpr=int(a["agpr"]), scratch_bytes_per_lane=int(a["scratch"]), lds_bytes=int(a["lds"]),
                        instr_per_instance_iter=I / iters if iters else None,
                        valu_per_wave_cycle=I / a["SQ_WAVE_CYCLES"] if a.get("SQ_WAVE_CYCLES") else None,
                        issue_util=4.0 * I / (1024.0 * t * 2.4e9) if t > 0 else None,
                        flops_per_instr=iters * fl / I if I else None,
                        fp64_frac_profiled=iters * fl / t / 78.6e12 if t > 0 else None,
                        uniform_instr_per_instance_iter=(uni[cell]["instr_per_instance_iter"] if cell in uni else None),
                        uniform_issue_util=(uni[cell]["issue_util"] if cell in uni else None)))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
