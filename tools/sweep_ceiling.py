#!/usr/bin/env python3
"""CPU only: what each cell of the config-5 sweep CAN reach with the mapping it runs on, and where the measured
FP64 fraction goes, factor by factor.

    python tools/sweep_ceiling.py profiles/r06_sweep_config5.json profiles/r06_sweep_counters.json profiles/r06_sweep_counters_uniform.json > profiles/r06_sweep_ceiling.md

FP64 peak = every issue slot of every SIMD an FMA on all 64 lanes (128 FLOP per wave-instruction; 78.6 TFLOP/s at 2.4 GHz).  A cell's
measured fraction is the product of three factors, each read from a different source:

  instr_eff     (the MAPPING)  algorithmic FLOPs per wave-iteration (SURVEY.md 8 footnote 1 x instances per wave) / (128 x instructions
                of one loop pass).  The loop's length is MEASURED: `sweep_bench.py --uniform 100` under rocprofv3 --pmc SQ_INSTS_VALU runs
                every instance for exactly 100 iterations (tolerances 0: no lock step, no split, every row busy), instructions issued /
                wave-iterations is what one pass costs (the static count of the gfx950 assembly -- hipcc -S, the loop with the most FP64
                FMAs -- stands next to it: the tile kernel's loop is a slot state machine whose load / store / regenerate blocks the
                static count includes although they run once per solve).  Folds lane use ((nx+nu) of the 16 W lanes), rows idle in a
                sweep (R > 1), the FP64 share of the instruction stream and the FMA density of the FP64 instructions.  THIS is the
                cell's ceiling: every issue slot taken, every row of every wave busy with an instance of its own.
  packing       (LOCK STEP and everything outside the loop, from SQ_INSTS_VALU of a profiled run: tools/sweep_counters.py)  loop
                instructions per ideal wave-iteration / VALU instructions actually issued per ideal wave-iteration (iterations of all
                instances / instances per wave).  Rows that idle while a neighbour iterates still issue; load / store code, probes and
                launch tails issue too.
  issue_util    (the SCHEDULE, same counters)  4 x VALU instructions / (1024 SIMDs x kernel time x 2.4 GHz): the share of the chip's
                issue slots taken -- waits for LDS / memory / dependent results, one wave per SIMD, clocks below nominal.

  measured = instr_eff x packing x issue_util (the profiled run's own fraction: column "product"; the clean run's: "measured").
  lockstep_eff  for reference: mean iterations / E[max over the instances of a wave] from the cell's own iteration histogram (random
                grouping, static rows) -- what packing would be WITHOUT the dynamic slot forms / split solves.
"""
import collections
import concurrent.futures
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tinympc_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_loop_stats as ils  # noqa: E402


def flops_per_iter(nx, nu, N):
    sys.path.insert(0, ROOT)
    import bench
    return bench.flops_per_iter(nx, nu, N)


def tile_first_entry(nx, nu, N):
    for line in open(os.path.join(CSRC, "tile_dims.txt")):
        f = line.split("#")[0].split()
        if len(f) >= 5 and tuple(map(int, f[:3])) == (nx, nu, N):
            return int(f[3]), int(f[4]), (int(f[5]) if len(f) > 5 else 99)
    return None


def analyse_cell(cell):
    nx, nu, N, kern = cell["nx"], cell["nu"], cell["N"], cell["kernel"]
    nz = nx + nu
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "u.hip")
        with open(src, "w") as f:
            if nz <= 16:
                f.write("#define TINYMPC_FUSED_NX %d\n#define TINYMPC_FUSED_NU %d\n" % (nx, nu))
            f.write('#include "%s/kernel_entry.hpp"\n#include "%s/tile_kernel.hip.h"\n' % (CSRC, CSRC))
            if kern == "tile":
                W, R, lm = tile_first_entry(nx, nu, N)
                ipw = (64 // (8 * R)) if W == 0 else 64 // (16 * W * R)
                dyn = ipw > 1
                f.write("namespace tinympc_amd { TileKernelFn pick() { return tile_kernel_or_null<%d, %d, %d, %d, %d, %d, true, %s>(); } }\n" %
                        (nx, nu, N, W, R, lm, "true" if dyn else "false"))
                form = "tile W=%s R=%d LM=%s%s" % ("half" if W == 0 else W, R, lm, " dyn" if dyn else "")
                lane_use = (2 * nz / 16.0) if W == 0 else nz / (16.0 * W)
            else:
                half = nz <= 8
                ipw = 8 if half else 4
                f.write("namespace tinympc_amd { template __global__ void admm_solve_kernel<%d, %d, %d, false, false, 2, 0, false, LIN_KMAX, false, true, %s>(const SolveArgs); }\n" %
                        (nx, nu, N, "true" if half else "false"))
                form = "one-row%s" % (" HALF" if half else "")
                R = 1
                lane_use = (2 * nz / 16.0) if half else nz / 16.0
        out = os.path.join(tmp, "u.s")
        p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
        if p.returncode != 0:
            return dict(cell, error=p.stderr[-300:])
        best = None
        for sym, lines, loop, meta in ils.analyse(out):
            if not loop:
                continue
            body = lines[loop[0]:loop[1] + 1]
            nf = sum(1 for l in body if re.match(r"v_fma?c?_f64", l))
            if best is None or nf > best[0]:
                best = (nf, body, meta)
        if best is None:
            return dict(cell, error="no loop")
        _, body, meta = best
    # the tile kernel's loop is a slot state machine: the paths that load an instance into a free slot and store a finished one sit in
    # it behind branches and run once per SOLVE -- basic blocks with memory traffic and no sweep arithmetic are not part of an iteration
    if kern == "tile":
        blocks, cur = [], []
        for l in body:
            cur.append(l)
            if re.match(r"s_c?branch", l):
                blocks.append(cur); cur = []
        if cur:
            blocks.append(cur)
        body = [l for b in blocks if not (any(x.startswith(("global_load", "global_atomic")) for x in b) and not any("_f64_dpp" in x for x in b)) for l in b]
    c = collections.Counter(ils.classify(l) for l in body)
    total = len(body)
    regs = meta.get("next_free_vgpr", 512)
    waves = max(1, min(2 if kern != "tile" else 8, 512 // max(regs, 1)))
    waves = 2 if waves >= 2 else 1
    fl = flops_per_iter(nx, nu, N)
    instr_eff = fl * ipw / (2.0 * 64.0 * total)
    # lock step: E[max of ipw draws] from the histogram
    hist = {int(k): v for k, v in cell.get("iter_histogram", {}).items()}
    n = float(sum(hist.values())) or 1.0
    mean = sum(k * v for k, v in hist.items()) / n
    cdf, acc, emax, prev = {}, 0.0, 0.0, 0.0
    for k in sorted(hist):
        acc += hist[k] / n
        emax += k * (acc ** ipw - prev ** ipw)
        prev = acc
    lock = mean / emax if emax > 0 else 1.0
    return dict(cell, form=form, ipw=ipw, lane_use=lane_use, rows=R, loop_instr=total, fp64_share=c["fp64"] / float(total), regs=regs, waves=waves,
                instr_eff=instr_eff, lockstep_eff=lock, ceiling=instr_eff, flops_per_wave_iter=fl * ipw)


def main():
    cells = json.load(open(sys.argv[1]))
    if isinstance(cells, dict):
        cells = cells.get("cells", [])
    counters = {}
    if len(sys.argv) > 2:
        counters = {(c["nx"], c["nu"], c["N"]): c for c in json.load(open(sys.argv[2]))}
    uniform = {}
    if len(sys.argv) > 3:
        uniform = {(c["nx"], c["nu"], c["N"]): c for c in json.load(open(sys.argv[3]))}
    with concurrent.futures.ThreadPoolExecutor(8) as ex:
        rows = list(ex.map(analyse_cell, cells))
    for r in rows:                                     # the loop's MEASURED length where the uniform run has it (the static count stays in its own column)
        u = uniform.get((r.get("nx"), r.get("nu"), r.get("N")))
        if u and "error" not in r and u.get("instr_per_instance_iter"):
            r["loop_static"] = r["loop_instr"]
            r["loop_instr"] = u["instr_per_instance_iter"] * r["ipw"]
            r["instr_eff"] = r["ceiling"] = r["flops_per_wave_iter"] / (128.0 * r["loop_instr"])
            r["uniform_util"] = u["issue_util"]
    print("# Round 6: the config-5 sweep cell by cell -- the measured FP64 fraction as mapping x packing x schedule (tools/sweep_ceiling.py; measured: %s, counters: %s)\n"
          % (os.path.basename(sys.argv[1]), os.path.basename(sys.argv[2]) if len(sys.argv) > 2 else "none"))
    print("FP64 peak" + __doc__.split("FP64 peak")[1])
    print("| (nx,nu,N) | form | inst / wave | lane use | loop instr (measured; static count) | FP64 share (static) | VGPR+AGPR | scratch B/lane | waves / SIMD | **instr_eff = ceiling** | packing | issue_util | product | measured | measured / ceiling | lockstep_eff (static rows) | largest loss |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    below = 0
    tot = 0.0
    for r in sorted(rows, key=lambda r: (r["N"], r["nx"], r["nu"])):
        if "error" in r:
            print("| (%d,%d,%d) | error: %s |" % (r["nx"], r["nu"], r["N"], r["error"].replace("\n", " ")[:80]))
            continue
        tot += r["ms"]
        c = counters.get((r["nx"], r["nu"], r["N"]))
        ratio = r["fp64_frac"] / r["ceiling"]
        pack = util = prod = None
        regs, scratch = r["regs"], None
        if c and c.get("instr_per_instance_iter"):
            issued_per_wave_iter = c["instr_per_instance_iter"] * r["ipw"]
            pack = r["loop_instr"] / issued_per_wave_iter
            util = c["issue_util"]
            prod = r["instr_eff"] * pack * util
            regs, scratch = c["vgpr"] + c["agpr"], c["scratch_bytes_per_lane"]
        loss = []
        cand = [("mapping: lane use %.2f" % r["lane_use"], r["lane_use"]), ("mapping: %.0f %% of the loop is not FP64" % (100 * (1 - r["fp64_share"])), r["fp64_share"])]
        if pack is not None:
            cand += [("packing %.2f%s" % (pack, "" if "dyn" in r["form"] else " (static rows: lock step %.2f)" % r["lockstep_eff"]), min(pack, 1.0)),
                     ("schedule: issue_util %.2f at %d wave%s per SIMD" % (util, r["waves"], "" if r["waves"] == 1 else "s"), util)]
        cand.sort(key=lambda kv: kv[1])
        loss = "; ".join(k for k, v in cand[:2] if v < 0.9) or "at its ceiling"
        if ratio < 0.8:
            below += 1
        f = lambda v, fmt="%.2f": "--" if v is None else fmt % v
        print("| (%d,%d,%d) | %s | %d | %.2f | %d; %s | %.2f | %s | %s | %d | **%.3f** | %s | %s | %s | %.3f | %.2f | %.2f | %s |" %
              (r["nx"], r["nu"], r["N"], r["form"], r["ipw"], r["lane_use"], r["loop_instr"], r.get("loop_static", "--"), r["fp64_share"], regs, f(scratch, "%d"), r["waves"], r["instr_eff"],
               f(pack), f(util), f(prod, "%.3f"), r["fp64_frac"], ratio, r["lockstep_eff"], loss))
    print("\nsweep total %.1f ms; %d of %d cells below 0.8 of their own ceiling (instr_eff: every issue slot taken, every row busy)." % (tot, below, len(rows)))


if __name__ == "__main__":
    main()
