#!/usr/bin/env python3
"""CPU only: what each cell of the config-5 sweep CAN reach with the mapping it runs on, so that a measured FP64
fraction of 0.11 splits into what the mapping forbids and what is left on the table.

    python tools/sweep_ceiling.py profiles/r05_sweep_config5.json > profiles/r06_sweep_ceiling.md

FP64 peak = every issue slot of every SIMD an FMA on all 64 lanes.  Per cell, from the gfx950 assembly of the instantiation the cell
launches (hipcc -S, the innermost loop with the most FP64 FMAs = one ADMM iteration of every instance the wave holds):

  instr_eff     algorithmic FLOPs per wave-iteration (SURVEY.md 8 footnote 1 x instances per wave) / (2 x 64 x instructions per
                wave-iteration) -- folds lane use ((nx+nu) of the 16 W lanes), rows idle in a sweep (R > 1), the FP64 share of the
                instruction stream and the FMA density of the FP64 instructions
  issue_util    the share of issue slots a SIMD fills, from SQ_INSTS_VALU / SQ_WAVE_CYCLES of profiles/r04_kernel_counters_table.md: two
                waves per SIMD interleave their dependent chains (0.50 + 0.50 = every slot); ONE one-row wave reaches 0.79-0.89 (its fused
                step blocks carry two chains), ONE tile-kernel wave 0.47-0.67 -> 1.0 / 0.85 / 0.58
  lockstep_eff  rows of a wave iterate together: mean iterations / E[max over the instances of a wave] from the cell's own iteration
                histogram (random grouping).  Dynamic slot forms and split solves recover most of it: listed separately, NOT in the ceiling
  ceiling       instr_eff x issue_util;   measured / ceiling = what lock step, load / store phases, launch tails and waits leave
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
    # measured VALU instructions per wave-cycle (profiles/r04_kernel_counters_table.md): two waves 0.50 each = every slot; a lone one-row
    # wave 0.79-0.89 (its fused step blocks interleave two chains), a lone tile wave 0.47-0.67
    issue = 1.0 if waves >= 2 else (0.85 if kern != "tile" else 0.58)
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
                instr_eff=instr_eff, issue_util=issue, lockstep_eff=lock, ceiling=instr_eff * issue)


def main():
    cells = json.load(open(sys.argv[1]))
    if isinstance(cells, dict):
        cells = cells.get("cells", [])
    with concurrent.futures.ThreadPoolExecutor(8) as ex:
        rows = list(ex.map(analyse_cell, cells))
    print("# Round 6: the config-5 sweep cell by cell -- measured FP64 fraction against what the mapping allows (tools/sweep_ceiling.py, measured column: %s)\n" % os.path.basename(sys.argv[1]))
    print(__doc__.split("FP64 peak")[1].join(["FP64 peak", ""]) if False else "FP64 peak" + __doc__.split("FP64 peak")[1])
    print("| (nx,nu,N) | form | inst / wave | lane use | loop instr | FP64 share | regs | waves / SIMD | instr_eff | issue_util | **ceiling** | measured | measured / ceiling | lockstep_eff (static rows) | what blocks it |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    below = 0
    for r in sorted(rows, key=lambda r: (r["N"], r["nx"], r["nu"])):
        if "error" in r:
            print("| (%d,%d,%d) | error: %s |" % (r["nx"], r["nu"], r["N"], r["error"].replace("\n", " ")[:80]))
            continue
        ratio = r["fp64_frac"] / r["ceiling"]
        why = []
        if r["lane_use"] < 0.7:
            why.append("lane use %.2f" % r["lane_use"])
        if r["waves"] < 2:
            why.append("one wave per SIMD")
        if r["fp64_share"] < 0.8:
            why.append("%.0f %% of the stream is not FP64" % (100 * (1 - r["fp64_share"])))
        if ratio < 0.8:
            why.append("measured %.2f of the ceiling: lock step %.2f%s" % (ratio, r["lockstep_eff"], "" if "dyn" in r["form"] else " (static rows)") )
            below += 1
        print("| (%d,%d,%d) | %s | %d | %.2f | %d | %.2f | %d | %d | %.3f | %.2f | **%.3f** | %.3f | %.2f | %.2f | %s |" %
              (r["nx"], r["nu"], r["N"], r["form"], r["ipw"], r["lane_use"], r["loop_instr"], r["fp64_share"], r["regs"], r["waves"], r["instr_eff"],
               r["issue_util"], r["ceiling"], r["fp64_frac"], ratio, r["lockstep_eff"], "; ".join(why) or "at its ceiling"))
    print("\n%d of %d cells below 0.8 of their own ceiling." % (below, len(rows)))


if __name__ == "__main__":
    main()
