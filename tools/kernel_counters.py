#!/usr/bin/env python3
"""rocprofv3 PMC passes of an arbitrary command -> one markdown table per solve kernel (tools/gpu_stage.sh stage "counters").
    python tools/kernel_counters.py <dir with sq/ fetch/ write/ trace/ sub-directories> > profiles/rNN_kernel_counters.md

Columns: launches seen, average duration (kernel trace), VALU instructions per wave-cycle (SQ_INSTS_VALU / SQ_WAVE_CYCLES: a
wave issues at most one VALU instruction per 4 cycles... 0.25 = a lone wave issuing back to back; with W waves per SIMD each
wave's share is 0.25 x busy / W), waves, HBM bytes per launch (FETCH_SIZE x 2 per MI355X_MICROARCH.md's gfx950 correction for
coalesced streams is NOT applied here: raw counter x 64 B / x 32 B as the guide prescribes is noted in the header line)."""
import csv
import glob
import os
import re
import sys

root = sys.argv[1]


def rows(sub):
    fs = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)
    for f in fs:
        for r in csv.DictReader(open(f)):
            yield r


def short(name):
    m = re.search(r"(admm_\w+)<([^>]*)>", name)
    if not m:
        return None
    a = [t.strip() for t in m.group(2).split(",")]
    if m.group(1) == "admm_solve_kernel":
        tag = "one-row"
        flags = []
        if len(a) > 3 and a[3] == "true": flags.append("cone")
        if len(a) > 9 and a[9] == "true": flags.append("adaptive")
        if len(a) > 10 and a[10] == "true": flags.append("UB")
        return f"{tag} ({a[0]},{a[1]},{a[2]})" + (" " + "+".join(flags) if flags else "")
    if m.group(1) == "admm_tile_kernel":
        return f"tile ({a[0]},{a[1]},{a[2]}) {a[3]}x{a[4]}"
    return m.group(1)


acc = {}
for sub in ("sq", "fetch", "write"):
    for r in rows(sub):
        k = short(r.get("Kernel_Name", ""))
        if k is None:
            continue
        d = acc.setdefault(k, {})
        c = r["Counter_Name"]
        d.setdefault(c, []).append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r.get("Kernel_Name", ""))
        if k:
            dur.setdefault(k, []).append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
print("| kernel | launches | avg ms | waves / launch | VALU inst / wave-cycle | SQ busy cycles / launch | FETCH_SIZE (KB) / launch | WRITE_SIZE (KB) / launch |")
print("|---|---|---|---|---|---|---|---|")
avg = lambda v: sum(v) / len(v) if v else float("nan")
for k in sorted(acc):
    d = acc[k]
    wc = avg(d.get("SQ_WAVE_CYCLES", []))
    print(f"| {k} | {len(dur.get(k, []))} | {avg(dur.get(k, [])):.3f} | {avg(d.get('SQ_WAVES', [])):.0f} | {avg(d.get('SQ_INSTS_VALU', [])) / wc if wc == wc and wc else float('nan'):.3f} | "
          f"{avg(d.get('SQ_BUSY_CYCLES', [])):.3e} | {avg(d.get('FETCH_SIZE', [])):.0f} | {avg(d.get('WRITE_SIZE', [])):.0f} |")
