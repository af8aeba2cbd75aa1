"""kernel trace of a rocprofv3 run (--kernel-trace --output-format csv) in time order: start (us from the first kernel), duration,
gap to the kernel before, grid, name -- the solve kernels and the step_regroup / repack passes between them
    python tools/trace_summary.py <dir> [max rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, last, n = None, None, 0
for r in rows:
    name = r["Kernel_Name"][:48]
    if "admm_solve" in name or "repack" in name or "regroup" in name or "lockstep" in name:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 is None: t0 = st
        gap = (st - last) / 1e3 if last is not None else 0.0
        print(f"{(st-t0)/1e3:10.1f} us  +{(en-st)/1e3:9.1f} us  gap {gap:8.1f}  grid {r.get('Grid_Size','?'):>9}  q{r.get('Queue_Id','?')}  {name}")
        last = en if last is None else max(last, en)
        n += 1
        if len(sys.argv) > 2 and n >= int(sys.argv[2]): break
