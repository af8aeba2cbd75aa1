import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    n = r["Kernel_Name"][:60]
    if "admm_solve" in n or "repack" in n:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 is None: t0 = st
        print(f"{(st-t0)/1e3:9.1f} us  +{(en-st)/1e3:8.1f} us  grid {r.get('Grid_Size','?'):>9}  {n}")
