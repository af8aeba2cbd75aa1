#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) as the --stats kernel table:
   python tools/rocpd_summary.py gpurun_out/x/prof/foo_results.db > profiles/foo_kernel_stats.csv"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                 "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                 "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPR,AGPR,SGPR,LDS,Scratch,GridX,WorkgroupX")
for r in rows:
    print('"%s",%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s,%s,%s,%s,%s' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot,
                                                             r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
