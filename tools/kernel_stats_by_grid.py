#!/usr/bin/env python3
"""rocprofv3 --kernel-trace csv -> per (kernel, grid size) launch statistics.  The stock *_kernel_stats.csv averages EVERY launch of a
kernel - the 64-ray parity launch and the warm-up next to the full-frame launches ("30.9 ms average" for a
33.3-ms kernel); grouped by grid size, the full-frame row reproduces bench.py's roofline.kernel_ms directly.
   python tools/kernel_stats_by_grid.py <kernel_trace.csv> [min_launches]"""
import csv, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
min_n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
    g[(name, grid, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
total = sum(sum(v) for v in g.values())
print("kernel,grid_threads,workgroup,launches,avg_us,min_us,max_us,total_us,share")
for (name, grid, wg), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= min_n:
        print(f"\"{name}\",{grid},{wg},{len(v)},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f},{sum(v):.1f},{sum(v) / total:.4f}")
