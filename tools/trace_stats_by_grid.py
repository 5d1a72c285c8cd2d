#!/usr/bin/env python
"""rocprofv3's --stats table merges launches of one kernel instance whatever their grid (the Winograd conv of the C = 128 and the C = 256
stage is the same template instance).  This splits a --kernel-trace CSV by (kernel, workgroups): calls, average / min / max ns — the
per-launch-shape view bench.py's `roofline` uses.    python tools/trace_stats_by_grid.py <rocprofv3 output dir> <out.csv>"""
import collections, csv, glob, os, sys
path = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if "Grid_Size" in r:
        wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
    else:   # kernel-trace CSVs carry the three dimensions separately
        wg = 1
        for ax in "XYZ":
            wg *= max(1, int(r[f"Grid_Size_{ax}"])) // max(1, int(r[f"Workgroup_Size_{ax}"]))
    agg[(r["Kernel_Name"], wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Workgroups", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
    for (name, wg), d in rows:
        w.writerow([name, wg, len(d), sum(d), f"{sum(d) / len(d):.1f}", min(d), max(d)])
print(f"{len(rows)} (kernel, grid) rows from {path}")
