#!/usr/bin/env python
"""Groups a rocprofv3 --kernel-trace CSV by (kernel, workgroup count): `--stats` lumps the B=32 step launches with the B=1
latency launches of the same template, this separates them.
usage: python tools/trace_by_grid.py name=<rocprof_dir> [name=<dir> ...] > out.json"""
import collections, csv, glob, json, os, sys


def group(d):
    path = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        wg = max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
        threads = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        acc[(r["Kernel_Name"], threads // wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = [{"kernel": k, "workgroups": g, "calls": len(v), "avg_us": round(sum(v) / len(v), 2), "total_us": round(sum(v), 1)}
            for (k, g), v in acc.items()]
    return sorted(rows, key=lambda r: -r["total_us"])


print(json.dumps({a.split("=", 1)[0]: group(a.split("=", 1)[1]) for a in sys.argv[1:]}, indent=0))
