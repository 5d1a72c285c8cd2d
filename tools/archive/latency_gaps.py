#!/usr/bin/env python
"""From a rocprofv3 --kernel-trace CSV of tools/probe_latency.py: wall time of the last B=1 forwards (graph replays), the
sum of their kernel durations and the idle gaps between dependent kernels.  python tools/latency_gaps.py <rocprof_dir>"""
import csv, glob, os, sys
path = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
# forwards = bursts separated by > 100 us of idle
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur) > 100_000:
        bursts.append(cur); cur = [r]
    else:
        cur.append(r)
bursts.append(cur)
for b in bursts[-5:]:
    t0, t1 = b[0][0], max(x[1] for x in b)
    busy = 0; end = t0
    for s, e, _ in b:        # union of kernel intervals
        if e > end:
            busy += e - max(s, end); end = e
    print(f"{len(b):3d} kernels  wall {1e-3*(t1-t0):8.1f} us  GPU busy (union) {1e-3*busy:8.1f} us  idle {1e-3*(t1-t0-busy):7.1f} us  sum of durations {1e-3*sum(e-s for s,e,_ in b):8.1f} us")
