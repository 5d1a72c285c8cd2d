#!/usr/bin/env python
"""Timeline of the last replayed B=1 forward in a rocprofv3 --kernel-trace CSV of tools/probe_latency.py: start / end / queue of
every kernel (relative to the first), so that fork / join gaps and the critical branch can be read off.
python tools/latency_timeline.py <rocprof_dir> [n_kernels_per_forward=76]"""
import csv, glob, os, sys
path = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 76
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"], r["Grid_Size_X"]) for r in csv.DictReader(open(path))))
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur) > 100_000:
        bursts.append(cur); cur = [r]
    else:
        cur.append(r)
bursts.append(cur)
cands = [x for x in bursts if len(x) % n == 0] or bursts
b = max(cands, key=len)[-n:]   # back-to-back replayed forwards merge into one long burst: its last forward (the profiled pass — single
                                # stream, events around every launch — is a burst of its own, n kernels long)
t0 = b[0][0]
end = t0
for s, e, q, name, grid in b:
    gap = s - end if s > end else 0
    print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f}us q={q} grid={int(grid) // 256:4d} {'GAP %.1f' % (gap / 1e3) if gap > 2000 else '':10s} {name[:70]}")
    end = max(end, e)
