#!/usr/bin/env python
"""Concurrency of one replayed throughput step (three branch streams) from a rocprofv3 --kernel-trace CSV of
tools/probe_step.py: per stage (delimited by the upsampler convs) wall time, summed kernel time, time with 1 / 2 / 3+ kernels in
flight, the last kernel of each queue — what the step is bound by when it is not the sum of its kernels.
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/probe_step.py <model>;  python tools/step_timeline.py <dir>"""
import csv, glob, os, sys
path = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in csv.DictReader(open(path))
              if "fv::" in r["Kernel_Name"])
# steps are separated by host syncs (> 200 us of nothing)
steps, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur) > 200_000:
        steps.append(cur); cur = [r]
    else:
        cur.append(r)
steps.append(cur)
st = steps[-1]
t0 = st[0][0]
print(f"{len(steps)} steps in the trace; last one: {len(st)} kernels, {(max(x[1] for x in st) - t0) / 1e6:.3f} ms wall, "
      f"{sum(x[1] - x[0] for x in st) / 1e6:.3f} ms summed kernel time")


def concurrency(ks, a, b):
    ev = []
    for s, e, _, _ in ks:
        s, e = max(s, a), min(e, b)
        if e > s:
            ev += [(s, 1), (e, -1)]
    ev.sort()
    hist, n, last = {}, 0, a
    for t, d in ev:
        hist[n] = hist.get(n, 0) + t - last
        last, n = t, n + d
    hist[n] = hist.get(n, 0) + b - last
    return hist


# stage boundaries: kernels on the main queue whose name says transposed conv are not distinguishable by name; use the pattern
# "a kernel that runs alone on the main queue right after every other queue went idle" = fork points: simply cut at the starts
# of kernels that begin while nothing else is in flight and that are followed by >= 2 concurrent queues
cuts = []
for i, (s, e, q, name) in enumerate(st):
    if all(x[1] <= s for x in st[:i]) and i > 0:
        cuts.append(s)
cuts = [t0] + cuts + [max(x[1] for x in st)]
print(f"{'segment':>8} {'wall us':>9} {'kernel-sum us':>13} {'idle':>7} {'1 in flight':>11} {'2':>7} {'3+':>7}  kernels")
for a, b in zip(cuts[:-1], cuts[1:]):
    ks = [x for x in st if x[0] >= a and x[0] < b]
    if b - a < 20_000:
        continue
    h = concurrency(st, a, b)
    three = sum(v for k, v in h.items() if k >= 3)
    print(f"{(a - t0) / 1e3:8.0f} {(b - a) / 1e3:9.1f} {sum(x[1] - x[0] for x in ks) / 1e3:13.1f} {h.get(0, 0) / 1e3:7.1f} {h.get(1, 0) / 1e3:11.1f} "
          f"{h.get(2, 0) / 1e3:7.1f} {three / 1e3:7.1f}  {len(ks)}")
h = concurrency(st, cuts[0], cuts[-1])
tot = cuts[-1] - cuts[0]
print("whole step: idle %.1f %%, 1 kernel %.1f %%, 2 kernels %.1f %%, 3+ %.1f %%" %
      (100 * h.get(0, 0) / tot, 100 * h.get(1, 0) / tot, 100 * h.get(2, 0) / tot, 100 * sum(v for k, v in h.items() if k >= 3) / tot))
