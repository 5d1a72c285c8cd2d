#!/usr/bin/env python
"""From a rocprofv3 --kernel-trace CSV of a multi-stream forward: the transposed-conv (upsampler) and branch-mean kernels of the last
forward with their durations.  python tools/trace_upsamplers.py <dir>"""
import csv, glob, os, sys
path = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])) for r in csv.DictReader(open(path))))
sel = [r for r in rows if "mean_of_three" in r[2] or "conv_mfma_kernel<1," in r[2] or "conv_mfma_kernel<2," in r[2] or "conv_mfma_kernel<4," in r[2]]
for s, e, n, g in sel[-10:]:
    print(f"{(e - s) / 1e3:8.1f} us grid={g // 256:6d} {n[:80]}")
print("sum of the last 10: %.1f us" % (sum(e - s for s, e, _, _ in sel[-10:]) / 1e3))
