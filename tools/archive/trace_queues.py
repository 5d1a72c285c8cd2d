import csv, glob, os, sys, collections
path = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"], int(r["Grid_Size_X"])) for r in csv.DictReader(open(path))))
# last 76 kernels with big grids
big=[r for r in rows if r[4]//256 > 400]
last=big[-76:]
t0=last[0][0]
cnt=collections.Counter()
for s,e,q,n,g in last:
    cnt[q]+=1
print("queues used by the last step's kernels:", dict(cnt))
for s,e,q,n,g in last[:40]:
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:7.1f}us q={q} grid={g//256:5d} {n[:60]}")
