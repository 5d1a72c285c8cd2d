R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_step -- python $R/tools/probe_step.py hifigan > /dev/null 2>&1
python $R/tools/step_timeline.py $O/trace_step > $O/step_timeline.txt 2>&1
python - <<PY > $O/step_kernels_instep.txt 2>&1
import csv, glob, os, collections
path = max(glob.glob(os.path.join("$O/trace_step", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"], r.get("Grid_Size_X","")) for r in csv.DictReader(open(path)) if "fv::" in r["Kernel_Name"])
steps, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur) > 200_000: steps.append(cur); cur = [r]
    else: cur.append(r)
steps.append(cur)
st = steps[-1]
agg = collections.OrderedDict()
for s, e, q, n, g in st:
    key = (n.split("(")[0][:70], g)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
for (n, g), (c, t) in agg.items(): print(f"{t:9.1f} us  n={c:3d}  avg {t / c:7.1f}  grid={g}  {n}")
print("total", sum(v[1] for v in agg.values()))
PY
rm -rf $O/trace_step
cd $R
python tools/probe_latency.py > $O/latency_b1.txt 2>&1
cat $O/step_timeline.txt; cat $O/step_kernels_instep.txt; tail -70 $O/latency_b1.txt
