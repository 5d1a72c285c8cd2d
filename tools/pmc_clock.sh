#!/bin/bash
# effective shader clock of each kernel of a B=1 HiFiGAN forward: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/pmc_clock; rm -rf $D
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $D -- python $R/tools/probe_model.py hifigan ${1:-1} 20 > /dev/null 2>&1
python - <<PY
import csv,glob,collections,os
f=max(glob.glob("$D/*/*_counter_collection.csv"), key=os.path.getmtime)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="GRBM_GUI_ACTIVE":
        d=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); n=r["Kernel_Name"]; g=int(r["Grid_Size"])//max(int(r["Workgroup_Size"]),1)
        agg[(n[:70],g)].append((float(r["Counter_Value"])/8/(d*1e-9)/1e9, d/1e3))
for k,v in sorted(agg.items(), key=lambda kv:-sum(x[1] for x in kv[1]))[:14]:
    print("%-72s grid %5d  n=%3d  mean %.1f us  clock %.2f GHz" % (k[0],k[1],len(v),sum(x[1] for x in v)/len(v), sum(x[0] for x in v)/len(v)))
PY
rm -rf $D
