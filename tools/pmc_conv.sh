#!/bin/bash
# SQ counters of one fused-conv shape (whichever conv kernel the layer takes; FV_WINO=0 for the direct one): bash tools/pmc_conv.sh C T k d
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_conv_a -- python $R/tools/probe_one.py $1 $2 $3 $4 > /dev/null 2>&1
python - <<PY
import csv,glob,collections,os
f=max(glob.glob("$R/gpurun_out/pmc_conv_a/*/*_counter_collection.csv"), key=os.path.getmtime)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if any(s in r["Kernel_Name"] for s in ("conv_mfma_kernel", "conv_wino_kernel", "conv_wino4_kernel", "conv_wino44_kernel")): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c,v in agg.items(): print("%-28s n=%d mean=%.4g"%(c,len(v),sum(v)/len(v)))
kt=max(glob.glob("$R/gpurun_out/pmc_conv_a/*/*_kernel_trace.csv"), key=os.path.getmtime)
ds=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(kt)) if any(s in r["Kernel_Name"] for s in ("conv_mfma_kernel", "conv_wino_kernel", "conv_wino4_kernel", "conv_wino44_kernel"))]
print("duration us", sum(ds)/len(ds), len(ds))
a={c:sum(v)/len(v) for c,v in agg.items()}
clk=a["GRBM_GUI_ACTIVE"]/8/(sum(ds)/len(ds)*1e-6)/1e9
print("clock GHz %.3f  MFMA pipe utilisation %.3f"%(clk, a["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*a["GRBM_GUI_ACTIVE"]/8)))
PY
