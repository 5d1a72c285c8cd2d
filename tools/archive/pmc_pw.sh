#!/bin/bash
# SQ counters of one pointwise GEMM launch: bash tools/pmc_pw.sh cin cout gelu res variant [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${6:-pw}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/pmc_$TAG
rm -rf $D
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $D -- python $R/tools/probe_pw_one.py $1 $2 $3 $4 $5 > /dev/null 2>&1
python - <<PY
import csv,glob,collections,os
f=max(glob.glob("$D/*/*_counter_collection.csv"), key=os.path.getmtime)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gemm_pw" in r["Kernel_Name"] or "conv_mfma_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
kt=max(glob.glob("$D/*/*_kernel_trace.csv"), key=os.path.getmtime)
ds=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(kt)) if "gemm_pw" in r["Kernel_Name"] or "conv_mfma_kernel" in r["Kernel_Name"]]
a={c:sum(v)/len(v) for c,v in agg.items()}
dur=sum(ds)/len(ds)
clk=a["GRBM_GUI_ACTIVE"]/8/(dur*1e-6)/1e9
print("$TAG: $1->$2 gelu=$3 res=$4 variant=$5  duration %.1f us (n=%d)  clock %.3f GHz  MFMA pipe utilisation %.3f" % (dur, len(ds), clk, a["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*a["GRBM_GUI_ACTIVE"]/8)))
wc=a["SQ_WAVE_CYCLES"]
print("   wave-cycles split: active %.3f  wait_inst %.3f  wait_any(parked) %.3f ; VALU insts per wave-kcycle %.2f ; valu active %.3f ; waves/SIMD avg %.2f" % (
    a["SQ_ACTIVE_INST_ANY"]/wc, a["SQ_WAIT_INST_ANY"]/wc, a["SQ_WAIT_ANY"]/wc, a["SQ_INSTS_VALU"]/wc*1000, a["SQ_ACTIVE_INST_VALU"]/wc, wc*4/(1024*a["GRBM_GUI_ACTIVE"]/8)))
PY
rm -rf $D
