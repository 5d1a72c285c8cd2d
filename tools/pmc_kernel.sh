#!/bin/bash
# SQ counters of the kernels matching a name pattern in a model forward: bash tools/pmc_kernel.sh <model> <kernel-substring> [B]
R=${GRAFT_REPO_ROOT:-$(pwd)}
M=$1; PAT=$2; B=${3:-}
cd /tmp && export TMPDIR=/tmp
for pass in A B; do
  D=$R/gpurun_out/pmck_$pass; rm -rf $D
  if [ $pass = A ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE";
  else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE"; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python $R/tools/probe_model.py $M $B 2 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections,os
f=max(glob.glob("$D/*/*_counter_collection.csv"), key=os.path.getmtime)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "$PAT" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
kt=max(glob.glob("$D/*/*_kernel_trace.csv"), key=os.path.getmtime)
ds=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(kt)) if "$PAT" in r["Kernel_Name"]]
print("pass $pass: $PAT in $M: %d launches, mean %.1f us, total %.2f ms" % (len(ds), sum(ds)/len(ds), sum(ds)/1e3))
for c,v in sorted(agg.items()): print("   %-24s sum %.4g  mean %.4g" % (c, sum(v), sum(v)/len(v)))
PY
  rm -rf $D
done
