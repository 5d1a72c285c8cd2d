#!/bin/bash
# SQ counters of one fused-pair shape: bash tools/pmc_pair.sh C T k d
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_pair_a -- python $R/tools/probe_pair_one.py $1 $2 $3 $4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_pair_b -- python $R/tools/probe_pair_one.py $1 $2 $3 $4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc_pair_c -- python $R/tools/probe_pair_one.py $1 $2 $3 $4 > /dev/null 2>&1
python - <<PY
import csv,glob,collections,os
for d in ("a","b","c"):
    if not glob.glob("$R/gpurun_out/pmc_pair_%s/*/*_counter_collection.csv"%d): print("pass",d,"missing"); continue
    f=max(glob.glob("$R/gpurun_out/pmc_pair_%s/*/*_counter_collection.csv"%d), key=os.path.getmtime)
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pair" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c,v in agg.items(): print("%-28s n=%d mean=%.4g"%(c,len(v),sum(v)/len(v)))
    kt=max(glob.glob("$R/gpurun_out/pmc_pair_%s/*/*_kernel_trace.csv"%d), key=os.path.getmtime)
    ds=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(kt)) if "pair" in r["Kernel_Name"]]
    print("duration us", sum(ds)/len(ds), len(ds))
PY
