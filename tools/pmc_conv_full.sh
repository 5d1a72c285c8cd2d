#!/bin/bash
# SQ counters of one fused-conv shape in three passes (matrix pipe / instruction mix + VALU, LDS, VMEM activity / LDS conflicts): bash tools/pmc_conv_full.sh C T k d
# (tools/pmc_conv.sh is pass a alone)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_convf_a -- python $R/tools/probe_one.py $1 $2 $3 $4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_convf_b -- python $R/tools/probe_one.py $1 $2 $3 $4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc_convf_c -- python $R/tools/probe_one.py $1 $2 $3 $4 > /dev/null 2>&1
python - <<PY
import csv,glob,collections,os
K=("conv_mfma_kernel", "conv_wino_kernel", "conv_wino4_kernel", "conv_wino44_kernel", "conv_wino44p_kernel")
for d in ("a","b","c"):
    if not glob.glob("$R/gpurun_out/pmc_convf_%s/*/*_counter_collection.csv"%d): print("pass",d,"missing"); continue
    f=max(glob.glob("$R/gpurun_out/pmc_convf_%s/*/*_counter_collection.csv"%d), key=os.path.getmtime)
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(s in r["Kernel_Name"] for s in K): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c,v in agg.items(): print("%-28s n=%d mean=%.4g"%(c,len(v),sum(v)/len(v)))
    kt=max(glob.glob("$R/gpurun_out/pmc_convf_%s/*/*_kernel_trace.csv"%d), key=os.path.getmtime)
    ds=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(kt)) if any(s in r["Kernel_Name"] for s in K)]
    print("duration us", sum(ds)/len(ds), len(ds))
    a={c:sum(v)/len(v) for c,v in agg.items()}
    if "GRBM_GUI_ACTIVE" in a:
        print("clock GHz %.3f  MFMA pipe utilisation %.3f"%(a["GRBM_GUI_ACTIVE"]/8/(sum(ds)/len(ds)*1e-6)/1e9, a["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*a["GRBM_GUI_ACTIVE"]/8)))
PY
rm -rf $R/gpurun_out/pmc_convf_a $R/gpurun_out/pmc_convf_b $R/gpurun_out/pmc_convf_c
