#!/bin/bash
# Two PMC passes (FETCH_SIZE, WRITE_SIZE cannot share one) over single-stream forwards: bash tools/pmc_collect.sh [f32|f16x3]
PREC=${1:-f32}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${PREC}_$c -- python $R/tools/probe_forward.py 32 2 $PREC > $R/gpurun_out/pmc_${PREC}_$c.log 2>&1
done
ls $R/gpurun_out/pmc_${PREC}_FETCH_SIZE/*/ | head -3
