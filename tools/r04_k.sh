R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
bash tools/power_trace.sh 6 $O/power_trace.txt 2>&1 | tail -15
head -3 $O/power_trace.txt.samples
# aa_snake: is it VALU-issue bound?
cat tools/pmc_kernel.sh | head -20
