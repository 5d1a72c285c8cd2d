R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
for s in "16 44032 11 1" "16 44032 3 1" "32 22016 11 1" "32 22016 11 5"; do
  echo "== $s" >> $O/pair_wino_pmc.txt
  bash tools/pmc_pair.sh $s >> $O/pair_wino_pmc.txt 2>&1
done
rm -rf gpurun_out/pmc_pair_a gpurun_out/pmc_pair_b gpurun_out/pmc_pair_c
