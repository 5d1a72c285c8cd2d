R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
for s in "128 5504 3 1" "64 11008 3 1"; do
  echo "== $s" >> $O/pair_wino_pmc.txt
  bash tools/pmc_pair.sh $s >> $O/pair_wino_pmc.txt 2>&1
done
rm -rf gpurun_out/pmc_pair_a gpurun_out/pmc_pair_b gpurun_out/pmc_pair_c
cat $O/pair_wino_pmc.txt
