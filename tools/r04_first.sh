# round 4, first GPU call: baseline stand-alone pair timings + SQ counters of the pair kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
python tools/probe_pair.py > $O/probe_pair.txt 2>&1
for s in "16 44032 3 1" "16 44032 11 1" "32 22016 3 1" "32 22016 11 1" "128 5504 3 1"; do
  echo "== $s" >> $O/pair_pmc.txt
  bash tools/pmc_pair.sh $s >> $O/pair_pmc.txt 2>&1
done
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counters.txt)
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err
rm -rf gpurun_out/pmc_pair_a gpurun_out/pmc_pair_b gpurun_out/pmc_pair_c
