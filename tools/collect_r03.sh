# round-3 evidence: bash tools/collect_r03.sh <tag> [tests]   (run through gpurun from the repo root)
set -x
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$2" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -5 $O/pytest_gpu.log
fi
cd /tmp && export TMPDIR=/tmp
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (they cannot share one), kernel trace only
for m in hifigan bigvgan vocos; do
  B=$([ $m = hifigan ] && echo 32 || ([ $m = bigvgan ] && echo 64 || echo 128))
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${m}_$c -- python $R/tools/probe_model.py $m $B 2 > $O/pmc_${m}_$c.log 2>&1 || true
  done
  python $R/tools/pmc_summary.py $O/pmc_${m}_FETCH_SIZE $O/pmc_${m}_WRITE_SIZE 2 $O/${m}_hbm_traffic.json > $O/${m}_hbm_traffic.txt 2>&1
done
python $R/tools/pmc_traffic.py $O/pmc_hifigan_FETCH_SIZE $O/pmc_hifigan_WRITE_SIZE $O/traffic.json --build $TAG
# ... merged into profiles/traffic.json of this copy first, so that the bench line below finds its dominant kernel in it
python $R/tools/pmc_traffic.py $O/pmc_hifigan_FETCH_SIZE $O/pmc_hifigan_WRITE_SIZE $R/profiles/traffic.json --merge --build $TAG; cp $R/profiles/traffic.json $O/traffic_merged.json
cd $R
timeout 1200 python bench.py --profile-json $O/bench_kernels_hipevents.json > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
cd /tmp && export TMPDIR=/tmp
# the roofline section of bench.py itself under the profiler (single stream, every launch comparable): its hipEvent averages and rocprofv3's agree
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ro -- python $R/bench.py --roofline-only > $O/bench_roofline_only.json 2> /dev/null
cp $(ls -t $(find $O/prof_ro -name "*kernel_stats.csv") | head -1) $O/bench_roofline_only_kernel_stats.csv
python $R/tools/trace_stats_by_grid.py $O/prof_ro $O/bench_roofline_only_kernel_stats_by_grid.csv; rm -rf $O/prof_ro
for m in hifigan bigvgan vocos; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -- python $R/tools/probe_model.py $m > $O/prof_$m.log 2>&1
  cp $(find $O/prof_$m -name "*kernel_stats.csv" | head -1) $O/${m}_kernel_stats_serialized.csv
  rm -rf $O/prof_$m
done
# effective shader clock per kernel under the B = 32 headline step (GRBM_GUI_ACTIVE), and the matrix-pipe counters of the dominant kernel
cd $R
bash tools/pmc_clock.sh 32 > $O/clock_per_kernel.txt 2>&1
bash tools/pmc_conv.sh 128 5504 11 1 > $O/conv_pmc.txt 2>&1
./tools/ubench/mfma_mix > $O/ubench_mfma_mix.txt 2>&1
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE gpurun_out/pmc_conv_a
ls $O
