set -x
TAG=r05z
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_hifigan_$c -- python $R/tools/probe_model.py hifigan 32 2 > $O/pmc_hifigan_$c.log 2>&1 || true
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_bigvgan_$c -- python $R/tools/probe_model.py bigvgan 64 2 > $O/pmc_bigvgan_$c.log 2>&1 || true
done
python $R/tools/pmc_summary.py $O/pmc_bigvgan_FETCH_SIZE $O/pmc_bigvgan_WRITE_SIZE 2 $O/bigvgan_hbm_traffic.json > $O/bigvgan_hbm_traffic.txt 2>&1
python $R/tools/pmc_traffic.py $O/pmc_hifigan_FETCH_SIZE $O/pmc_hifigan_WRITE_SIZE $R/profiles/traffic.json --merge --build $TAG; cp $R/profiles/traffic.json $O/traffic_merged.json
cd $R
timeout 1200 python bench.py --profile-json $O/bench_kernels_hipevents.json > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bigvgan -- python $R/tools/probe_model.py bigvgan > $O/prof_bigvgan.log 2>&1
cp $(find $O/prof_bigvgan -name "*kernel_stats.csv" | head -1) $O/bigvgan_kernel_stats_serialized.csv
rm -rf $O/prof_bigvgan $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
cd $R; python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
