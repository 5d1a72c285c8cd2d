# round-2 evidence: bash tools/collect_r02.sh <tag> [tests]   (run through gpurun from the repo root)
set -x
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$2" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -5 $O/pytest_gpu.log
fi
timeout 900 python bench.py --profile-json $O/bench_kernels_hipevents.json > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
cd /tmp && export TMPDIR=/tmp
for m in hifigan bigvgan vocos; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -- python $R/tools/probe_model.py $m > $O/prof_$m.log 2>&1
  cp $(find $O/prof_$m -name "*kernel_stats.csv" | head -1) $O/${m}_kernel_stats_serialized.csv
  rm -rf $O/prof_$m
done
for m in bigvgan vocos; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${m}_$c -- python $R/tools/probe_model.py $m $([ $m = bigvgan ] && echo 64 || echo 128) 2 > $O/pmc_${m}_$c.log 2>&1 || true
  done
done
ls $O
