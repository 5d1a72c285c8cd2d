set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01e
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --profile-json $O/bench_kernels_hipevents.json --no-cpu-baseline --no-alt-precision > /dev/null 2>&1
python bench.py --precision f16x3 --no-cpu-baseline --profile-json $O/f16x3_bench_kernels_hipevents.json > $O/f16x3_bench.json 2>/dev/null
python tools/bench_models.py f32 > $O/other_models_f32.jsonl 2>/dev/null
python tools/bench_models.py f16x3 > $O/other_models_f16x3.jsonl 2>/dev/null
cd /tmp && export TMPDIR=/tmp
FV_SINGLE_STREAM=1 FV_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f16x3 -- python $R/bench.py --precision f16x3 --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_f16x3.log 2>&1
FV_SINGLE_STREAM=1 FV_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision > $O/prof_f32.log 2>&1
find $O -name "*kernel_stats.csv" | head
