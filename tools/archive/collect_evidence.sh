set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01f
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
python $R/tools/trace_by_grid.py f32=$O/prof_f32 f16x3=$O/prof_f16x3 > $O/kernel_trace_by_grid.json
cat $O/other_models_f32.jsonl $O/other_models_f16x3.jsonl > $O/other_models.jsonl
for p in f32 f16x3; do cp $(find $O/prof_$p -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serialized_$p.csv; done
rm -rf $O/prof_f32 $O/prof_f16x3
