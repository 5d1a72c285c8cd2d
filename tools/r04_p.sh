R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04p
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py -q -x -k "branch_mean or golden or batch_256" 2>&1 | tail -30
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_b1 -- python $R/tools/probe_latency.py > /dev/null 2>&1
python $R/tools/latency_timeline.py $O/trace_b1 47 > $O/b1_timeline_trio.txt 2>&1
rm -rf $O/trace_b1
cat $O/b1_timeline_trio.txt | cut -c1-150
