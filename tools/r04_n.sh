R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_b1 -- python $R/tools/probe_latency.py > /dev/null 2>&1
python $R/tools/latency_timeline.py $O/trace_b1 > $O/b1_timeline.txt 2>&1
rm -rf $O/trace_b1
cat $O/b1_timeline.txt | cut -c1-140
