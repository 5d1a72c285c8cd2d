R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "pair" > $O/pytest_pair.log 2>&1; tail -5 $O/pytest_pair.log
timeout 600 python tools/probe_pair_wino.py 32 16 > $O/probe_pair_wino.txt 2>&1; tail -30 $O/probe_pair_wino.txt
