R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "pair" > $O/pytest_pair.log 2>&1; tail -3 $O/pytest_pair.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "ragged_shards or batch_256 or round3_fusions or winograd_convs" > $O/pytest_models.log 2>&1; tail -5 $O/pytest_models.log
timeout 600 python tools/probe_pair_wino.py 128 64 32 16 > $O/probe_main.txt 2>&1; tail -40 $O/probe_main.txt
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*" | sort -u > $O/tcc_counters.txt)
