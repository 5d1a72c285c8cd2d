R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "latency or winograd" 2>&1 | tail -3
for v in 10 11 12; do FV_WINO_LAT=$v timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "latency" 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_models.py -q -x 2>&1 | tail -2
for v in 0 1 10 11 12 1; do echo "FV_WINO_LAT=$v"; FV_WINO_LAT=$v python tools/probe_latency.py 2>&1 | grep "p50\|serialized"; done
FV_WINO_LAT=1 python tools/probe_latency.py > $O/latency_b1.txt 2>&1; head -30 $O/latency_b1.txt | cut -c1-140
