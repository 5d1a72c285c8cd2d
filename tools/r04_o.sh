R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1200 python -m pytest tests/test_gpu_models.py -q -x 2>&1 | tail -2
for v in "FV_NO_TRIO=1" "FV_X=0" "FV_NO_TRIO=1" "FV_X=0" "FV_WINO_LAT=11" "FV_WINO_LAT=10"; do echo $v; env $v python tools/probe_latency.py 2>&1 | grep "p50"; done
