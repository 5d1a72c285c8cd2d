R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_modules.py tests/test_gpu_f16x3.py -q 2>&1 | tail -2
for v in 1 2 3; do python tools/probe_latency.py 2>&1 | grep "p50"; done
TOP=60 python tools/probe_latency.py 2>&1 | grep "conv_post"
