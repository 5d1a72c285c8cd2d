R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_models.py -q -x -k "transpose or golden or convt or upsampl" 2>&1 | tail -2
bash tools/ab_env.sh FV_VEC_STORE "0 1" 3 2>&1 | tee $O/ab_vec_store.txt
