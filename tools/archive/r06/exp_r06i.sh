set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_pers.so timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_models.py -m gpu -x -q > $O/pytest_pers.log 2>&1; tail -3 $O/pytest_pers.log
for r in 1 2; do
  python tools/probe_w44_ablation.py base >> $O/w44_standalone.txt 2>&1
  FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_pers.so python tools/probe_w44_ablation.py pers >> $O/w44_standalone.txt 2>&1
done
grep -v amdgpu.ids $O/w44_standalone.txt
bash tools/ab_libs.sh "x_pers base" 3 > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
