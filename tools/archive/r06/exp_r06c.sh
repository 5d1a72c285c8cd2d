# round-6 experiment c: conv_wino44 with the staging inside the matrix loop + per-tile upfront epilogue operands, against the previous build
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "winograd or wino or f44 or flat" > $O/pytest_wino.log 2>&1; tail -3 $O/pytest_wino.log
for r in 1 2; do
  FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_prev.so python tools/probe_w44_ablation.py prev >> $O/w44_standalone.txt 2>&1
  python tools/probe_w44_ablation.py inloop >> $O/w44_standalone.txt 2>&1
done
bash tools/ab_libs.sh "x_prev base" 3 > $O/ab_step.txt 2>&1
grep -v amdgpu.ids $O/w44_standalone.txt; cat $O/ab_step.txt
