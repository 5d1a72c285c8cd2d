set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R
for r in 1 2; do
  FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_prev.so python tools/probe_w44_ablation.py prev >> $O/w44_standalone.txt 2>&1
  FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_phase.so python tools/probe_w44_ablation.py phase+epi >> $O/w44_standalone.txt 2>&1
  python tools/probe_w44_ablation.py inloop+epi >> $O/w44_standalone.txt 2>&1
done
bash tools/ab_libs.sh "x_prev x_phase base" 3 > $O/ab_step.txt 2>&1
grep -v amdgpu.ids $O/w44_standalone.txt; cat $O/ab_step.txt
