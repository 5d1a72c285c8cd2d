set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06l; mkdir -p $O; cd $R
for r in 1 2; do
  python tools/probe_w44_ablation.py new >> $O/w44_standalone.txt 2>&1
  FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_w44old.so python tools/probe_w44_ablation.py old >> $O/w44_standalone.txt 2>&1
done
grep -v amdgpu.ids $O/w44_standalone.txt
bash tools/ab_libs.sh "x_w44old base" 3 > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
