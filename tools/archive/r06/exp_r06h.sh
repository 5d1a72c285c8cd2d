set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
for r in 1 2; do
  python tools/probe_pair44.py inloop 32 16 >> $O/pair44_standalone.txt 2>&1
  FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_pqocc4.so python tools/probe_pair44.py occ4 32 >> $O/pair44_standalone.txt 2>&1
  FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_pq16.so python tools/probe_pair44.py inloop16 16 >> $O/pair44_standalone.txt 2>&1
done
grep -v amdgpu.ids $O/pair44_standalone.txt
bash tools/ab_libs.sh "x_pqocc4 x_pq16 base" 2 > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
