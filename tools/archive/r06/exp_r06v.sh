set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06v; mkdir -p $O; cd $R
for r in 1 2; do
  { echo "== round $r: shipped (amp_conv at C <= 64, k = 3)"; python tools/ab_bigvgan.py 2; } >> $O/bigvgan_amp_fusion.txt 2>&1
  { echo "== round $r: FV_AMP_MAXC=32 (amp_conv at C = 32 only)"; FV_AMP_MAXC=32 python tools/ab_bigvgan.py 2; } >> $O/bigvgan_amp_fusion.txt 2>&1
  { echo "== round $r: FV_NO_AMP_FUSION=1 (aa_snake + conv everywhere)"; FV_NO_AMP_FUSION=1 python tools/ab_bigvgan.py 2; } >> $O/bigvgan_amp_fusion.txt 2>&1
done
grep -v amdgpu.ids $O/bigvgan_amp_fusion.txt
