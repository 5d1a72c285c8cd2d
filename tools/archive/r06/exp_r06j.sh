set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R
S=$R/vocoder_amd/csrc/libfishvoc_x_surr.so
for r in 1 2; do
  { echo "== round $r: shipped"; python tools/ab_bigvgan.py 2; } >> $O/bigvgan_surrogate.txt 2>&1
  { echo "== round $r: aa_snake skipped in front of the k = 7 / 11 convs at C = 64 / 128 (FV_X_ABL_AA_SNAKE=2; wrong results): the ceiling of fusing those 24 launches"; FV_X_ABL_AA_SNAKE=2 python tools/ab_bigvgan.py 2; } >> $O/bigvgan_surrogate.txt 2>&1
  { echo "== round $r: the same + those convs carry the activation's vector issue in their staging (-DFV_X_W44_SURR=38): lower bound of a fused step"; FV_LIB_PATH=$S FV_X_ABL_AA_SNAKE=2 python tools/ab_bigvgan.py 2; } >> $O/bigvgan_surrogate.txt 2>&1
done
{ echo "== every aa_snake in front of a k = 7 / 11 conv skipped (FV_X_ABL_AA_SNAKE=1)"; FV_X_ABL_AA_SNAKE=1 python tools/ab_bigvgan.py 2; } >> $O/bigvgan_surrogate.txt 2>&1
grep -v amdgpu.ids $O/bigvgan_surrogate.txt
