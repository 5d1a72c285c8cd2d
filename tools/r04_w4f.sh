mkdir -p gpurun_out/r04v
R=$(pwd)
{
for s in "128 11 5504 1" "128 7 5504 1" "256 11 688 1" "64 11 11008 1"; do
  echo "== $s F(4,3)"; FV_WINO=2 FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_ts.so timeout 120 python tools/probe_conv_timeline.py $s 2>&1 | grep -v amdgpu.ids
  echo "== $s F(2,3)"; FV_WINO=2 FV_WINO4=0 FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_ts.so timeout 120 python tools/probe_conv_timeline.py $s 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r04v/wino4_phase_timeline.txt 2>&1
{
for v in main da2 da4 main da2 da4; do
  echo "== $v"
  if [ $v = main ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_$v.so; fi
  timeout 200 python tools/probe_wino_time.py 2>&1 | grep "C=" | grep -v "k=3"
done
} > gpurun_out/r04v/ab_da.txt 2>&1
cat gpurun_out/r04v/wino4_phase_timeline.txt; cat gpurun_out/r04v/ab_da.txt
