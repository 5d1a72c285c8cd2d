R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04m
mkdir -p $O
cd $R
export FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_ts.so
for a in "256 11 688 1 1" "128 11 5504 1 1" "64 11 11008 1 1" "256 3 688 1 1"; do
  python tools/probe_conv_timeline.py $a res 2>&1 | grep -v amdgpu.ids | tee -a $O/lat_timeline.txt
done
