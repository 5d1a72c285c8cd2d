R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
for v in main $VARIANTS main; do
  echo "== $v"
  if [ $v = main ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_$v.so; fi
  timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "pair" 2>&1 | tail -1
  timeout 600 python tools/probe_pair_wino.py $VC 2>&1 | grep "${VK:-k=3}" | sed 's/direct.*wino/wino/'
done
