R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "pair" > $O/pytest_pair.log 2>&1; tail -3 $O/pytest_pair.log
timeout 600 python tools/probe_pair_wino.py 32 16 > $O/probe_main.txt 2>&1; tail -20 $O/probe_main.txt
for v in $VARIANTS; do
  echo "== variant $v"
  FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_$v.so timeout 600 python tools/probe_pair_wino.py $VC > $O/probe_$v.txt 2>&1; tail -${VT:-11} $O/probe_$v.txt
done
