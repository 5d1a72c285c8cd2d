mkdir -p gpurun_out/k3
python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "pair" 2>&1 | tail -4 | tee gpurun_out/k3/tests.txt
for r in 1 2; do
  python tools/probe_pair_wino.py 32 16 2>/dev/null | grep "k=3\|k= 3\|^ *3 " | sed 's/^/reg  /'
  FV_LIB_PATH=$PWD/vocoder_amd/csrc/libfishvoc_x_k3lds.so python tools/probe_pair_wino.py 32 16 2>/dev/null | grep "k=3\|k= 3\|^ *3 " | sed 's/^/lds  /'
done | tee gpurun_out/k3/probe.txt
python tools/probe_pair_wino.py 16 2>/dev/null | tail -12
bash tools/ab_libs.sh "base x_k3lds" 4 2>&1 | tee gpurun_out/k3/ab_step.txt
