mkdir -p gpurun_out/k3
for r in 1 2; do
  python tools/probe_pair_wino.py 32 16 2>/dev/null | grep -v "k=3" | sed 's/^/tail1 /'
  FV_LIB_PATH=$PWD/vocoder_amd/csrc/libfishvoc_x_pqtail0.so python tools/probe_pair_wino.py 32 16 2>/dev/null | grep -v "k=3" | sed 's/^/tail0 /'
done | tee gpurun_out/k3/probe_pqtail.txt
