mkdir -p gpurun_out/k3
python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "pair" 2>&1 | tail -2
for r in 1 2; do
  python tools/probe_pair_wino.py 32 16 2>/dev/null | grep -v "k=3" | sed 's/^/pk1 /'
  FV_LIB_PATH=$PWD/vocoder_amd/csrc/libfishvoc_x_pqpk0.so python tools/probe_pair_wino.py 32 16 2>/dev/null | grep -v "k=3" | sed 's/^/pk0 /'
done > gpurun_out/k3/probe_pqpk.txt
awk '{print $1,$2,$3,$4,$13,$14}' gpurun_out/k3/probe_pqpk.txt | grep -v sum | sort -k2,4 -s | paste - - - - | awk '{printf "%s %s %s  pk1 %s %s  pk0 %s %s\n",$2,$3,$4,$5,$17,$11,$23}'
grep sum gpurun_out/k3/probe_pqpk.txt
