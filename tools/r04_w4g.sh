mkdir -p gpurun_out/r04v
R=$(pwd)
{
for v in main occ4 prio main occ4 prio; do
  if [ $v = main ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['avg_ms'])"
done
} > gpurun_out/r04v/ab_occ_prio.txt 2>&1
cat gpurun_out/r04v/ab_occ_prio.txt
