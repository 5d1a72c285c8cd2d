mkdir -p gpurun_out/aa
for r in 1 2 3; do
  python tools/ab_bigvgan.py 2 2>/dev/null | grep -E "round|serialized" | sed 's/^/full    /'
  FV_X_ABL_AA_SNAKE=1 python tools/ab_bigvgan.py 2 2>/dev/null | grep -E "round|serialized" | sed 's/^/no-aa   /'
done | tee gpurun_out/aa/ablation.txt
