python -m pytest tests/test_gpu_conv.py -q -k "f44 or heavy or 2_gib" 2>&1 | tail -3
python -m pytest tests/test_gpu_models.py -q -x -k "hifigan or winograd or golden" 2>&1 | tail -3
for r in 1 2; do
  FV_X_W44_NO_QR=1 python tools/probe_w44_ablation.py "no QR" 2>/dev/null
  python tools/probe_w44_ablation.py "QR" 2>/dev/null
done
bash tools/ab_env.sh FV_X_W44_NO_QR "1 0" 3
