for r in 1 2 3; do
for v in x_old base; do
  if [ $v = base ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$PWD/vocoder_amd/csrc/libfishvoc_$v.so; fi
  echo "$v: $(python tools/ab_bigvgan.py 2 2>/dev/null | tail -3 | tr '\n' ' ')"
done; done
python -m pytest tests -m gpu -q -k "bigvgan or snake or fuzz" 2>&1 | tail -2
