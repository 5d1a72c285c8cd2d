python -m pytest tests/test_gpu_conv.py -q -k "pair" 2>&1 | tail -3
python -m pytest tests/test_gpu_models.py -q -x -k "hifigan or narrow or template or fuzz" 2>&1 | tail -3
python tools/probe_pair_wino.py 2>/dev/null | tail -25
FV_LIB_PATH=$PWD/vocoder_amd/csrc/libfishvoc_x_nopair8.so python tools/probe_pair_wino.py 2>/dev/null | tail -25
bash tools/ab_libs.sh "x_nopair8 base" 3
