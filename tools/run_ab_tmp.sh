FV_PAIR_WINO44=2 python -m pytest tests/test_gpu_conv.py -q -x -k "winograd_pair_matches and 3-" 2>&1 | tail -3
python tools/probe_pair_wino.py 32 16 2>/dev/null | grep "k=3"
FV_PAIR_WINO44=2 python tools/probe_pair_wino.py 32 16 2>/dev/null | grep "k=3"
bash tools/ab_env.sh FV_PAIR_WINO44 "1 2" 3
