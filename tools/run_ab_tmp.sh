python -m pytest tests/test_gpu_conv.py -q -x -s -k "pairs_with_heavy" 2>&1 | grep "heavy\|passed\|failed"
python - <<'PY'
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
rng=np.random.default_rng(0)
w=(rng.normal(size=(32,32,11))/19).astype(np.float32); b=np.zeros(32,np.float32)
c1=FusedConv(w,b,dilation=3,padding=15); c2=FusedConv(w,b,padding=5)
x=torch.randn(2,32,3000,device='cuda'); c1.pair(c2,x); torch.cuda.synchronize(); print('kernel:', _lib.last_kernel())
PY
FV_PAIR_WINO44=0 python tools/probe_pair_wino.py 2>/dev/null | grep -v "k=3" | tail -13
python tools/probe_pair_wino.py 2>/dev/null | grep -v "k=3" | tail -13
bash tools/ab_env.sh FV_PAIR_WINO44 "0 1" 3
