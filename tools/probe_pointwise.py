#!/usr/bin/env python
"""Times the pointwise (k=1) convs of the Vocos-24k ConvNeXt trunk: B=128 clips x 94 frames (python tools/probe_pointwise.py [f32|f16x3])."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
B, T = 128, 94
rng = np.random.default_rng(0)
for cin, cout, act, res in [(512, 2048, _lib.FV_ACT_GELU, False), (2048, 512, _lib.FV_ACT_NONE, True),
                            (1024, 4096, _lib.FV_ACT_GELU, False), (4096, 1024, _lib.FV_ACT_NONE, True),
                            (512, 2048, _lib.FV_ACT_NONE, False)]:
    w = (rng.normal(size=(cout, cin, 1)) / np.sqrt(cin)).astype(np.float32)
    conv = FusedConv(w, np.zeros(cout, np.float32), post_act=act).set_precision(prec)
    x = torch.randn(B, cin, T, device="cuda")
    r = torch.randn(B, cout, T, device="cuda") if res else None
    y = torch.empty(B, cout, T, device="cuda")
    for _ in range(3):
        conv(x, r, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        conv(x, r, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{cin:5d} -> {cout:5d} act={act} res={int(res)} {_lib.last_kernel():>40} {ms:7.3f} ms {2.0 * cin * cout * B * T / ms / 1e9:7.1f} TFLOP/s")
