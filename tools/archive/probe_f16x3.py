#!/usr/bin/env python
"""Times the f16x3 conv kernel on a few HiFiGAN-V1 stage shapes (B=32): python tools/probe_f16x3.py [shape ...]
Each shape is C,T,k,d.  FV_LIB_PATH selects an experiment build of the library."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
shapes = [tuple(map(int, s.split(","))) for s in sys.argv[1:]] or [(128, 5504, 11, 1), (128, 5504, 7, 1), (128, 5504, 3, 1),
                                                                    (64, 11008, 11, 1), (64, 11008, 3, 1), (256, 688, 11, 1), (256, 688, 3, 1)]
B = int(os.environ.get("PROBE_B", "32"))
rng = np.random.default_rng(0)
out = []
for C, T, k, d in shapes:
    w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    conv = FusedConv(w, np.zeros(C, np.float32), dilation=d, padding=(k * d - d) // 2, pre_act=_lib.FV_ACT_SILU).set_precision("f16x3")
    x = torch.randn(B, C, T, device="cuda:0"); r = torch.randn(B, C, T, device="cuda:0"); y = torch.empty_like(x)
    for _ in range(3):
        conv(x, r, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        conv(x, r, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out.append(f"{C},{T},{k},{d}: {ms*1e3:7.1f} us {2.0*C*C*k*T*B/ms/1e9:6.1f} TF [{_lib.last_kernel()}]")
print(os.environ.get("FV_LIB_PATH", "default"), " | ".join(out))
