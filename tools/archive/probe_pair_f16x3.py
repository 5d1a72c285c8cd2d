#!/usr/bin/env python
"""Times the fused f16x3 (c1, c2) pair kernel: python tools/probe_pair_f16x3.py [C,T,k,d ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
shapes = [tuple(map(int, s.split(","))) for s in sys.argv[1:]] or [(128, 5504, 11, 1), (128, 5504, 7, 1), (128, 5504, 3, 1), (64, 11008, 11, 1), (64, 11008, 3, 1)]
B = int(os.environ.get("PROBE_B", "32"))
rng = np.random.default_rng(0)
out = []
for C, T, k, d in shapes:
    w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    c1 = FusedConv(w, np.zeros(C, np.float32), dilation=d, padding=(k * d - d) // 2).set_precision("f16x3")
    c2 = FusedConv(w, np.zeros(C, np.float32), padding=(k - 1) // 2).set_precision("f16x3")
    x = torch.randn(B, C, T, device="cuda:0")
    for _ in range(3):
        c1.pair(c2, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c1.pair(c2, x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out.append(f"{C},{T},{k},{d}: {ms*1e3:7.1f} us {4.0*C*C*k*T*B/ms/1e9:6.1f} TF")
print(os.environ.get("FV_LIB_PATH", "default")[-22:], _lib.last_kernel(), " | ".join(out))
