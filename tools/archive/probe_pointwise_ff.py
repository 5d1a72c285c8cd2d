#!/usr/bin/env python
"""Pointwise convs of the Firefly-GAN base backbone (B=32 x 86 frames): python tools/probe_pointwise_ff.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
B, T = 32, 86
rng = np.random.default_rng(0)
tot = 0.0
for cin, cout, n in [(128, 512, 3), (512, 128, 3), (256, 1024, 3), (1024, 256, 3), (384, 1536, 9), (1536, 384, 9), (512, 2048, 3), (2048, 512, 3)]:
    w = (rng.normal(size=(cout, cin, 1)) / np.sqrt(cin)).astype(np.float32)
    conv = FusedConv(w, np.zeros(cout, np.float32))
    x = torch.randn(B, cin, T, device="cuda"); y = torch.empty(B, cout, T, device="cuda")
    for _ in range(3): conv(x, None, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): conv(x, None, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tot += ms * n
    print(f"{cin:5d}->{cout:5d} {_lib.last_kernel():>40} {ms*1e3:7.1f} us {2.0*cin*cout*B*T/ms/1e9:6.1f} TF")
print("backbone pointwise total ms", tot)
