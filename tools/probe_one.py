#!/usr/bin/env python
"""Runs ONE fused-conv shape a few times (for rocprofv3 --pmc passes): python tools/probe_one.py C T k d [B] [iters]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
C, T, k, d = map(int, sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
rng = np.random.default_rng(0)
w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
conv = FusedConv(w, np.zeros(C, np.float32), dilation=d, padding=(k * d - d) // 2)
x = torch.randn(B, C, T, device="cuda:0"); r = torch.randn(B, C, T, device="cuda:0"); y = torch.empty_like(x)
for _ in range(iters):
    conv(x, r, y)
torch.cuda.synchronize()
print(_lib.last_kernel())
