#!/usr/bin/env python
"""One fused-pair shape a few times (for rocprofv3 --pmc): python tools/probe_pair_one.py C T k d"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd.engine import FusedConv
C, T, k, d = map(int, sys.argv[1:5])
rng = np.random.default_rng(0)
w1 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
c1 = FusedConv(w1, np.zeros(C, np.float32), dilation=d, padding=(k * d - d) // 2)
c2 = FusedConv(w1, np.zeros(C, np.float32), padding=(k - 1) // 2)
x = torch.randn(32, C, T, device="cuda")
for _ in range(4):
    c1.pair(c2, x)
torch.cuda.synchronize()
