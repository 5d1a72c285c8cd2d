#!/usr/bin/env python
"""Times the fused ResBlock pair kernels at the HiFiGAN-V1 stage-3/4 shapes (B=32)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
rng = np.random.default_rng(0)
B = 32
tot = 0.0
for C, T in ((128, 5504), (64, 11008), (32, 22016), (16, 44032)):
    for k in ((3,) if C >= 64 else (3, 7, 11)):
        for d in (1, 3, 5):
            w1 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            c1 = FusedConv(w1, np.zeros(C, np.float32), dilation=d, padding=(k * d - d) // 2)
            c2 = FusedConv(w1, np.zeros(C, np.float32), padding=(k - 1) // 2)
            x = torch.randn(B, C, T, device="cuda")
            for _ in range(3):
                c1.pair(c2, x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                c1.pair(c2, x)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            tot += ms
            print(f"C={C} k={k} d={d}: {ms:.3f} ms  {4.0 * C * C * k * T * B / ms / 1e9:6.1f} TF")
print(f"sum {tot:.2f} ms")
