#!/usr/bin/env python
"""Time of the per-layer Winograd conv (conv_wino_impl.h) on the headline shapes, B = 32:  python tools/probe_wino_time.py [reps]
(FV_LIB_PATH selects an experimental build: A/B of kernel variants in interleaved runs)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
W = {3: 4 / 6, 7: 10 / 14, 11: 16 / 22}
for C, T in ((256, 688), (128, 5504), (64, 11008)):
    for k in (11, 7, 3):
        w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
        conv = FusedConv(w, np.zeros(C, np.float32), padding=(k - 1) // 2, pre_act=_lib.FV_ACT_SILU).set_algorithm("winograd")
        x = torch.randn(32, C, T, device="cuda"); r = torch.randn(32, C, T, device="cuda"); y = torch.empty_like(x)
        for _ in range(3): conv(x, r, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): conv(x, r, y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tf = 2.0 * C * C * k * T * 32 / ms / 1e9
        print(f"C={C} k={k}: {ms * 1e3:7.1f} us  {tf:6.1f} TF algorithmic  issued {tf * W[k] / 157.3:.2f}  {_lib.last_kernel()}")
