#!/usr/bin/env python
"""conv_wino44 on the headline's MFMA-bound layers, one process per library variant (FV_LIB_PATH): us per launch back to back in the c1 form
(SiLU in front and behind) and the c2 form (no activation, residual).  Used with the timing ablations of conv_wino44_impl.h
(-DFV_X_W44_ABL=<mask>: WRONG results, only the time is meaningful): what each part of the staging costs the launch (LOG R5.1).
  python tools/probe_w44_ablation.py [label]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv

label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("FV_LIB_PATH", "shipped"))
rng = np.random.default_rng(0)
reps = 30


def timed(conv, x, r, y):
    for _ in range(4):
        conv(x, r, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        conv(x, r, y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cells = []
for C, T, k, d in ((128, 5504, 11, 1), (128, 5504, 7, 1), (256, 688, 11, 1), (64, 11008, 11, 1), (128, 5504, 11, 5)):
    w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    b = rng.normal(size=C).astype(np.float32)
    pad = (k - 1) // 2 * d
    c1 = FusedConv(w, b, padding=pad, dilation=d, pre_act=_lib.FV_ACT_SILU, post_act=_lib.FV_ACT_SILU).set_algorithm("winograd")
    c2 = FusedConv(w, b, padding=pad, dilation=d).set_algorithm("winograd")
    x = torch.randn(32, C, T, device="cuda")
    r = torch.randn(32, C, T, device="cuda")
    y = torch.empty_like(x)
    t1 = timed(c1, x, None, y)
    kern = _lib.last_kernel()
    t2 = timed(c2, x, r, y)
    cells.append(f"C={C} k={k} d={d}: c1 {t1:6.1f} c2 {t2:6.1f}")
    assert kern.startswith("conv_wino44<"), kern
print(f"{label:>14s} | " + " | ".join(cells))
