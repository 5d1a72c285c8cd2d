#!/usr/bin/env python
"""What the SiLU in front of a conv_wino44 launch costs it: the c1 form (SiLU in front and behind, no residual) against the same launch without the
activation in front (PRE = 0 instance), back to back, us per launch.   python tools/probe_w44_silu_cost.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv

rng = np.random.default_rng(0)


def timed(conv, x, y, reps=30):
    for _ in range(4):
        conv(x, None, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        conv(x, None, y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rnd in range(2):
    cells = []
    for C, T, k, d in ((128, 5504, 11, 1), (128, 5504, 11, 3), (128, 5504, 7, 1), (128, 5504, 7, 5), (256, 688, 11, 1), (256, 688, 7, 1), (64, 11008, 11, 1), (64, 11008, 7, 1)):
        w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
        b = rng.normal(size=C).astype(np.float32)
        pad = (k - 1) // 2 * d
        c_silu = FusedConv(w, b, padding=pad, dilation=d, pre_act=_lib.FV_ACT_SILU, post_act=_lib.FV_ACT_SILU).set_algorithm("winograd")
        c_none = FusedConv(w, b, padding=pad, dilation=d, pre_act=_lib.FV_ACT_NONE, post_act=_lib.FV_ACT_SILU).set_algorithm("winograd")
        x = torch.randn(32, C, T, device="cuda")
        y = torch.empty_like(x)
        a, n = timed(c_silu, x, y), timed(c_none, x, y)
        cells.append(f"C={C} k={k} d={d}: {a:6.1f} / {n:6.1f} ({(a / n - 1) * 100:+.1f} %)")
    print(f"round {rnd}: SiLU in front / none | " + " | ".join(cells))
