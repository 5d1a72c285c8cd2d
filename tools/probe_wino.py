#!/usr/bin/env python
"""Winograd F(2,3) conv (conv_wino_impl.h, FV_WINO) against the direct MFMA conv on the same layer: largest deviation on ragged small
shapes and on the BASELINE shapes, and the time of both (hipEvents over `reps` launches).
    python tools/probe_wino.py [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vocoder_amd import _lib  # noqa: E402
from vocoder_amd.engine import FusedConv  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")


def run(conv, x, res, mode, n=1, cfg=-1):
    os.environ["FV_WINO"] = str(mode)
    os.environ["FV_WINO_MIN_M"] = "32"
    os.environ["FV_WINO_CFG"] = str(cfg)
    _lib.reload_env()
    y = conv(x, res)
    torch.cuda.synchronize()
    if n <= 1:
        return y, 0.0, _lib.last_kernel() if hasattr(_lib, "last_kernel") else ""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = torch.empty_like(y)
    e0.record()
    for _ in range(n):
        conv(x, res, out=out)
    e1.record()
    torch.cuda.synchronize()
    return y, e0.elapsed_time(e1) / n, ""


def case(cin, cout, k, d, B, T, timed, silu=True, with_res=True, cfg=-1):
    g = torch.Generator().manual_seed(cin * 131 + k * 7 + d + T)
    x = torch.randn(B, cin, T, generator=g).to(dev)
    w = (torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5)
    b = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, T, generator=g).to(dev) if with_res else None
    conv = FusedConv(w, b, dilation=d, padding=(k - 1) * d // 2, pre_act=_lib.FV_ACT_SILU if silu else _lib.FV_ACT_NONE)
    y0, t0, _ = run(conv, x, res, 0, reps if timed else 1)
    y1, t1, _ = run(conv, x, res, 2, reps if timed else 1, cfg)
    if timed:   # the second of two timed runs measures ~5 % faster (clock): time the direct kernel again, after the other
        _, t0b, _ = run(conv, x, res, 0, reps)
        t0 = min(t0, t0b)
    err = float((y0 - y1).abs().max())
    ref = torch.nn.functional.conv1d(torch.nn.functional.silu(x.double()) if silu else x.double(), w.double().to(dev), b.double().to(dev),
                                     padding=(k - 1) * d // 2, dilation=d)
    if res is not None:
        ref = ref + res.double()
    e0 = float((y0.double() - ref).abs().max())
    e1 = float((y1.double() - ref).abs().max())
    fl = 2.0 * cin * cout * k * T * B
    msg = f"cfg {cfg:2d} C {cin:3d}->{cout:3d} k={k:2d} d={d} B={B:2d} T={T:5d}: wino-direct {err:.2e}  vs fp64: direct {e0:.2e} wino {e1:.2e}"
    if timed:
        msg += f"   direct {t0 * 1e3:7.1f} us ({fl / t0 / 1e9:6.1f} TF)  wino {t1 * 1e3:7.1f} us ({fl / t1 / 1e9:6.1f} TF alg)  x{t0 / t1:.2f}"
    print(msg, flush=True)
    return err


worst = 0.0
SMALL = [(128, 128, 11, 1, 1, 517), (128, 128, 7, 3, 2, 300), (128, 128, 3, 5, 1, 131), (256, 256, 11, 5, 1, 97), (256, 256, 3, 1, 2, 200),
         (128, 192, 7, 5, 1, 1000), (64, 64, 11, 3, 2, 700), (64, 64, 7, 1, 1, 129), (64, 128, 3, 3, 1, 64), (40, 64, 7, 5, 1, 333),
         (128, 128, 11, 5, 1, 9), (128, 128, 7, 3, 1, 1), (32, 32, 11, 5, 2, 900), (32, 32, 7, 3, 1, 77), (64, 96, 11, 1, 1, 255)]
for cfg in ([] if os.environ.get('PROBE_TIMED_ONLY') else range(3)):
    for (cin, cout, k, d, B, T) in SMALL:
        worst = max(worst, case(cin, cout, k, d, B, T, False, cfg=cfg))
        worst = max(worst, case(cin, cout, k, d, B, T, False, silu=False, with_res=False, cfg=cfg))
print(f"worst wino-direct deviation on the small cases: {worst:.2e}")
for (c, T, B, cfgs) in [(128, 5504, 32, (0,)), (256, 688, 32, (0,)), (64, 11008, 32, (1,)), (32, 12032, 64, (2,))]:   # (3, 4 with -DFV_X_WINO_NT2)
    for k in (3, 7, 11):
        for d in (1, 5):
            for cfg in cfgs:
                case(c, c, k, d, B, T, True, cfg=cfg)
