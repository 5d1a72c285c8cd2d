#!/usr/bin/env python
"""conv_wino4_impl.h (Winograd F(4,3) tap groups) against conv_wino_impl.h (F(2,3)) on the headline shapes, B = 32: parity with the CPU oracle on a
ragged small case per (C, k, d), then time of both:  python tools/probe_wino4.py [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
from oracle import oracle as orc
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)


def setenv(v, rows=0):
    os.environ["FV_WINO4"] = str(v)
    os.environ["FV_WINO4_ROWS"] = str(rows)
    _lib.reload_env()


def timed(conv, x, r, y):
    for _ in range(3): conv(x, r, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): conv(x, r, y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot = [0.0, 0.0]
for C, T in ((256, 688), (128, 5504), (64, 11008)):
    for k in (11, 7, 3):
        for d in (1, 3, 5):
            w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            b = rng.normal(size=C).astype(np.float32)
            pad = (k - 1) // 2 * d
            conv = FusedConv(w, b, padding=pad, dilation=d, pre_act=_lib.FV_ACT_SILU).set_algorithm("winograd")
            xs = rng.normal(size=(2, C, 333)).astype(np.float32); rs = rng.normal(size=(2, C, 333)).astype(np.float32)
            ref = orc.conv1d(orc.silu(xs), w, b, dilation=d, padding=pad) + rs
            errs = []; names = []
            for v in (1, 0):
                setenv(v)
                ys = torch.empty(2, C, 333, device="cuda")
                conv(torch.from_numpy(xs).cuda(), torch.from_numpy(rs).cuda(), ys)
                errs.append(float(np.abs(ys.cpu().numpy() - ref).max() / np.abs(ref).max())); names.append(_lib.last_kernel())
            x = torch.randn(32, C, T, device="cuda"); r = torch.randn(32, C, T, device="cuda"); y = torch.empty_like(x)
            ms = []
            for v in (1, 0):
                setenv(v)
                ms.append(timed(conv, x, r, y))
            setenv(1, 64)
            ms64 = timed(conv, x, r, y)
            conv_na = FusedConv(w, b, padding=pad, dilation=d).set_algorithm("winograd")   # c2 of a pair: no activation in front
            setenv(1)
            ms_na = [timed(conv_na, x, r, y)]
            setenv(0)
            ms_na.append(timed(conv_na, x, r, y))
            w = 1 if (k != 3 or d == 1) else 0          # weight in the headline step: c2 (d = 1) x 3 + c1 d = 1 / 3 / 5
            n = (4 if d == 1 else 1) * (1 if (k != 3 or C == 256) else 0)
            tot[0] += n * ms[0]; tot[1] += n * ms[1]
            tf = 2.0 * C * C * k * T * 32 / ms[0] / 1e9
            print(f"C={C} k={k} d={d}: F(4,3) {ms[0] * 1e3:7.1f} us ({tf:6.1f} TF alg) err {errs[0]:.2e} | F(2,3) {ms[1] * 1e3:7.1f} us err {errs[1]:.2e} | ratio {ms[0] / ms[1]:.3f} | 64 rows {ms64 * 1e3:7.1f} | no act {ms_na[0] * 1e3:7.1f} / {ms_na[1] * 1e3:7.1f} = {ms_na[0] / ms_na[1]:.3f}  {names[0]}")
print(f"headline-weighted sum per step: F(4,3) {tot[0]:.3f} ms, F(2,3) {tot[1]:.3f} ms")
