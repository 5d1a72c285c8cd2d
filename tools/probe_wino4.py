#!/usr/bin/env python
"""The per-layer Winograd conv kernels side by side on the headline shapes, B = 32 — conv_wino44_impl.h (F(4,4) tap groups), conv_wino4_impl.h (F(4,3),
FV_WINO44=0) and conv_wino_impl.h (F(2,3), FV_WINO4=0): deviation from the CPU oracle on a ragged small case per (C, k, d), then µs per launch back to
back, with the SiLU in front (c1 of a ResBlock pair) and without (c2):  python tools/probe_wino4.py [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
from oracle import oracle as orc
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
FORMS = (("F(4,4)", {"FV_WINO4": "1", "FV_WINO44": "1"}), ("F(4,3)", {"FV_WINO4": "1", "FV_WINO44": "0"}), ("F(2,3)", {"FV_WINO4": "0", "FV_WINO44": "0"}))


def setenv(env):
    os.environ.update(env)
    _lib.reload_env()


def timed(conv, x, r, y):
    for _ in range(3): conv(x, r, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): conv(x, r, y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot = [0.0] * len(FORMS)
for C, T in ((256, 688), (128, 5504), (64, 11008)):
    for k in (11, 7):
        for d in (1, 3, 5):
            w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            b = rng.normal(size=C).astype(np.float32)
            pad = (k - 1) // 2 * d
            xs = rng.normal(size=(2, C, 333)).astype(np.float32); rs = rng.normal(size=(2, C, 333)).astype(np.float32)
            ref = orc.conv1d(orc.silu(xs), w, b, dilation=d, padding=pad) + rs
            x = torch.randn(32, C, T, device="cuda"); r = torch.randn(32, C, T, device="cuda"); y = torch.empty_like(x)
            cells = []
            for fi, (name, env) in enumerate(FORMS):
                setenv(env)   # (before the layers are created: a layer packs only the Winograd form its knobs select — conv_layer.hip)
                conv = FusedConv(w, b, padding=pad, dilation=d, pre_act=_lib.FV_ACT_SILU).set_algorithm("winograd")
                conv_na = FusedConv(w, b, padding=pad, dilation=d).set_algorithm("winograd")   # c2 of a pair: no activation in front
                ys = torch.empty(2, C, 333, device="cuda")
                conv(torch.from_numpy(xs).cuda(), torch.from_numpy(rs).cuda(), ys)
                err = float(np.abs(ys.cpu().numpy() - ref).max() / np.abs(ref).max())
                kern = _lib.last_kernel().split("<")[0]
                ms, ms_na = timed(conv, x, r, y), timed(conv_na, x, r, y)
                n = 3 if d == 1 else 1              # launches per step: c1 (SiLU in front) of this dilation; + c2 (d = 1, no activation) x 3
                tot[fi] += (n if d != 1 else 1) * ms + (3 * ms_na if d == 1 else 0.0)
                cells.append(f"{name} {kern:11s} {ms * 1e3:6.1f} / {ms_na * 1e3:6.1f} us err {err:.1e}")
            print(f"C={C:3d} k={k:2d} d={d}: " + " | ".join(cells))
print("headline-weighted sum per step (c1 of each dilation + three c2): " + ", ".join(f"{n} {t:.3f} ms" for (n, _), t in zip(FORMS, tot)))
