#!/usr/bin/env python
"""Fused (c1, c2) pair kernels: Winograd (pair_wino_impl.h) against the direct-sum ones (resblock_pair.hip) at the HiFiGAN-V1 stage
shapes (B = 32): deviation of each from a float64 torch reference (2 clips) and time per launch.
  python tools/probe_pair_wino.py [C ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, torch.nn.functional as F
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv

L = _lib.lib()
rng = np.random.default_rng(0)
B = 32
shapes = {128: 5504, 64: 11008, 32: 22016, 16: 44032}
want = [int(a) for a in sys.argv[1:]] or [32, 16]


def set_mode(m):
    os.environ["FV_PAIR_WINO"] = str(m)
    L.fv_reload_env()


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = {0: 0.0, 1: 0.0}
for C in want:
    T = shapes[C]
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            w1 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            w2 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            b1 = rng.normal(size=C).astype(np.float32)
            b2 = rng.normal(size=C).astype(np.float32)
            c1 = FusedConv(w1, b1, dilation=d, padding=(k * d - d) // 2)
            c2 = FusedConv(w2, b2, padding=(k - 1) // 2)
            x = torch.randn(B, C, T, device="cuda")
            xs = x[:2].double()
            xt = F.conv1d(F.silu(xs), torch.from_numpy(w1).cuda().double(), torch.from_numpy(b1).cuda().double(), dilation=d, padding=(k * d - d) // 2)
            ref = xs + F.conv1d(F.silu(xt), torch.from_numpy(w2).cuda().double(), torch.from_numpy(b2).cuda().double(), padding=(k - 1) // 2)
            out = {}
            for mode in (0, 1):
                set_mode(mode)
                try:
                    y = c1.pair(c2, x)
                except Exception as e:   # no kernel of that kind for this width
                    out[mode] = None
                    continue
                torch.cuda.synchronize()
                err = float((y[:2].double() - ref).abs().max())
                ms = timed(lambda: c1.pair(c2, x))
                out[mode] = (err, ms, _lib.last_kernel() if hasattr(_lib, "last_kernel") else "")
                tot[mode] += ms
            line = f"C={C} k={k} d={d}:"
            for mode in (0, 1):
                if out[mode]:
                    err, ms, _ = out[mode]
                    line += f"  {'wino' if mode else 'direct'} {ms:.4f} ms {4.0 * C * C * k * T * B / ms / 1e9:6.1f} TF err {err:.2e}"
            if out[0] and out[1]:
                line += f"  ratio {out[1][1] / out[0][1]:.3f}"
            print(line, flush=True)
print(f"sum direct {tot[0]:.3f} ms, wino {tot[1]:.3f} ms")
