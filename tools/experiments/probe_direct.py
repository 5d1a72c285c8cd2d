#!/usr/bin/env python
"""A/B of the LDS-free direct conv kernel (conv_direct.hip, FV_DIRECT=1) against the tiled LDS kernel (FV_DIRECT=0) on the
MFMA-bound ResBlock shapes (no pre-activation: the direct kernel reads an already-activated tensor), interleaved rounds,
medians; checks that the two agree bit for bit (same summation order).   python tools/probe_direct.py [B] [hifigan|bigvgan]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = sys.argv[2] if len(sys.argv) > 2 else "hifigan"
stages = [(256, 688), (128, 5504), (64, 11008)] if model == "hifigan" else [(256, 752), (128, 6016), (64, 12032)]
ROUNDS = int(os.environ.get("FV_PROBE_ROUNDS", "5"))
rng = np.random.default_rng(0)


def timeit(f, iters=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot = {"0": 0.0, "1": 0.0}
for C, T in stages:
    for k in (3, 7, 11):
        for d in (1, 5):
            w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            conv = FusedConv(w, rng.normal(size=C).astype(np.float32), dilation=d, padding=(k * d - d) // 2)
            x = torch.randn(B, C, T, device="cuda")
            r = torch.randn(B, C, T, device="cuda")
            ys, names, times = {}, {}, {"0": [], "1": []}
            for rnd in range(ROUNDS):
                for v in ("0", "1"):
                    os.environ["FV_DIRECT"] = v
                    _lib.reload_env()
                    if rnd == 0:
                        y = torch.full_like(x, float("nan"))
                        conv(x, r, y)
                        ys[v] = y
                        names[v] = _lib.last_kernel()
                    yy = ys[v]
                    times[v].append(timeit(lambda: conv(x, r, yy)))
            fl = 2.0 * C * C * k * T * B
            m0, m1 = np.median(times["0"]), np.median(times["1"])
            w8 = 4 if d == 1 else 2          # a forward runs each (k, d = 1) conv 4x (3 c2 + 1 c1), the d = 3 / 5 c1 once each
            tot["0"] += m0 * w8; tot["1"] += m1 * w8
            same = bool(torch.equal(ys["0"], ys["1"]))
            print(f"C={C:4d} T={T:6d} k={k:2d} d={d}  tiled {m0:7.3f} ms {fl / m0 / 1e9:6.1f} TF ({names['0']})   direct {m1:7.3f} ms "
                  f"{fl / m1 / 1e9:6.1f} TF ({names['1']})   x{m0 / m1:5.3f}  identical={same}  maxdiff={float((ys['0'] - ys['1']).abs().max()):.2e}")
print(f"weighted sum: tiled {tot['0']:.2f} ms, direct {tot['1']:.2f} ms")
# ragged lengths / alignment: odd T, T not a multiple of 64, single item
for C, T, k, d, b in ((64, 1001, 7, 3, 3), (128, 777, 11, 5, 2), (64, 130, 3, 1, 5), (64, 64, 11, 1, 2), (128, 63, 7, 1, 2)):
    w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    conv = FusedConv(w, rng.normal(size=C).astype(np.float32), dilation=d, padding=(k * d - d) // 2)
    x = torch.randn(b, C, T, device="cuda"); r = torch.randn(b, C, T, device="cuda")
    out = {}
    for v in ("0", "1"):
        os.environ["FV_DIRECT"] = v
        _lib.reload_env()
        y = torch.full((b, C, T + 8), float("nan"), device="cuda")[:, :, :T]
        yc = torch.full_like(x, float("nan"))
        conv(x, r, yc)
        out[v] = (yc, _lib.last_kernel())
    print(f"ragged C={C} T={T} k={k} d={d} B={b}: {out['1'][1]} identical={bool(torch.equal(out['0'][0], out['1'][0]))} "
          f"finite={bool(torch.isfinite(out['1'][0]).all())}")
