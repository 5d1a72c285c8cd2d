#!/usr/bin/env python
"""Times the pointwise (k=1) convs of the ConvNeXt trunks (Vocos-24k: B=128 x 94 frames; Firefly: B=32 x 86) on the general
conv kernel (FV_PW=old) and on every variant of the LDS-free GEMM kernel (gemm_pw.hip, FV_PW=<n>), and checks each variant
against the conv kernel's output:  python tools/probe_pointwise.py [variants, e.g. 0,1,2] [B] [T]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
variants = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else ["0", "1"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
T = int(sys.argv[3]) if len(sys.argv) > 3 else 94
rng = np.random.default_rng(0)
G, N = _lib.FV_ACT_GELU, _lib.FV_ACT_NONE
shapes = [(512, 2048, G, False), (2048, 512, N, True), (1024, 4096, G, False), (4096, 1024, N, True),
          (256, 1024, G, False), (1024, 256, N, True), (128, 512, G, False), (512, 128, N, True), (512, 1024, N, False)]


def timeit(conv, x, r, y, iters=20):
    for _ in range(3):
        conv(x, r, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        conv(x, r, y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ROUNDS = int(os.environ.get("FV_PROBE_ROUNDS", "5"))
for cin, cout, act, res in shapes:
    w = (rng.normal(size=(cout, cin, 1)) / np.sqrt(cin)).astype(np.float32)
    conv = FusedConv(w, rng.normal(size=cout).astype(np.float32), post_act=act)
    x = torch.randn(B, cin, T, device="cuda")
    r = torch.randn(B, cout, T, device="cuda") if res else None
    y = torch.empty(B, cout, T, device="cuda")
    flops = 2.0 * cin * cout * B * T
    os.environ["FV_PW"] = "old"
    _lib.reload_env()      # the library caches its knobs
    conv(x, r, y)
    ref = y.clone()
    names, errs, times = {}, {}, {v: [] for v in ["old"] + variants}
    # variants are timed round-robin, several rounds: the chip's clock follows its recent power draw, so a variant's time
    # depends on what ran just before it; medians over interleaved rounds are comparable, single passes are not
    for rnd in range(ROUNDS):
        for v in ["old"] + variants:
            os.environ["FV_PW"] = v
            _lib.reload_env()
            if rnd == 0:
                y.fill_(float("nan"))
                conv(x, r, y)
                errs[v] = float((y - ref).abs().max())
                names[v] = _lib.last_kernel()
            times[v].append(timeit(conv, x, r, y, iters=10))
    print(f"{cin:5d} -> {cout:5d} act={act} res={int(res)} B={B} T={T}   (median / min of {ROUNDS} interleaved rounds)")
    for v in ["old"] + variants:
        ms, mn = float(np.median(times[v])), float(np.min(times[v]))
        print(f"    {names[v]:>44} {ms:7.3f} ms {flops / ms / 1e9:7.1f} TFLOP/s  {flops / ms / 1e9 / 157.3:5.3f}   min {mn:7.3f} ms {flops / mn / 1e9 / 157.3:5.3f}   max|d| vs conv kernel {errs[v]:.2e}")
    sys.stdout.flush()
