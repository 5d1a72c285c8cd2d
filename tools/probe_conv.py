#!/usr/bin/env python
"""Times the fused conv kernel on the HiFiGAN-V1-44k stage shapes (B=32, 1 s clips) and prints TFLOP/s + GB/s.
Runs on the GPU box: python tools/probe_conv.py [--batch 32]"""
import argparse
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--precision", default="f32", choices=["f32", "f16x3"])
ap.add_argument("--quick", action="store_true", help="MFMA-bound stages, d=1 only, no transposed convs")
args = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
B = args.batch
stages = [(256, 688), (128, 5504), (64, 11008), (32, 22016), (16, 44032)]
if args.quick:
    stages = stages[:3]
print(f"{'shape':>28} {'kernel':>44} {'ms':>8} {'TFLOP/s':>8} {'GB/s':>8}")
tot_ms = 0.0
for C, T in stages:
    for k in (3, 7, 11):
        for d in ((1,) if args.quick else (1, 3, 5)):
            w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            conv = FusedConv(w, np.zeros(C, np.float32), dilation=d, padding=(k * d - d) // 2, pre_act=_lib.FV_ACT_SILU).set_precision(args.precision)
            x = torch.randn(B, C, T, device=dev)
            r = torch.randn(B, C, T, device=dev)
            y = torch.empty_like(x)
            for _ in range(3):
                conv(x, r, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                conv(x, r, y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            fl = 2.0 * C * C * k * T * B
            by = 3.0 * C * T * B * 4
            # the forward runs each (k, d=1) conv 3x as c2 plus one c1 per dilation
            weight = 1 if d != 1 else 4
            tot_ms += ms * weight
            print(f"{f'C={C} T={T} k={k} d={d}':>28} {_lib.last_kernel():>44} {ms:8.3f} {fl / ms / 1e9:8.1f} {by / ms / 1e6:8.0f}")
print(f"estimated ResBlock-path time per batch of {B}: {tot_ms:.2f} ms")
ups = [] if args.quick else [(512, 256, 16, 8, 86), (256, 128, 16, 8, 688), (128, 64, 8, 2, 5504), (64, 32, 2, 2, 11008), (32, 16, 2, 2, 22016)]
for cin, cout, k, u, T in ups:
    w = (rng.normal(size=(cin, cout, k)) / np.sqrt(cin)).astype(np.float32)
    conv = FusedConv(w, np.zeros(cout, np.float32), transposed=True, stride=u, padding=(k - u) // 2, pre_act=_lib.FV_ACT_SILU)
    x = torch.randn(B, cin, T, device=dev)
    y = conv(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        conv(x, None, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    fl = 2.0 * cin * cout * k * T * B
    by = (cin * T + cout * T * u) * B * 4.0
    print(f"{f'convT {cin}->{cout} k={k} u={u} T={T}':>28} {_lib.last_kernel():>44} {ms:8.3f} {fl / ms / 1e9:8.1f} {by / ms / 1e6:8.0f}")
