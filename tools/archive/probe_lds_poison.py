#!/usr/bin/env python
"""Runs every conv shape of a B=1 BigVGAN / HiFiGAN forward through fv_conv_forward, once after a clean start and several times
after an LDS-poisoning kernel (fv_debug_poison_lds): a kernel that reads LDS it never wrote changes its result.
python tools/probe_lds_poison.py [f32|f16x3] [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = _lib.lib()
L.fv_debug_poison_lds.argtypes = [ctypes.c_void_p]
rng = np.random.default_rng(0)
stages = [(256, 752), (128, 6016), (64, 12032), (32, 24064)]
bad_total = 0
for C, T in stages:
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            conv = FusedConv(w, rng.normal(size=C).astype(np.float32), dilation=d, padding=(k * d - d) // 2).set_precision(prec)
            x = torch.randn(B, C, T, device="cuda")
            r = torch.randn(B, C, T, device="cuda")
            y0 = conv(x, r).clone()
            torch.cuda.synchronize()
            bad = 0
            for _ in range(4):
                L.fv_debug_poison_lds(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                y = conv(x, r)
                torch.cuda.synchronize()
                if not torch.equal(y, y0):
                    bad += 1
            if bad:
                d_ = (y - y0)
                print(f"C={C} T={T} k={k} d={d} {_lib.last_kernel()}: {bad}/4 runs changed after LDS poison; finite={bool(torch.isfinite(y).all())} max|d|={float(torch.nan_to_num(d_).abs().max()):.3e}")
                bad_total += 1
# transposed convs of the upsampler
for cin, cout, k, u, T in [(512, 256, 16, 8, 94), (256, 128, 16, 8, 752), (128, 64, 4, 2, 6016), (64, 32, 4, 2, 12032)]:
    w = (rng.normal(size=(cin, cout, k)) / np.sqrt(cin * k / u)).astype(np.float32)
    conv = FusedConv(w, rng.normal(size=cout).astype(np.float32), transposed=True, stride=u, padding=(k - u) // 2).set_precision(prec)
    x = torch.randn(B, cin, T, device="cuda")
    y0 = conv(x).clone()
    torch.cuda.synchronize()
    bad = 0
    for _ in range(4):
        L.fv_debug_poison_lds(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        y = conv(x)
        torch.cuda.synchronize()
        bad += int(not torch.equal(y, y0))
    if bad:
        print(f"convT {cin}->{cout} k={k} u={u} T={T} {_lib.last_kernel()}: {bad}/4 changed; finite={bool(torch.isfinite(y).all())}")
        bad_total += 1
print(f"{prec} B={B}: {bad_total} shapes changed under LDS poisoning")
