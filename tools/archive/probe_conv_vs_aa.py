#!/usr/bin/env python
"""Every conv shape of a small-batch BigVGAN forward (f16x3 / f32) while aa_snake launches run on two other streams: a kernel
that consumes uninitialised registers / LDS changes its result depending on what else ran on its CU.
python tools/probe_conv_vs_aa.py [f16x3|f32] [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = _lib.lib()
vp = ctypes.c_void_p
L.fv_debug_aa_snake.argtypes = [vp] * 6 + [ctypes.c_int] * 3 + [vp]
rng = np.random.default_rng(0)
taps = torch.tensor([0.002, -0.01, 0.03, -0.08, 0.2, 0.36, 0.36, 0.2, -0.08, 0.03, -0.01, 0.002], device="cuda")
side = [torch.cuda.Stream() for _ in range(2)]
main = torch.cuda.Stream()
stages = [(256, 752), (128, 6016), (64, 12032), (32, 24064)]
tot = 0
def noise(C, T):
    for s_ in side:
        xa = torch.randn(B, C, T, device="cuda") * 50.0
        ya = torch.empty_like(xa)
        al = torch.rand(C, device="cuda") + 0.5
        with torch.cuda.stream(s_):
            for _ in range(3):
                L.fv_debug_aa_snake(xa.data_ptr(), ya.data_ptr(), al.data_ptr(), al.data_ptr(), taps.data_ptr(), taps.data_ptr(), B, C, T, vp(s_.cuda_stream))
for C, T in stages:
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            conv = FusedConv(w, rng.normal(size=C).astype(np.float32), dilation=d, padding=(k * d - d) // 2).set_precision(prec)
            x = torch.randn(B, C, T, device="cuda")
            r = torch.randn(B, C, T, device="cuda")
            y0 = conv(x, r).clone()
            torch.cuda.synchronize()
            bad = 0; worst = 0.0
            for _ in range(30):
                noise(C, T)
                with torch.cuda.stream(main):
                    y = conv(x, r)
                torch.cuda.synchronize()
                if not torch.equal(y, y0):
                    bad += 1; worst = max(worst, float(torch.nan_to_num(y - y0).abs().max()))
            if bad:
                print(f"C={C} T={T} k={k} d={d} {_lib.last_kernel()}: {bad}/30 differ, max|d| {worst:.3e}")
                tot += 1
for cin, cout, k, u, T in [(512, 256, 16, 8, 94), (256, 128, 16, 8, 752), (128, 64, 4, 2, 6016), (64, 32, 4, 2, 12032)]:
    w = (rng.normal(size=(cin, cout, k)) / np.sqrt(cin * k / u)).astype(np.float32)
    conv = FusedConv(w, rng.normal(size=cout).astype(np.float32), transposed=True, stride=u, padding=(k - u) // 2).set_precision(prec)
    x = torch.randn(B, cin, T, device="cuda")
    y0 = conv(x).clone()
    torch.cuda.synchronize()
    bad = 0
    for _ in range(30):
        noise(cout, T * u)
        with torch.cuda.stream(main):
            y = conv(x)
        torch.cuda.synchronize()
        bad += int(not torch.equal(y, y0))
    if bad:
        print(f"convT {cin}->{cout} k={k} u={u} T={T} {_lib.last_kernel()}: {bad}/30 differ"); tot += 1
# conv_pre
w = (rng.normal(size=(512, 80, 7)) / np.sqrt(80 * 7)).astype(np.float32)
conv = FusedConv(w, rng.normal(size=512).astype(np.float32), padding=3).set_precision(prec)
x = torch.randn(B, 80, 94, device="cuda"); y0 = conv(x).clone(); torch.cuda.synchronize(); bad = 0
for _ in range(30):
    noise(256, 752)
    with torch.cuda.stream(main):
        y = conv(x)
    torch.cuda.synchronize(); bad += int(not torch.equal(y, y0))
if bad: print(f"conv_pre {_lib.last_kernel()}: {bad}/30 differ"); tot += 1
print(f"{prec} B={B}: {tot} shapes changed when run beside aa_snake launches")
