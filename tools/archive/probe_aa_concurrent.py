#!/usr/bin/env python
"""aa_snake on three streams at once (as the three MRF branches run it) vs alone: python tools/probe_aa_concurrent.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
L = _lib.lib()
vp = ctypes.c_void_p
L.fv_debug_aa_snake.argtypes = [vp] * 6 + [ctypes.c_int] * 3 + [vp]
L.fv_debug_poison_lds.argtypes = [vp]
torch.manual_seed(0)
taps = torch.tensor([0.002, -0.01, 0.03, -0.08, 0.2, 0.36, 0.36, 0.2, -0.08, 0.03, -0.01, 0.002], device="cuda")
for B, C, T in [(1, 128, 6016), (1, 64, 12032), (1, 32, 24064), (2, 128, 6016), (1, 256, 752)]:
    xs = [torch.randn(B, C, T, device="cuda") for _ in range(3)]
    al = [torch.rand(C, device="cuda") + 0.5 for _ in range(3)]
    ib = [torch.rand(C, device="cuda") + 0.5 for _ in range(3)]
    ref = []
    for j in range(3):
        y = torch.empty_like(xs[j])
        L.fv_debug_aa_snake(xs[j].data_ptr(), y.data_ptr(), al[j].data_ptr(), ib[j].data_ptr(), taps.data_ptr(), taps.data_ptr(), B, C, T, vp(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        ref.append(y)
    streams = [torch.cuda.Stream() for _ in range(3)]
    bad = 0
    worst = 0.0
    for it in range(200):
        ys = [torch.empty_like(x) for x in xs]
        for j in range(3):
            with torch.cuda.stream(streams[j]):
                if it % 2: L.fv_debug_poison_lds(vp(streams[j].cuda_stream))
                L.fv_debug_aa_snake(xs[j].data_ptr(), ys[j].data_ptr(), al[j].data_ptr(), ib[j].data_ptr(), taps.data_ptr(), taps.data_ptr(), B, C, T, vp(streams[j].cuda_stream))
        torch.cuda.synchronize()
        for j in range(3):
            if not torch.equal(ys[j], ref[j]):
                bad += 1
                worst = max(worst, float(torch.nan_to_num(ys[j] - ref[j]).abs().max()))
    print(f"B={B} C={C} T={T}: {bad} / 600 concurrent launches differ from the solo result (max |d| {worst:.3e})")
