#!/usr/bin/env python
"""Phase time stamps of ONE launch of the split-K latency kernel (needs a library built with -DFV_X_SPLITK_TS):
FV_LIB_PATH=.../libfishvoc_x1.so python tools/probe_splitk_timeline.py C k T [d]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
C, k, T = map(int, sys.argv[1:4])
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(0)
w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
conv = FusedConv(w, np.zeros(C, np.float32), dilation=d, padding=(k * d - d) // 2, pre_act=_lib.FV_ACT_SILU)
x = torch.randn(1, C, T, device="cuda:0"); r = torch.randn(1, C, T, device="cuda:0"); y = torch.empty_like(x)
for _ in range(5):
    conv(x, r, y)
torch.cuda.synchronize()
ts = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda:0")
L = _lib.lib()
L.fv_debug_set_splitk_timestamps.argtypes = [ctypes.c_void_p]
for rep in range(3):
    ts.zero_()
    L.fv_debug_set_splitk_timestamps(ts.data_ptr())
    for _ in range(3):
        conv(x, r, y)          # the third launch overwrites the first two: back-to-back steady state
    torch.cuda.synchronize()
    L.fv_debug_set_splitk_timestamps(None)
    a = ts.cpu().numpy().reshape(-1, 8)
    a = a[a[:, 0] != 0][:, :5].astype(np.float64) / 100.0   # us (100 MHz counter)
    t0 = a[:, 0].min()
    a -= t0
    ph = np.diff(a, axis=1)
    print(f"{_lib.last_kernel()}: {len(a)} workgroups, span {a[:, 4].max():.1f} us; start p50 {np.percentile(a[:, 0], 50):.1f} p90 {np.percentile(a[:, 0], 90):.1f} max {a[:, 0].max():.1f}")
    for i, n in enumerate(["start -> first chunk staged", "K loop", "reduce barrier", "sum + epilogue (incl. store drain)"]):
        print(f"    {n:36s} p10 {np.percentile(ph[:, i], 10):6.2f}  p50 {np.percentile(ph[:, i], 50):6.2f}  p90 {np.percentile(ph[:, i], 90):6.2f} us")
    print(f"    workgroup life                       p10 {np.percentile(a[:, 4] - a[:, 0], 10):6.2f}  p50 {np.percentile(a[:, 4] - a[:, 0], 50):6.2f}  p90 {np.percentile(a[:, 4] - a[:, 0], 90):6.2f} us")
