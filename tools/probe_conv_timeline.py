#!/usr/bin/env python
"""Per-workgroup phase time stamps of ONE launch of the tiled conv kernel (library built with -DFV_X_CONV_TS):
FV_LIB_PATH=.../libfishvoc_x1.so python tools/probe_conv_timeline.py C k T [d] [B] [res|nores]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
C, k, T = map(int, sys.argv[1:4])
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
use_res = (sys.argv[6] != "nores") if len(sys.argv) > 6 else True
rng = np.random.default_rng(0)
w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
conv = FusedConv(w, np.zeros(C, np.float32), dilation=d, padding=(k * d - d) // 2, pre_act=_lib.FV_ACT_SILU)
x = torch.randn(B, C, T, device="cuda:0"); r = torch.randn(B, C, T, device="cuda:0") if use_res else None; y = torch.empty_like(x)
for _ in range(5):
    conv(x, r, y)
torch.cuda.synchronize()
ts = torch.zeros(16384 * 16, dtype=torch.int64, device="cuda:0")
L = _lib.lib()
L.fv_debug_set_splitk_timestamps.argtypes = [ctypes.c_void_p]
L.fv_debug_set_splitk_timestamps(ts.data_ptr())
for _ in range(2):
    conv(x, r, y)
torch.cuda.synchronize()
L.fv_debug_set_splitk_timestamps(None)
a = ts.cpu().numpy().reshape(-1, 16)
a = a[a[:, 0] != 0]
hw = a[:, 15]
t = a[:, :15].astype(np.float64) / 100.0
t0 = t[:, 0].min()
nch = int(((t[0, 1:13] != 0).sum()))
start, first, loop_end, end = t[:, 0] - t0, t[:, 1] - t0, t[:, 13] - t0, t[:, 14] - t0
print(f"{_lib.last_kernel()}: {len(a)} workgroups, {nch} chunks, span {end.max():.1f} us")
pct = lambda v: f"p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}"
print("  start                     ", pct(start))
print("  start -> chunk 0 staged   ", pct(first - start))
if nch > 1:
    per = np.diff(t[:, 1:1 + nch], axis=1)
    print("  chunk period              ", pct(per.ravel()))
print("  last chunk staged -> loop end", pct(loop_end - (t[:, nch] - t0)))
print("  epilogue (incl. drain)    ", pct(end - loop_end))
print("  workgroup life            ", pct(end - start))
# per CU occupancy over time: HW_ID bits: wave_id 3:0 simd 5:4 ... cu 11:8 sh 12 se 15:13 ; xcc in high word
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | ((hw >> 32) << 8)
ucu = np.unique(cu)
cnt = np.array([np.sum(cu == c) for c in ucu])
print(f"  CUs seen {len(ucu)}; workgroups per CU min {cnt.min()} mean {cnt.mean():.2f} max {cnt.max()}")
ends = np.array([end[cu == c].max() for c in ucu])
print("  last end per CU           ", pct(ends))
