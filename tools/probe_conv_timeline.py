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
# Are the co-resident workgroups of a CU in lock-step?  For every workgroup: the partner on the same CU whose life overlaps its own the longest; the
# offset between the two "chunk c staged" stamp trains, folded into one chunk period (0 = they stage together, 0.5 = perfectly alternating).
if nch > 3:
    offs, ovl = [], []
    for c in ucu:
        idx = np.where(cu == c)[0]
        for i in idx:
            best, bo = -1, 0.0
            for j in idx:
                if j == i:
                    continue
                o = min(end[i], end[j]) - max(start[i], start[j])
                if o > bo:
                    best, bo = j, o
            if best < 0:
                continue
            ti, tj = t[i, 1:1 + nch], t[best, 1:1 + nch]
            per_i = np.median(np.diff(ti))
            m = nch // 2
            d = np.abs(tj - ti[m]).min()
            offs.append((d % per_i) / per_i)
            ovl.append(bo / (end[i] - start[i]))
    offs = np.minimum(offs, 1.0 - np.array(offs))
    print(f"  partner stamp offset / chunk period (0 = lock-step, 0.5 = alternating): p10 {np.percentile(offs, 10):.2f} p50 {np.percentile(offs, 50):.2f} p90 {np.percentile(offs, 90):.2f}; "
          f"life overlap with that partner p50 {np.percentile(ovl, 50):.2f}")
    # chunk period of the workgroups by partner offset: in lock-step the two waves of a SIMD stage together and share the pipe in between
    per_wg = np.median(np.diff(t[:, 1:1 + nch], axis=1), axis=1)
    print("  workgroups per CU over time: max concurrent", max(int(sum((start[cu == c] <= tt) & (end[cu == c] > tt))) for c in ucu[:8] for tt in np.linspace(0, end.max(), 50)))
    print("  chunk period by launch order (first 512 / rest): p50 %.2f / %.2f" % (np.percentile(per_wg[np.argsort(start)[:512]], 50), np.percentile(per_wg[np.argsort(start)[512:]], 50) if len(per_wg) > 512 else float("nan")))
