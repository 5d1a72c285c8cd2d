#!/usr/bin/env python
"""Per-wave life spans of ONE persistent pointwise-GEMM launch (debug hook fv_debug_set_pw_timestamps):
python tools/probe_pw_timeline.py cin cout gelu res variant [B] [T]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
cin, cout, gelu, res = map(int, sys.argv[1:5])
os.environ["FV_PW"] = sys.argv[5]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 128
T = int(sys.argv[7]) if len(sys.argv) > 7 else 94
rng = np.random.default_rng(0)
w = (rng.normal(size=(cout, cin, 1)) / np.sqrt(cin)).astype(np.float32)
conv = FusedConv(w, rng.normal(size=cout).astype(np.float32), post_act=_lib.FV_ACT_GELU if gelu else _lib.FV_ACT_NONE)
x = torch.randn(B, cin, T, device="cuda:0")
r = torch.randn(B, cout, T, device="cuda:0") if res else None
y = torch.empty(B, cout, T, device="cuda:0")
for _ in range(3):
    conv(x, r, y)
torch.cuda.synchronize()
NW = 256 * 4 * 4 * 4
ts = torch.zeros(NW * 4, dtype=torch.int64, device="cuda:0")
L = _lib.lib()
L.fv_debug_set_pw_timestamps.argtypes = [ctypes.c_void_p]
L.fv_debug_set_pw_timestamps(ts.data_ptr())
conv(x, r, y)
torch.cuda.synchronize()
L.fv_debug_set_pw_timestamps(None)
a = ts.cpu().numpy().reshape(-1, 4)
a = a[a[:, 1] != 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0      # us
hw = a[:, 2] & 0xffffffff
xcc = a[:, 2] >> 32
units = (a[:, 3] & 0xffffffff) - (a[:, 3] >> 32)
if os.environ.get("FV_TL_SAVE"):
    np.save(os.environ["FV_TL_SAVE"], a)
print(f"{_lib.last_kernel()}: {len(a)} waves, kernel span {en.max():.1f} us; start p50 {np.percentile(st, 50):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f}; "
      f"end p10 {np.percentile(en, 10):.1f} p50 {np.percentile(en, 50):.1f} p90 {np.percentile(en, 90):.1f}; life mean {np.mean(en - st):.1f}")
print("units per wave:", dict(zip(*np.unique(units, return_counts=True))))
for k in np.unique(units):
    m = units == k
    print(f"  units={k}: n={m.sum()} life mean {np.mean((en - st)[m]):.1f} us, end mean {en[m].mean():.1f}")
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
simd = (hw >> 4) & 3
slot = hw & 0xf
key = xcc * 100000 + se * 1000 + sh * 100 + cu
print("distinct (xcc,se,sh,cu):", len(np.unique(key)), " waves per CU histogram:", dict(zip(*np.unique(np.unique(key, return_counts=True)[1], return_counts=True))))
print("late starters (> 5 us):", int((st > 5).sum()), " wave slots seen:", dict(zip(*np.unique(slot, return_counts=True))))
for xx in np.unique(xcc):
    m = xcc == xx
    print(f"  xcc {xx}: waves {m.sum()} end max {en[m].max():.1f} mean {en[m].mean():.1f} start max {st[m].max():.1f}")
