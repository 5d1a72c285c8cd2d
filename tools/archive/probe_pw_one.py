#!/usr/bin/env python
"""Runs ONE pointwise-conv shape a few times (for rocprofv3 --pmc passes):
python tools/probe_pw_one.py cin cout gelu(0/1) res(0/1) variant [B] [T] [iters]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
cin, cout, gelu, res = map(int, sys.argv[1:5])
os.environ["FV_PW"] = sys.argv[5]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 128
T = int(sys.argv[7]) if len(sys.argv) > 7 else 94
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 5
rng = np.random.default_rng(0)
w = (rng.normal(size=(cout, cin, 1)) / np.sqrt(cin)).astype(np.float32)
conv = FusedConv(w, rng.normal(size=cout).astype(np.float32), post_act=_lib.FV_ACT_GELU if gelu else _lib.FV_ACT_NONE)
x = torch.randn(B, cin, T, device="cuda:0")
r = torch.randn(B, cout, T, device="cuda:0") if res else None
y = torch.empty(B, cout, T, device="cuda:0")
for _ in range(iters):
    conv(x, r, y)
torch.cuda.synchronize()
print(_lib.last_kernel())
