#!/usr/bin/env python
"""pair_wino44 launches of the HiFiGAN-V1 narrow stages (B = 32), one library per process (FV_LIB_PATH): time per launch (hipEvents, back to back),
a checksum of the output (two builds that form the same sums print the same one) and the deviation from a float64 torch reference (2 clips).
  python tools/probe_pair44.py LABEL [C ...]"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, torch.nn.functional as F
from vocoder_amd.engine import FusedConv

label = sys.argv[1] if len(sys.argv) > 1 else "lib"
want = [int(a) for a in sys.argv[2:]] or [32]
shapes = {32: 22016, 16: 44032}
B = 32
torch.manual_seed(0)


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = 0.0
cells = []
for C in want:
    T = shapes[C]
    x = torch.randn(B, C, T, device="cuda", generator=torch.Generator(device="cuda").manual_seed(C))
    for k in (7, 11):
        for d in (1, 3, 5):
            rng = np.random.default_rng(1000 * C + 10 * k + d)   # (a cell's data does not depend on which other cells run)
            w1 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            w2 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            b1 = rng.normal(size=C).astype(np.float32)
            b2 = rng.normal(size=C).astype(np.float32)
            c1 = FusedConv(w1, b1, dilation=d, padding=(k * d - d) // 2)
            c2 = FusedConv(w2, b2, padding=(k - 1) // 2)
            y = c1.pair(c2, x)
            torch.cuda.synchronize()
            xs = x[:2].double()
            xt = F.conv1d(F.silu(xs), torch.from_numpy(w1).cuda().double(), torch.from_numpy(b1).cuda().double(), dilation=d, padding=(k * d - d) // 2)
            ref = xs + F.conv1d(F.silu(xt), torch.from_numpy(w2).cuda().double(), torch.from_numpy(b2).cuda().double(), padding=(k - 1) // 2)
            err = float((y[:2].double() - ref).abs().max())
            h = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:8]
            us = timed(lambda: c1.pair(c2, x))
            tot += us
            cells.append(f"C={C} k={k} d={d}: {us:6.1f} us {h} {err:.1e}")
print(f"{label:>10} | " + " | ".join(cells) + f" | sum {tot:7.1f} us")
