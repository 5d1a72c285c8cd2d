#!/usr/bin/env python
"""A/B of the two fused ResBlock pair kernels on the narrow HiFiGAN-V1 stages (B = 32): resblock_pair.hip (one tile per 4-wave
workgroup, FV_PAIR_SYNC=0) against pair_sync.hip (persistent, phase-synchronous 8-wave workgroups, FV_PAIR_SYNC=1);
interleaved rounds, medians, results compared.    python tools/probe_pair_sync.py [B] [C,T ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(32, 22016), (64, 11008)]
ROUNDS = int(os.environ.get("FV_PROBE_ROUNDS", "5"))
rng = np.random.default_rng(0)


def timeit(f, iters=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot = {"0": 0.0, "1": 0.0}
for C, T in shapes:
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            w1 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            w2 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            c1 = FusedConv(w1, rng.normal(size=C).astype(np.float32), dilation=d, padding=(k * d - d) // 2)
            c2 = FusedConv(w2, rng.normal(size=C).astype(np.float32), padding=(k - 1) // 2)
            x = torch.randn(B, C, T, device="cuda")
            ys, names, times = {}, {}, {"0": [], "1": []}
            ok = True
            for rnd in range(ROUNDS):
                for v in ("0", "1"):
                    os.environ["FV_PAIR_SYNC"] = v
                    _lib.reload_env()
                    try:
                        if rnd == 0:
                            ys[v] = c1.pair(c2, x)
                            names[v] = _lib.last_kernel()
                        times[v].append(timeit(lambda: c1.pair(c2, x)))
                    except Exception as exc:   # shape without a kernel in this mode
                        ok = False
                        names[v] = f"n/a ({str(exc)[:40]})"
                        times[v].append(float("nan"))
            fl = 4.0 * C * C * k * T * B
            m0, m1 = np.median(times["0"]), np.median(times["1"])
            if ok:
                tot["0"] += m0; tot["1"] += m1
            same = bool(torch.equal(ys["0"], ys["1"])) if ok else None
            md = float((ys["0"] - ys["1"]).abs().max()) if ok else float("nan")
            print(f"C={C:3d} T={T:6d} k={k:2d} d={d}  tile {m0:7.3f} ms {fl / m0 / 1e9:6.1f} TF   sync {m1:7.3f} ms {fl / m1 / 1e9:6.1f} TF "
                  f"x{m0 / m1:5.3f}  identical={same} maxdiff={md:.1e}  ({names['0']} | {names['1']})", flush=True)
print(f"sum over shapes both kernels ran: tile {tot['0']:.2f} ms, sync {tot['1']:.2f} ms")
