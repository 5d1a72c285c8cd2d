#!/usr/bin/env python
"""Experiment: the three same-depth convs of a stage (k = 11 / 7 / 3) as (a) three launches on three streams — what the engine
does — against (b) ONE persistent launch over a shared tile queue (conv_fused3.hip, fv_debug_conv_fused3).  Wall time per round
of the three convs, interleaved rounds, medians; outputs compared.   python tools/experiments/probe_fused3.py [B]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = _lib.lib()
vp = ctypes.c_void_p
L.fv_debug_conv_fused3.argtypes = [vp, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_int32, ctypes.c_int32, vp, vp]
L.fv_debug_conv_fused3.restype = ctypes.c_int32
rng = np.random.default_rng(0)
for C, T in ((128, 5504), (256, 688), (64, 11008)):
    for d in (1, 5):
        convs, xs, ys, rs = [], [], [], []
        for k in (11, 7, 3):
            w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            convs.append(FusedConv(w, rng.normal(size=C).astype(np.float32), dilation=d, padding=(k * d - d) // 2, pre_act=_lib.FV_ACT_SILU))
            xs.append(torch.randn(B, C, T, device="cuda")); rs.append(torch.randn(B, C, T, device="cuda")); ys.append(torch.empty(B, C, T, device="cuda"))
        streams = [torch.cuda.Stream() for _ in range(3)]
        counter = torch.zeros(4, dtype=torch.int32, device="cuda")

        def run_streams():
            for c, x, r, y, s in zip(convs, xs, rs, ys, streams):
                with torch.cuda.stream(s):
                    c(x, r, y)

        def run_fused(out):
            arr = lambda ts: (vp * 3)(*[vp(t.data_ptr()) for t in ts])
            st = L.fv_debug_conv_fused3(convs[0]._h, convs[1]._h, convs[2]._h, arr(xs), arr(out), arr(rs), B, T, vp(counter.data_ptr()),
                                        vp(torch.cuda.current_stream().cuda_stream))
            assert st == 0, _lib.last_error() if hasattr(_lib, "last_error") else st

        run_streams(); torch.cuda.synchronize()
        y2 = [torch.full_like(y, float("nan")) for y in ys]
        run_fused(y2); torch.cuda.synchronize()
        same = all(bool(torch.equal(a, b)) for a, b in zip(ys, y2))
        ta, tb = [], []
        for rnd in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): run_streams()
            torch.cuda.synchronize(); ta.append((time.perf_counter() - t0) / 10 * 1e3)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): run_fused(y2)
            torch.cuda.synchronize(); tb.append((time.perf_counter() - t0) / 10 * 1e3)
        fl = sum(2.0 * C * C * k * T * B for k in (11, 7, 3))
        a, b = np.median(ta), np.median(tb)
        print(f"C={C} T={T} d={d} B={B}: three streams {a:.3f} ms ({fl / a / 1e9:.1f} TF)   fused queue {b:.3f} ms ({fl / b / 1e9:.1f} TF)   x{a / b:.3f}  identical={same}", flush=True)
