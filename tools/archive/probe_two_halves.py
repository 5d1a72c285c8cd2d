#!/usr/bin/env python
"""Does a B=32 step run faster as two concurrent B=16 half-steps on two streams (serial sections of one half under the branch
sections of the other)?  Two engines with the same weights: python tools/probe_two_halves.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
cfg = dict(syn.HIFIGAN_V1_44K); sd = syn.hifigan_state_dict(cfg, 0)
B, T = 32, 86
mel = torch.from_numpy(syn.synthetic_mel(B, 80, T, 1)).cuda()
out = torch.empty((B, 1, T * 512), device="cuda")
def timeit(fn, n=20):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return float(np.median(ts))
e0 = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd)
s0 = torch.cuda.Stream()
def whole():
    with torch.cuda.stream(s0): e0(mel, out)
print("one B=32 step       %.3f ms" % timeit(whole))
for parts in (2, 4):
    engs = [Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    h = B // parts
    mels = [mel[i * h:(i + 1) * h].contiguous() for i in range(parts)]
    outs = [out[i * h:(i + 1) * h] for i in range(parts)]
    def split():
        for e, st, m, o in zip(engs, streams, mels, outs):
            with torch.cuda.stream(st): e(m, o)
    print("%d concurrent B=%d   %.3f ms" % (parts, h, timeit(split)))
