#!/usr/bin/env python
"""Does more stream-level concurrency help?  One engine at B=32 vs. n engines at B=32/n on their own streams."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
cfg = dict(syn.HIFIGAN_V1_44K)
sd = syn.hifigan_state_dict(cfg, 0)
for n in (1, 2, 4):
    B = 32 // n
    engs = [Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    mels = [torch.from_numpy(syn.synthetic_mel(B, 80, 86, 1234 + i)).cuda() for i in range(n)]
    outs = [torch.empty((B, 1, 86 * 512), device="cuda") for _ in range(n)]
    def step():
        for e, s, m, o in zip(engs, streams, mels, outs):
            with torch.cuda.stream(s):
                e(m, o)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    print(n, "engines x B =", B, ":", (time.perf_counter() - t0) / 20 * 1e3, "ms per 32 clips")
