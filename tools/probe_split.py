#!/usr/bin/env python
"""Experiment: does running the 32-clip batch as S independent sub-batches on S streams (each with its own engine,
branch streams and graph) beat one 32-clip forward?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
cfg = dict(syn.HIFIGAN_V1_44K)
sd = syn.hifigan_state_dict(cfg, 0)
B = 32
mel = torch.from_numpy(syn.synthetic_mel(B, 80, 86, 1234)).cuda()
for S in (1, 2, 4):
    engs = [Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    parts = [mel[i * B // S:(i + 1) * B // S].contiguous() for i in range(S)]
    outs = [torch.empty((B // S, 1, 86 * 512), device="cuda") for _ in range(S)]
    def step():
        for e, s, p, o in zip(engs, streams, parts, outs):
            with torch.cuda.stream(s):
                e(p, o)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    print(f"S={S}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per 32 clips")
    del engs
