#!/usr/bin/env python
"""Experiment: the B = 32 headline step as N concurrent sub-batches on N engines / streams (does filling one sub-batch's solo upsampler and stage tails with the
other's branch kernels pay?):  python tools/probe_split_batch.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = dict(syn.HIFIGAN_V1_44K); sd = syn.hifigan_state_dict(cfg, seed=0)
dev = torch.device("cuda:0")
mel = torch.from_numpy(syn.synthetic_mel(32, 80, 86, seed=1)).to(dev)
for n in (1, 2, 4, 1, 2, 4):
    engs = [Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd) for _ in range(n)]
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    parts = [p.contiguous() for p in mel.chunk(n)]
    outs = [torch.empty((p.shape[0], 1, engs[0].output_length(86)), device=dev) for p in parts]
    def step():
        for e, s, p, o in zip(engs, streams, parts, outs):
            with torch.cuda.stream(s):
                e(p, o)
    for _ in range(4): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{n} sub-batch(es) of {32 // n}: {ms:.3f} ms per 32-clip step")
    for e in engs: e.close()
