#!/usr/bin/env python
"""Locates the first buffer that differs between repeated forwards of the opt-in f16x3 BigVGAN engine.  With
FV_DEBUG_STOP=<stage*100 + pair*10 + half> the engine stops after that dilation pair of that stage and leaves the branch buffers in
the workspace (cur / S / Y, then XB, XT, XA per branch); this probe repeats the truncated forward and reports, per run that
differs from the first, which buffers differ and where:   FV_DEBUG_STOP=210 python tools/probe_f16_locate.py [iters] [prec]"""
import os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
cfg = dict(syn.BIGVGAN_24K)
eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0), precision=prec)
eng.set_graph_replay(False)
T, B = 94, 1
mel = torch.from_numpy(syn.synthetic_mel(B, 80, T, 12)).cuda()
out = torch.empty((B, 1, eng.output_length(T)), device="cuda")
names = ["cur", "S", "Y"] + [f"{n}{j}" for j in range(3) for n in ("XB", "XT", "XA")]
ref, bad = None, 0
tally = {}
for i in range(iters):
    eng(mel, out)
    torch.cuda.synchronize()
    ws = eng._ws.clone()
    if ref is None:
        ref = ws
        continue
    if torch.equal(ref.view(torch.int32), ws.view(torch.int32)):
        continue
    bad += 1
    me = ws.numel() // len(names)
    desc = []
    for b, n in enumerate(names):
        d = (ref[b * me:(b + 1) * me].view(torch.int32) != ws[b * me:(b + 1) * me].view(torch.int32))
        if d.any():
            idx = torch.nonzero(d)[:, 0]
            tally[n] = tally.get(n, 0) + 1
            desc.append((n, int(idx.numel()), int(idx.min()), int(idx.max())))
    if bad <= 6:
        print(f"  run {i}: " + "; ".join(f"{n}: {c} words in [{lo}, {hi}]" for n, c, lo, hi in desc))
print(f"FV_DEBUG_STOP={os.environ.get('FV_DEBUG_STOP')}: {bad} / {iters - 1} runs differ; buffers that differed: {tally}")
