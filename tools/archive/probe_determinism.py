#!/usr/bin/env python
"""Repeats one forward many times and counts distinct outputs (bitwise): python tools/probe_determinism.py model B iters prec
env: FV_SINGLE_STREAM=1 (no branch streams), replay on/off via argv[5] (1/0)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config
model, B, iters, prec = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
replay = int(sys.argv[5]) if len(sys.argv) > 5 else 1
if model == "bigvgan":
    cfg = dict(syn.BIGVGAN_24K); eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0), precision=prec); T = 94
else:
    cfg = dict(syn.HIFIGAN_V1_44K); eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=syn.hifigan_state_dict(cfg, 0), precision=prec); T = 86
eng.set_graph_replay(bool(replay))
mel = torch.from_numpy(syn.synthetic_mel(B, 80, T, 12)).cuda()
out = torch.empty((B, 1, eng.output_length(T)), device="cuda")
ref = None
bad = 0
worst = 0.0
for i in range(iters):
    eng(mel, out)
    torch.cuda.synchronize()
    y = out.clone()
    if ref is None:
        ref = y
    elif not torch.equal(ref, y):
        bad += 1
        worst = max(worst, float((ref - y).abs().max()))
        if bad <= 4:
            d = (ref - y).abs().reshape(B, -1)
            idx = torch.nonzero(d > 0)
            tt = idx[:, 1]
            print(f"   run {i}: {idx.shape[0]} samples differ, items {sorted(set(idx[:, 0].tolist()))}, t range [{int(tt.min())}, {int(tt.max())}], "
                  f"{int((d > 1e-4).sum())} above 1e-4, first t above 1e-4: {int(torch.nonzero(d > 1e-4)[0, 1]) if (d > 1e-4).any() else -1}")
print(f"{model} B={B} prec={prec} replay={replay} single_stream={os.environ.get('FV_SINGLE_STREAM','0')}: {bad} / {iters - 1} runs differ from the first (max |d| {worst:.3e})")
