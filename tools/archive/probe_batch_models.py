#!/usr/bin/env python
"""Step time per clip against the batch size for BigVGAN-24k / Vocos-24k (1 s clips): python tools/probe_batch_models.py {bigvgan|vocos} [batches]
(does a smaller batch whose tensors fit the 256 MB infinity cache run faster per clip?)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config
model = sys.argv[1] if len(sys.argv) > 1 else "bigvgan"
batches = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16, 24, 32, 48, 64]
if model == "bigvgan":
    cfg = dict(syn.BIGVGAN_24K); eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0)); T, M = 94, 80
else:
    cfg = dict(syn.VOCOS_24K); eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]), state_dict=syn.vocos_state_dict(cfg, 0)); T, M = 94, cfg["backbone"]["input_channels"]
for B in batches:
    mel = torch.from_numpy(syn.synthetic_mel(B, M, T, 1)).cuda()
    out = torch.empty((B, 1, eng.output_length(T)), device="cuda")
    for _ in range(3): eng(mel, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5): eng(mel, out)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5 * 1e3)
    ms = float(np.median(ts))
    print(f"{model} B={B:3d} {ms:8.3f} ms/step {ms / B:7.4f} ms/clip")
