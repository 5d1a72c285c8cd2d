#!/usr/bin/env python
"""HiFiGAN-V1-44k step time against the batch size (1 s clips): python tools/probe_batch.py [f32|f16x3]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
cfg = dict(syn.HIFIGAN_V1_44K)
eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=syn.hifigan_state_dict(cfg, 0), precision=prec)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for B in ([int(v) for v in os.environ["PROBE_BATCHES"].split(",")] if os.environ.get("PROBE_BATCHES") else (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64)):
        mel = torch.from_numpy(syn.synthetic_mel(B, 80, 86, 1)).cuda()
        out = torch.empty((B, 1, 86 * 512), device="cuda")
        for _ in range(4):
            eng(mel, out)
        s.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            eng(mel, out)
        s.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"B={B:3d} {ms:8.3f} ms/step {ms / B:7.3f} ms/clip {B * 44032 / ms / 44.1:8.0f} x RT")
