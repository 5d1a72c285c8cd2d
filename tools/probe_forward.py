#!/usr/bin/env python
"""A few HiFiGAN-V1 forwards at the bench shape on ONE stream without graphs (for rocprofv3 --pmc passes, where per-kernel
counters must not overlap): FV_SINGLE_STREAM=1 FV_NO_GRAPH=1 python tools/probe_forward.py [B] [iters] [f32|f16x3]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("FV_SINGLE_STREAM", "1")
os.environ.setdefault("FV_NO_GRAPH", "1")
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
cfg = dict(syn.HIFIGAN_V1_44K)
eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=syn.hifigan_state_dict(cfg, 0), precision=prec)
mel = torch.from_numpy(syn.synthetic_mel(B, 80, 86, 1234)).cuda()
out = torch.empty((B, 1, 86 * 512), device="cuda")
for _ in range(iters):
    eng(mel, out)
torch.cuda.synchronize()
print("ok", float(out.abs().max()))
