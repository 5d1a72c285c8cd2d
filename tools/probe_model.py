#!/usr/bin/env python
"""A few forwards of one BASELINE config on ONE stream without graphs (for rocprofv3 --kernel-trace --stats / --pmc passes,
where per-kernel durations / counters must not overlap):
    python tools/probe_model.py {hifigan|bigvgan|vocos} [B] [iters] [f32|f16x3]
Defaults to the BASELINE batch of the model (32 / 64 / 128)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("FV_SINGLE_STREAM", "2")   # one stream, tree form: the kernel instances of the shipped step (LOG R6.9)
os.environ.setdefault("FV_NO_GRAPH", "1")
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config

model = sys.argv[1] if len(sys.argv) > 1 else "hifigan"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"hifigan": 32, "bigvgan": 64, "vocos": 128}[model]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
prec = sys.argv[4] if len(sys.argv) > 4 else "f32"
if model == "hifigan":
    cfg = dict(syn.HIFIGAN_V1_44K)
    eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=syn.hifigan_state_dict(cfg, 0), precision=prec)
    T = 86
elif model == "bigvgan":
    cfg = dict(syn.BIGVGAN_24K)
    eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0), precision=prec)
    T = 94
else:
    cfg = dict(syn.VOCOS_24K)
    eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]),
                 state_dict=syn.vocos_state_dict(cfg, 0), precision=prec)
    T = 94
mel = torch.from_numpy(syn.synthetic_mel(B, 80, T, 1234)).cuda()
out = torch.empty((B, 1, eng.output_length(T)), device="cuda")
for _ in range(iters):
    eng(mel, out)
torch.cuda.synchronize()
print("ok", model, B, float(out.abs().max()))
