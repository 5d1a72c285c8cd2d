#!/usr/bin/env python
"""A few replayed throughput steps of one BASELINE config as shipped (branch streams + hipGraph replay), a host sync between
steps: for kernel traces (tools/step_timeline.py).   python tools/probe_step.py {hifigan|bigvgan} [B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
model = sys.argv[1] if len(sys.argv) > 1 else "hifigan"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"hifigan": 32, "bigvgan": 64}[model]
if model == "hifigan":
    cfg = dict(syn.HIFIGAN_V1_44K); eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=syn.hifigan_state_dict(cfg, 0)); T = 86
else:
    cfg = dict(syn.BIGVGAN_24K); eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0)); T = 94
mel = torch.from_numpy(syn.synthetic_mel(B, 80, T, 1234)).cuda()
out = torch.empty((B, 1, eng.output_length(T)), device="cuda")
import time
for _ in range(6):
    eng(mel, out)
    torch.cuda.synchronize()
    time.sleep(0.002)
print("ok")
