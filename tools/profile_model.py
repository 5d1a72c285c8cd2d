#!/usr/bin/env python
"""Per-kernel hipEvent table (fv_profile_*) of one BASELINE config: python tools/profile_model.py {hifigan|bigvgan|vocos} [B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config
model = sys.argv[1] if len(sys.argv) > 1 else "vocos"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"hifigan": 32, "bigvgan": 64, "vocos": 128}[model]
if model == "hifigan":
    cfg = dict(syn.HIFIGAN_V1_44K); eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=syn.hifigan_state_dict(cfg, 0)); T = 86
elif model == "bigvgan":
    cfg = dict(syn.BIGVGAN_24K); eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0)); T = 94
else:
    cfg = dict(syn.VOCOS_24K); eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]), state_dict=syn.vocos_state_dict(cfg, 0)); T = 94
mel = torch.from_numpy(syn.synthetic_mel(B, 80, T, 1234)).cuda()
for _ in range(3): eng(mel)
tab = eng.profile(mel, repeats=3)
tot = sum(r["total_ms"] for r in tab) / 3
print(model, "B", B, "serialized kernel ms", round(tot, 3))
for r in sorted(tab, key=lambda r: -r["total_ms"]):
    ms = r["total_ms"] / 3
    print(f"{ms:7.3f} ms x{r['launches']//3:3d} avg {r['avg_ms']*1e3:7.1f} us {r['flops_per_launch']/r['avg_ms']/1e9:6.1f} TF {r['bytes_per_launch']/r['avg_ms']/1e6:7.0f} GB/s  {r['kernel']}")
