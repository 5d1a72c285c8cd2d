#!/usr/bin/env python
"""The engine's own per-launch table (profiling form: one stream, tree form, hipEvents around every launch) of a BASELINE configuration:
    python tools/probe_profile_table.py {hifigan|bigvgan|vocos} [B] [rows]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config

model = sys.argv[1] if len(sys.argv) > 1 else "hifigan"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"hifigan": 32, "bigvgan": 64, "vocos": 128}[model]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if model == "hifigan":
    cfg = dict(syn.HIFIGAN_V1_44K)
    eng, T = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=syn.hifigan_state_dict(cfg, 0)), 86
elif model == "bigvgan":
    cfg = dict(syn.BIGVGAN_24K)
    eng, T = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0)), 94
else:
    cfg = dict(syn.VOCOS_24K)
    eng, T = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]), state_dict=syn.vocos_state_dict(cfg, 0)), 94
mel = torch.from_numpy(syn.synthetic_mel(B, 80, T, 1234)).cuda()
for _ in range(2):
    eng(mel)
tab = eng.profile(mel, repeats=3)
tot = sum(r["total_ms"] for r in tab) / 3
print(f"{model} B = {B}: serialized kernel sum {tot:.3f} ms per forward")
for r in sorted(tab, key=lambda r: -r["total_ms"])[:rows]:
    n = r["launches"] / 3
    tf = r["flops_per_launch"] / (r["avg_ms"] * 1e-3) / 1e12
    gb = r["bytes_per_launch"] / (r["avg_ms"] * 1e-3) / 1e9
    print(f"  {r['total_ms'] / 3:7.3f} ms  x{n:5.1f}  {r['avg_ms'] * 1e3:8.1f} us  {tf:6.1f} TF/s alg  {gb:7.0f} GB/s alg  {r['kernel']}")
