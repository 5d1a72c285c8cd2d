#!/usr/bin/env python
"""Throughput of the other BASELINE configs (not bench lines, context for DESIGN.md):
config[2] BigVGAN 24 kHz B=64, config[3] Vocos 24 kHz B=128; plus per-kernel hipEvent tables."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, convnext_config, istft_head_config, refinegan_config, upsampler_config

PREC = sys.argv[1] if len(sys.argv) > 1 else "f32"   # f32 | f16x3

def run(name, eng, mel, sr, steps=10):
    out = torch.empty((mel.shape[0], 1, eng.output_length(mel.shape[2])), device="cuda")
    for _ in range(3):
        eng(mel, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng(mel, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n = out.numel()
    tab = eng.profile(mel, repeats=2)
    tot = sum(r["total_ms"] for r in tab) / 2
    top = sorted(tab, key=lambda r: -r["total_ms"])[:6]
    print(json.dumps({"model": name, "precision": PREC, "batch": mel.shape[0], "ms_per_step": dt * 1e3, "samples_per_s": n / dt,
                      "x_realtime": n / dt / sr, "finite": bool(torch.isfinite(out).all()),
                      "serialized_kernel_ms": tot,
                      "top_kernels": [(r["kernel"], round(r["total_ms"] / 2, 3), round(r["flops_per_launch"] / r["avg_ms"] / 1e9, 1)) for r in top]}))

cfg = dict(syn.BIGVGAN_24K)
eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0), precision=PREC)
run("bigvgan-24k", eng, torch.from_numpy(syn.synthetic_mel(64, 80, 94, 1)).cuda(), 24000)
del eng
cfg = dict(syn.VOCOS_24K)
eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]),
             state_dict=syn.vocos_state_dict(cfg, 0), precision=PREC)
run("vocos-24k", eng, torch.from_numpy(syn.synthetic_mel(128, 80, 94, 2)).cuda(), 24000)
del eng
cfg = dict(syn.FIREFLY_BASE_44K)   # the composition the reference ships scripts for (firefly-gan-base.yaml), 32 one-second clips
eng = Engine(_lib.FV_MODEL_FIREFLY, backbone=convnext_config(**cfg["backbone"]), ups=upsampler_config(**cfg["head"]),
             state_dict=syn.firefly_state_dict(cfg, 0), precision=PREC)
run("firefly-gan-base-44k", eng, torch.from_numpy(syn.synthetic_mel(32, 128, 86, 5)).cuda(), 44100)
del eng
# RefineGAN (reference ctor defaults: 44.1 kHz, hop 256, start_channels 16): 16 one-second clips
cfg = dict(syn.REFINEGAN_44K)
eng = Engine(_lib.FV_MODEL_REFINEGAN, refine=refinegan_config(**cfg), state_dict=syn.refinegan_state_dict(cfg, 0), precision=PREC)
B, T = 16, 172
mel = torch.from_numpy(syn.synthetic_mel(B, cfg["num_mels"], T, 3)).cuda()
tmpl = torch.from_numpy(syn.synthetic_template(B, T, cfg["hop_length"], 4)).cuda()
noise = torch.randn(eng.noise_elems(B, T), device="cuda")
out = torch.empty((B, 1, T * cfg["hop_length"]), device="cuda")
for _ in range(3):
    eng(mel, out, tmpl, noise)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    eng(mel, out, tmpl, noise)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(json.dumps({"model": "refinegan-44k", "precision": PREC, "batch": B, "ms_per_step": dt * 1e3, "samples_per_s": out.numel() / dt,
                  "x_realtime": out.numel() / dt / 44100, "finite": bool(torch.isfinite(out).all())}))
