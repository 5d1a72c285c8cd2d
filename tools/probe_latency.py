#!/usr/bin/env python
"""Per-kernel hipEvent table of a B=1 forward (single-clip latency): python tools/probe_latency.py [f32|f16x3]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
cfg = dict(syn.HIFIGAN_V1_44K); sd = syn.hifigan_state_dict(cfg, seed=0)
eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd, precision=prec)
mel = torch.from_numpy(syn.synthetic_mel(1, 80, 86, seed=1)).cuda()
for _ in range(5): eng(mel)
torch.cuda.synchronize()
lat = []
for _ in range(30):
    t = time.perf_counter(); eng(mel); torch.cuda.synchronize(); lat.append((time.perf_counter() - t) * 1e3)
print(prec, "p50 ms", np.percentile(lat, 50))
tab = eng.profile(mel, repeats=3)
tot = sum(r["total_ms"] for r in tab) / 3
print("serialized kernel ms", tot, "launches", sum(r["launches"] for r in tab) // 3)
for r in sorted(tab, key=lambda r: -r["total_ms"])[:int(os.environ.get("TOP", "24"))]:
    print(f"{r['total_ms']/3:7.3f} ms x{r['launches']//3:2d} avg {r['avg_ms']*1e3:7.1f} us {r['flops_per_launch']/r['avg_ms']/1e9:6.1f} TF  {r['kernel']}")
