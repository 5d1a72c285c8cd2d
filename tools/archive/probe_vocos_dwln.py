#!/usr/bin/env python
"""Vocos-24k B=128 step time and the dwconv + LayerNorm kernels of its per-launch table: python tools/probe_vocos_dwln.py
(FV_DWLN_RR=1: round-robin tile mapping instead of the XCD-grouped one)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, time
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, convnext_config, istft_head_config
cfg = dict(syn.VOCOS_24K)
eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]), state_dict=syn.vocos_state_dict(cfg, 0))
B, T = 128, 94
mel = torch.from_numpy(syn.synthetic_mel(B, cfg["backbone"]["input_channels"], T, 1)).cuda()
out = torch.empty((B, 1, eng.output_length(T)), device="cuda")
for _ in range(3): eng(mel, out)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(10): eng(mel, out)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
print("vocos B=128 ms/step median", np.median(ts), "min", min(ts))
tab = eng.profile(mel, repeats=3)
for r in sorted(tab, key=lambda r: -r["total_ms"]):
    if "dwconv" in r["kernel"] or "ln" in r["kernel"].lower():
        print(f"{r['total_ms']/3:7.3f} ms x{r['launches']//3:2d} avg {r['avg_ms']*1e3:7.1f} us {r['bytes_per_launch']/r['avg_ms']/1e6:7.0f} GB/s  {r['kernel']}")
