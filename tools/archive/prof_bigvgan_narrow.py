import os, sys
sys.path.insert(0, "/root/repo")
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
cfg = dict(syn.BIGVGAN_24K)
eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0))
mel = torch.from_numpy(syn.synthetic_mel(64, 80, 94, 1)).cuda()
tab = eng.profile(mel, repeats=2)
for r in tab:
    if "cin=256" in r["kernel"] or "cin=128" in r["kernel"] or "cin=512" in r["kernel"]: continue
    print(f"{r['kernel'][:70]:70s} n={r['launches']//2:3d} avg {r['avg_ms']:.4f} ms  {r['flops_per_launch']/r['avg_ms']/1e9:6.1f} TF  tot/step {r['total_ms']/2:.3f}")
