#!/usr/bin/env python
"""90 s of forwards over mixed (batch, frames) shapes on one engine (each shape change re-captures the graph): outputs must stay
bit-identical per shape and device memory flat.  python tools/soak.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
cfg = dict(syn.HIFIGAN_V1_44K); sd = syn.hifigan_state_dict(cfg, 0)
eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd)
free0 = torch.cuda.mem_get_info()[0]
ref = {}
t0 = time.time()
n = 0
rng = np.random.default_rng(0)
shapes = [(32, 86), (1, 86), (7, 33), (3, 200), (16, 86), (1, 12)]
mels = {s: torch.from_numpy(syn.synthetic_mel(s[0], 80, s[1], 5)).cuda() for s in shapes}
while time.time() - t0 < 90:
    s = shapes[int(rng.integers(len(shapes)))]
    y = eng(mels[s])
    n += 1
    if n % 50 == 0:
        torch.cuda.synchronize()
        yc = y.clone()
        if s in ref:
            assert torch.equal(ref[s], yc), ("output changed", s, float((ref[s] - yc).abs().max()))
        else:
            ref[s] = yc
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(f"soak: {n} forwards of mixed shapes in {time.time() - t0:.0f} s, outputs bit-stable, free memory {free0 >> 20} -> {free1 >> 20} MiB")
