import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/vocoder_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
cfg = dict(syn.HIFIGAN_V1_44K); sd = syn.hifigan_state_dict(cfg, seed=0)
eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd)
mel = torch.from_numpy(syn.synthetic_mel(1, 80, 86, seed=1)).cuda()
for _ in range(5): eng(mel)
torch.cuda.synchronize()
tab = eng.profile(mel, repeats=3)
tot = sum(r["total_ms"] for r in tab) / 3
print("serialized kernel ms", tot)
for r in sorted(tab, key=lambda r: -r["total_ms"]):
    print(f"{r['total_ms']/3*1e3:7.1f} us x{r['launches']//3:2d} avg {r['avg_ms']*1e3:7.1f} us {r['flops_per_launch']/r['avg_ms']/1e9:6.1f} TF {r['bytes_per_launch']/r['avg_ms']/1e6:7.0f} GB/s  {r['kernel']}")
