#!/usr/bin/env python
"""BigVGAN-24k B = 64 step time (BASELINE config[2]) for the library FV_LIB_PATH points at (default: the shipped one), a few
interleavable rounds; prints ms/step, the aa_snake share and the worst deviation from the CPU oracle on one short clip.
    FV_LIB_PATH=vocoder_amd/csrc/libfishvoc_x.so python tools/ab_bigvgan.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = dict(syn.BIGVGAN_24K)
sd = syn.bigvgan_state_dict(cfg, 0)
eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=sd)
mel = torch.from_numpy(syn.synthetic_mel(64, 80, 94, 1)).cuda()
out = torch.empty((64, 1, eng.output_length(94)), device="cuda")
for r in range(rounds):
    for _ in range(3):
        eng(mel, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng(mel, out)
    torch.cuda.synchronize()
    print(f"round {r}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms/step", flush=True)
tab = eng.profile(mel, repeats=2)
tot = sum(x["total_ms"] for x in tab) / 2
aa = sum(x["total_ms"] for x in tab if x["kernel"].startswith("aa_snake")) / 2
import hashlib
print("output sha1 of the B = 64 step:", hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16])
print(f"serialized kernel sum {tot:.2f} ms, aa_snake {aa:.2f} ms ({os.environ.get('FV_LIB_PATH', 'shipped library')})")
if "--oracle" in sys.argv:
    from oracle import oracle as orc
    m = syn.synthetic_mel(2, 80, 12, seed=5)
    ref = orc.bigvgan_forward(sd, cfg, m)
    y = eng(torch.from_numpy(m).cuda()).cpu().numpy()
    print(f"max|d| vs oracle on a (2, 80, 12) clip: {np.abs(y - ref).max():.3e} (waveform peak {np.abs(ref).max():.2f})")
if "--identical" in sys.argv:   # fused amp_conv path against the aa_snake + conv launches (FV_NO_AMP_FUSION=1 engine): bit for bit
    os.environ["FV_NO_AMP_FUSION"] = "1"
    eng2 = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=sd)
    del os.environ["FV_NO_AMP_FUSION"]
    for B, T in ((64, 94), (3, 37), (1, 5)):
        m = torch.from_numpy(syn.synthetic_mel(B, 80, T, seed=B + T)).cuda()
        ya, yb = eng(m), eng2(m)
        torch.cuda.synchronize()
        print(f"B={B} T={T}: fused == unfused bit for bit: {bool(torch.equal(ya, yb))}  max|d| = {float((ya - yb).abs().max()):.2e}")
