#!/usr/bin/env python
"""Single-clip latency of two settings of one library knob, interleaved call by call in ONE process (run-to-run p50 varies by +-0.03 ms: more than most knobs move
it).  One engine per setting: each captures its launch sequence under its setting; then the two replay alternately.
    python tools/ab_latency_knob.py FV_SPLITK_DIRECT 1 0 [calls]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
var, vals = sys.argv[1], sys.argv[2:4]
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 300
cfg = dict(syn.HIFIGAN_V1_44K); sd = syn.hifigan_state_dict(cfg, seed=0)
mel = torch.from_numpy(syn.synthetic_mel(1, 80, 86, seed=1)).cuda()
engs, outs = [], []
for v in vals:
    os.environ[var] = v
    _lib.reload_env()
    e = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd)
    o = torch.empty((1, 1, e.output_length(86)), device="cuda")
    for _ in range(6):
        e(mel, o)           # (the third call with the same buffers captures; later ones replay)
    torch.cuda.synchronize()
    engs.append(e); outs.append(o)
lat = [[], []]
span = [[], []]
for i in range(calls):
    for k in ((0, 1) if i % 2 == 0 else (1, 0)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = time.perf_counter()
        e0.record(); engs[k](mel, outs[k]); e1.record()
        torch.cuda.synchronize()
        lat[k].append((time.perf_counter() - t) * 1e3)
        span[k].append(e0.elapsed_time(e1))
for k in (0, 1):
    print(f"{var}={vals[k]}: host-clock p50 {np.percentile(lat[k], 50):.4f} ms  p10 {np.percentile(lat[k], 10):.4f}  mean {np.mean(lat[k]):.4f};  GPU span p50 {np.percentile(span[k], 50):.4f}  mean {np.mean(span[k]):.4f}   ({calls} calls each, alternating)")
print("outputs equal:", bool(torch.equal(outs[0], outs[1])), " max |d|:", float((outs[0] - outs[1]).abs().max()))
