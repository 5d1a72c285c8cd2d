#!/usr/bin/env python
"""HiFiGAN-V1-44k step time by batch size with the Winograd convs off (FV_WINO=0), gated by launch size (1, the default) and forced
(2): where the gate `workgroups >= FV_WINO_MIN_BLOCKS` should sit.    python tools/sweep_wino_batch.py [model]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config

model = sys.argv[1] if len(sys.argv) > 1 else "hifigan"
if model == "hifigan":
    cfg = dict(syn.HIFIGAN_V1_44K); sd = syn.hifigan_state_dict(cfg, 0); kind = _lib.FV_MODEL_HIFIGAN; frames = 86
else:
    cfg = dict(syn.BIGVGAN_24K); sd = syn.bigvgan_state_dict(cfg, 0); kind = _lib.FV_MODEL_BIGVGAN; frames = 94
GATES = [g for g in os.environ.get("GATES", "").split(",") if g]
for B in ((1, 2, 3, 4, 6, 8, 12, 16, 24, 32) if GATES else (1, 2, 3, 4, 6, 8, 12, 16, 24, 32)):
    mel = torch.from_numpy(syn.synthetic_mel(B, 80, frames, 1)).cuda()
    res = []
    for mode in (GATES or ("0", "1", "2")):
        if GATES:
            os.environ["FV_WINO"] = "1"
            os.environ["FV_WINO_MIN_BLOCKS"] = mode
        else:
            os.environ["FV_WINO"] = mode
        _lib.reload_env()
        eng = Engine(kind, ups=upsampler_config(**cfg), state_dict=sd)
        out = torch.empty((B, 1, eng.output_length(frames)), device="cuda")
        best = 1e9
        for r in range(3):
            for _ in range(3):
                eng(mel, out)
            torch.cuda.synchronize()
            n = max(10, 200 // B)
            t0 = time.perf_counter()
            for _ in range(n):
                eng(mel, out)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n * 1e3)
        res.append(best)
        eng.close()
    print(f"{model} B={B:3d}: " + ("  ".join(f"gate {g} {r:7.3f}" for g, r in zip(GATES, res)) if GATES else f"direct {res[0]:7.3f} ms  gated {res[1]:7.3f}  forced {res[2]:7.3f}"), flush=True)
