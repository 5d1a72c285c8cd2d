#!/usr/bin/env python
"""Differential fuzz: random ConvNeXt widths / depths / ISTFT geometries / batch sizes / clip lengths through the Vocos engine
(both precisions) against the CPU oracle.  python tools/fuzz_vocos.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, convnext_config, istft_head_config
from oracle import oracle as orc


def random_case(rng):
    n_stages = int(rng.integers(1, 4))
    dims = [int(rng.choice([24, 64, 72, 128, 192, 200, 256, 320, 384, 448, 512, 640, 704, 1024, 1100, 1408])) for _ in range(n_stages)]
    depths = [int(rng.integers(1, 3)) for _ in range(n_stages)]
    hop = int(rng.choice([64, 128, 256, 320]))
    n_fft = hop * int(rng.choice([2, 4])) if rng.random() < 0.8 else hop * 3 // 2 // 2 * 2
    mels = int(rng.choice([20, 80, 100]))
    cfg = dict(backbone=dict(input_channels=mels, depths=depths, dims=dims, kernel_size=7),
               head=dict(dim=dims[-1], n_fft=n_fft, hop_length=hop, win_length=n_fft, padding="same"))
    big = rng.random() < 0.3
    return cfg, int(rng.integers(8, 40) if big else rng.integers(1, 4)), int(rng.integers(30, 100) if big else rng.integers(1, 25))


def run(n_cases=30, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for i in range(n_cases):
        cfg, B, T = random_case(rng)
        sd = syn.vocos_state_dict(cfg, seed * 1000 + i)
        mel = syn.synthetic_mel(B, cfg["backbone"]["input_channels"], T, seed + i)
        ref = orc.vocos_forward(sd, cfg, mel)
        scale = float(np.abs(ref).max())   # bar relative to the waveform's own peak (small with the synthetic head)
        for prec in ("f32", "f16x3"):
            eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]),
                         state_dict=sd, precision=prec)
            x = torch.from_numpy(mel).cuda()
            y = eng(x); y2 = eng(x); y3 = eng(x)
            torch.cuda.synchronize()
            err = float(np.abs(np.nan_to_num(y.cpu().numpy(), nan=1e9) - ref).max())
            same = bool(torch.equal(y, y2) and torch.equal(y, y3))
            worst = max(worst, err / scale)
            if verbose or err > 1e-4 * scale or not same:
                print(f"case {i:3d} {prec:5s} B={B} T={T} dims={cfg['backbone']['dims']} depths={cfg['backbone']['depths']} "
                      f"n_fft={cfg['head']['n_fft']} hop={cfg['head']['hop_length']} err={err:.2e} (ref max {scale:.2f}) replay_identical={same}")
            assert err <= 1e-4 * scale and same, (cfg, B, T, prec, err, same)
            eng.close()
    return worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("worst |d| / scale =", run(n, s))
