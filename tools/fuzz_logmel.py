#!/usr/bin/env python
"""Differential fuzz of the log-mel front-end: random STFT geometries / mel counts / lengths vs the CPU oracle.
python tools/fuzz_logmel.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd.data.transforms import LogMelSpectrogram
from oracle import oracle as orc


def run(n_cases=25, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for i in range(n_cases):
        hop = int(rng.choice([32, 64, 100, 128, 160, 256, 300, 512]))
        n_fft = int(rng.choice([hop, 2 * hop, 3 * hop, 4 * hop, hop * 3 // 2 // 2 * 2, hop * 5 // 2 // 2 * 2]))
        n_fft = max(n_fft, 2) // 2 * 2
        sr = int(rng.choice([16000, 22050, 24000, 44100, 48000]))
        cfg = dict(sample_rate=sr, n_fft=n_fft, win_length=n_fft, hop_length=hop, n_mels=int(rng.choice([20, 64, 80, 100, 128])),
                   f_min=float(rng.choice([0.0, 40.0])), f_max=int(rng.choice([sr // 2, sr // 2, 8000 if sr >= 16000 else sr // 2])))
        B = int(rng.integers(1, 5))
        L = hop * int(rng.integers(2, 40)) + int(rng.choice([0, 0, 1, hop // 2]))
        if L <= (n_fft - hop) // 2 + 1:
            L = n_fft
        wave = (0.2 * rng.normal(size=(B, L))).astype(np.float32)
        try:
            ref = orc.logmel_forward(wave, cfg)
        except Exception as exc:   # the reference's reflect padding rejects clips shorter than the pad
            if verbose:
                print(f"case {i:3d} skipped by the oracle: {type(exc).__name__}")
            continue
        m = LogMelSpectrogram(**cfg).eval().cuda()
        mel = m(torch.from_numpy(wave).cuda()[:, None, :])
        torch.cuda.synchronize()
        err = float(np.abs(mel.cpu().numpy() - ref).max())
        worst = max(worst, err)
        if verbose or err > 2e-4:
            print(f"case {i:3d} sr={sr} n_fft={n_fft} hop={hop} mels={cfg['n_mels']} f=[{cfg['f_min']},{cfg['f_max']}] B={B} L={L} -> {tuple(mel.shape)} err={err:.2e}")
        assert mel.shape == ref.shape and err <= 2e-4, (cfg, B, L, err)
    return worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("worst |d| =", run(n, s))
