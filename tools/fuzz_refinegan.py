#!/usr/bin/env python
"""Differential fuzz: random RefineGAN widths / rates / clip lengths (both precisions) against the CPU oracle, with the AdaIN
noise injected on both sides.  python tools/fuzz_refinegan.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, refinegan_config
from oracle import oracle as orc


def run(n_cases=12, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    worst, done, i = 0.0, 0, 0
    while done < n_cases and i < 20 * n_cases:
        i += 1
        n = int(rng.integers(2, 5))
        down = tuple(int(r) for r in rng.choice([2, 2, 4, 8], size=n))
        up = tuple(int(r) for r in rng.permutation(down))
        cfg = dict(sampling_rate=44100, hop_length=int(np.prod(down)), downsample_rates=down, upsample_rates=up,
                   leaky_relu_slope=0.2, num_mels=int(rng.choice([12, 80, 128])), start_channels=int(rng.choice([2, 4, 8, 16])))
        B, T = int(rng.integers(1, 4)), int(rng.integers(1, 40))
        sd = syn.refinegan_state_dict(cfg, seed * 100 + i)
        mel = syn.synthetic_mel(B, cfg["num_mels"], T, seed + i)
        tmpl = syn.synthetic_template(B, T, cfg["hop_length"], seed + i + 1)
        noise = syn.refinegan_noise(cfg, B, T, seed=seed + i + 2)
        try:
            ref = orc.refinegan_forward(sd, cfg, mel, tmpl, noise)
        except Exception:   # lengths that the reference's torch.cat cannot join: the engine must refuse them too
            eng = Engine(_lib.FV_MODEL_REFINEGAN, refine=refinegan_config(**cfg), state_dict=sd)
            try:
                eng(torch.from_numpy(mel).cuda(), None, torch.from_numpy(tmpl).cuda(), torch.from_numpy(np.concatenate([q.reshape(-1) for q in noise])).cuda())
                raise AssertionError(f"engine accepted a length the oracle rejects: {cfg} B={B} T={T}")
            except AssertionError:
                raise
            except Exception:
                pass
            continue
        nz = torch.from_numpy(np.concatenate([q.reshape(-1) for q in noise])).cuda()
        for prec in ("f32", "f16x3"):
            eng = Engine(_lib.FV_MODEL_REFINEGAN, refine=refinegan_config(**cfg), state_dict=sd, precision=prec)
            y = eng(torch.from_numpy(mel).cuda(), None, torch.from_numpy(tmpl).cuda(), nz)
            torch.cuda.synchronize()
            err = float(np.abs(y.cpu().numpy() - ref).max())
            worst = max(worst, err)
            if verbose or err > 1e-4:
                print(f"case {done:3d} {prec:5s} B={B} T={T} down={down} up={up} start={cfg['start_channels']} mels={cfg['num_mels']} err={err:.2e}")
            assert y.shape == ref.shape and err <= 1e-4, (cfg, B, T, prec, err)
            eng.close()
        done += 1
    return worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("worst |d| =", run(n, s))
