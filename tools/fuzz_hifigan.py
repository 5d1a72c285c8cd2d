#!/usr/bin/env python
"""Differential fuzz: random small HiFiGAN configurations / batch sizes / clip lengths through the engine (both precisions)
against the CPU oracle.  python tools/fuzz_hifigan.py [n_cases] [seed] [large] [bigvgan]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
from oracle import oracle as orc


def random_case(rng, large=False):
    n_stages = int(rng.integers(2, 5))
    rates = [int(rng.choice([2, 2, 4, 8])) for _ in range(n_stages)]
    ks = [int(r * rng.choice([1, 2])) if r > 2 else int(rng.choice([2, 4])) for r in rates]
    c0 = int(rng.choice([16, 32, 64, 128, 256]))
    while c0 >> n_stages < 2:
        c0 *= 2
    nk = int(rng.choice([1, 2, 3]))
    rk = [int(k) for k in rng.choice([3, 5, 7, 11], size=nk, replace=False)]
    dil = [[int(d) for d in rng.choice([1, 2, 3, 5], size=3)] for _ in range(nk)]
    cfg = dict(hop_length=int(np.prod(rates)), upsample_rates=rates, upsample_kernel_sizes=ks, resblock_kernel_sizes=rk,
               resblock_dilation_sizes=dil, num_mels=int(rng.choice([5, 20, 80])), upsample_initial_channel=c0, use_template=False,
               pre_conv_kernel_size=int(rng.choice([3, 7, 13])), post_conv_kernel_size=int(rng.choice([3, 7, 13])))
    if large:   # enough columns for the full-size tile shapes and the fused pair kernels (oracle: a few seconds per case)
        return cfg, int(rng.integers(4, 13)), int(rng.integers(40, 160))
    return cfg, int(rng.integers(1, 4)), int(rng.integers(1, 30))


def run(n_cases=30, seed=0, verbose=True, large=False, model="hifigan"):
    """model: "hifigan" or "bigvgan"; a third of the cases of either take the use_template=True branch."""
    rng = np.random.default_rng(seed)
    worst = 0.0
    for i in range(n_cases):
        cfg, B, T = random_case(rng, large)
        tmpl = None
        cfg["use_template"] = bool(rng.random() < 0.34)
        if model == "bigvgan":
            sd = syn.bigvgan_state_dict(cfg, seed * 1000 + i)
            mel = syn.synthetic_mel(B, cfg["num_mels"], T, seed + i)
            if cfg["use_template"]:
                tmpl = syn.synthetic_template(B, T, cfg["hop_length"], seed + i + 5)
            ref = orc.bigvgan_forward(sd, cfg, mel, template=tmpl)
        else:
            sd = syn.hifigan_state_dict(cfg, seed * 1000 + i)
            mel = syn.synthetic_mel(B, cfg["num_mels"], T, seed + i)
            if cfg["use_template"]:
                tmpl = syn.synthetic_template(B, T, cfg["hop_length"], seed + i + 5)
            ref = orc.hifigan_forward(sd, cfg, mel, template=tmpl)
        kind = _lib.FV_MODEL_BIGVGAN if model == "bigvgan" else _lib.FV_MODEL_HIFIGAN
        for prec in ("f32", "f16x3"):
            eng = Engine(kind, ups=upsampler_config(**cfg), state_dict=sd, precision=prec)
            x = torch.from_numpy(mel).cuda()
            tt = None if tmpl is None else torch.from_numpy(tmpl).cuda()
            y = eng(x, None, tt)
            y2 = eng(x, None, tt)     # second call: the graph-capture path
            y3 = eng(x, None, tt)     # third: replay
            torch.cuda.synchronize()
            err = float(np.abs(y.cpu().numpy() - ref).max())
            same = bool(torch.equal(y, y2) and torch.equal(y, y3))
            worst = max(worst, err)
            if verbose or err > 1e-4 or not same:
                print(f"case {i:3d} {model}{'+template' if tmpl is not None else ''} {prec:5s} B={B} T={T} C0={cfg['upsample_initial_channel']} rates={cfg['upsample_rates']} "
                      f"k={cfg['upsample_kernel_sizes']} rb={cfg['resblock_kernel_sizes']} dil={cfg['resblock_dilation_sizes']} "
                      f"pre/post={cfg['pre_conv_kernel_size']}/{cfg['post_conv_kernel_size']} err={err:.2e} replay_identical={same}")
            assert y.shape == ref.shape, (y.shape, ref.shape)
            assert err <= 1e-4 and same, (cfg, B, T, prec, err, same)
            eng.close()
    return worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("worst |d| =", run(n, s, large="large" in sys.argv[3:], model="bigvgan" if "bigvgan" in sys.argv[3:] else "hifigan"))
