#!/usr/bin/env python
"""Differential fuzz of the drop-in module stack: UnifyGenerator(ConvNeXtEncoder, HiFiGANGenerator) (the Firefly composition)
with random widths / rates, loaded through load_state_dict(strict=True), both precisions, against the CPU oracle.
python tools/fuzz_firefly.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import synthetic as syn
from vocoder_amd.modules.encoders.convnext import ConvNeXtEncoder
from vocoder_amd.modules.generators.hifigan import HiFiGANGenerator
from vocoder_amd.modules.generators.unify import UnifyGenerator
from oracle import oracle as orc


def run(n_cases=12, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for i in range(n_cases):
        ns = int(rng.integers(1, 4))
        dims = [int(rng.choice([32, 64, 96, 128, 192, 320, 384])) for _ in range(ns)]
        nst = int(rng.integers(2, 4))
        rates = [int(rng.choice([2, 4, 8])) for _ in range(nst)]
        mels = int(rng.choice([20, 80, 128]))
        cfg = dict(backbone=dict(input_channels=mels, depths=[int(rng.integers(1, 3)) for _ in range(ns)], dims=dims, kernel_size=7),
                   head=dict(hop_length=int(np.prod(rates)), upsample_rates=rates, upsample_kernel_sizes=[2 * r for r in rates],
                             resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=dims[-1],
                             upsample_initial_channel=int(rng.choice([32, 64, 128])), use_template=False,
                             pre_conv_kernel_size=int(rng.choice([7, 13])), post_conv_kernel_size=int(rng.choice([7, 13]))))
        B, T = int(rng.integers(1, 4)), int(rng.integers(1, 25))
        sd = syn.firefly_state_dict(cfg, seed * 100 + i)
        mel = syn.synthetic_mel(B, mels, T, seed + i)
        ref = orc.firefly_forward(sd, cfg, mel)
        for prec in ("f32", "f16x3"):
            gen = UnifyGenerator(ConvNeXtEncoder(**cfg["backbone"]), HiFiGANGenerator(**cfg["head"]))
            gen.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
            gen = gen.eval().cuda()
            for m in gen.modules():
                if hasattr(m, "precision") and m.precision != prec:
                    m.precision = prec
            y = gen(torch.from_numpy(mel).cuda())
            torch.cuda.synchronize()
            err = float(np.abs(y.cpu().numpy() - ref).max())
            worst = max(worst, err)
            if verbose or err > 1e-4:
                print(f"case {i:3d} {prec:5s} B={B} T={T} dims={dims} rates={rates} C0={cfg['head']['upsample_initial_channel']} err={err:.2e}")
            assert y.shape == ref.shape and err <= 1e-4, (cfg, B, T, prec, err)
    return worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("worst |d| =", run(n, s))
