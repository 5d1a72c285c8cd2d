#!/usr/bin/env python
"""What the reference's one-shot caller pays before its only forward (test.py:31-38,88-90): engine creation from a device state dict, split
into the D2H copy, fv_load_weight, fv_finalize; then the first and second forward of one clip.  python tools/probe_create.py [repeats]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd import engine as E
from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config


def models():
    cfg = dict(syn.HIFIGAN_V1_44K)
    yield "hifigan-v1-44k", _lib.FV_MODEL_HIFIGAN, dict(ups=upsampler_config(**cfg)), syn.hifigan_state_dict(cfg, 0), (1, 80, 86)
    cfg = dict(syn.BIGVGAN_24K)
    yield "bigvgan-24k", _lib.FV_MODEL_BIGVGAN, dict(ups=upsampler_config(**cfg)), syn.bigvgan_state_dict(cfg, 0), (1, 80, 94)
    cfg = dict(syn.VOCOS_24K)
    yield ("vocos-24k", _lib.FV_MODEL_VOCOS, dict(backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"])),
           syn.vocos_state_dict(cfg, 0), (1, 80, 94))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    L = _lib.lib()
    for name, kind, kw, sd, shape in models():
        sd_dev = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in sd.items()}
        torch.cuda.synchronize()
        mel = torch.from_numpy(syn.synthetic_mel(*shape, seed=3)).to(dev)
        for r in range(reps):
            t0 = time.perf_counter()
            host = list(E._host_arrays(sd_dev))
            t1 = time.perf_counter()
            eng = Engine(kind, state_dict=dict(host), **kw)
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            eng(mel); torch.cuda.synchronize()
            t4 = time.perf_counter()
            eng(mel); torch.cuda.synchronize()
            t5 = time.perf_counter()
            eng(mel); torch.cuda.synchronize()
            t6 = time.perf_counter()
            eng.close()
            # the whole thing as the module path does it
            t7 = time.perf_counter()
            eng = Engine(kind, state_dict=sd_dev, **kw)
            torch.cuda.synchronize()
            t8 = time.perf_counter()
            eng.close()
            print(f"{name:16s} rep {r}: {len(sd)} tensors {sum(v.size for v in sd.values()) * 4 / 1e6:.0f} MB | d2h {1e3 * (t1 - t0):7.1f} ms | load+finalize {1e3 * (t2 - t1):7.1f} ms | "
                  f"create from device dict {1e3 * (t8 - t7):7.1f} ms | forward 1st {1e3 * (t4 - t3):7.2f} 2nd (capture) {1e3 * (t5 - t4):7.2f} 3rd (replay) {1e3 * (t6 - t5):6.2f} ms", flush=True)


if __name__ == "__main__":
    main()
