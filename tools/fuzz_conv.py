#!/usr/bin/env python
"""Differential fuzz of the fused conv entry point (fv_conv_forward): random channel counts, kernel sizes, dilations, paddings,
strides (transposed), activations, residuals, batch sizes and lengths, both precisions, against the CPU oracle.
python tools/fuzz_conv.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib
from vocoder_amd.engine import FusedConv
from oracle import oracle as orc

ACT = {_lib.FV_ACT_NONE: lambda v: v, _lib.FV_ACT_SILU: orc.silu, _lib.FV_ACT_TANH: orc.tanh, _lib.FV_ACT_GELU: orc.gelu}


def _nan_bordered(a: np.ndarray) -> torch.Tensor:
    buf = torch.full((a.size + 2048,), float("nan"), device="cuda")
    v = buf[1024:1024 + a.size].view(*a.shape)
    v.copy_(torch.from_numpy(a))
    return v


def run(n_cases=60, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for i in range(n_cases):
        if rng.random() < 0.2:   # fused (c1, c2) ResBlock pair entry point (fv_conv_pair_forward)
            C, k, d = int(rng.choice([16, 32, 64, 128, 256])), int(rng.choice([3, 7, 11])), int(rng.choice([1, 3, 5]))
            big = rng.random() < 0.4
            B, T = (int(rng.integers(4, 20)), int(rng.integers(100, 1200))) if big else (int(rng.integers(1, 4)), int(rng.integers(1, 140)))
            w1 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k) * 1.2).astype(np.float32)
            w2 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
            b1, b2 = rng.normal(size=C).astype(np.float32) * 0.1, rng.normal(size=C).astype(np.float32) * 0.1
            x = (rng.normal(size=(B, C, T)) * 1.5).astype(np.float32)
            h = orc.conv1d(orc.silu(x), w1, b1, dilation=d, padding=(k - 1) * d // 2)
            ref = x + orc.conv1d(orc.silu(h), w2, b2, dilation=1, padding=(k - 1) // 2)
            scale = max(1.0, float(np.abs(ref).max()))
            for prec in ("f32", "f16x3"):
                if (prec == "f32" or os.environ.get("FV_NO_F16X3_PAIRS")) and not (C in (16, 32) or (C == 64 and k == 3)):
                    continue   # no exact-fp32 pair kernel for the wide stages (they run per layer)
                c1 = FusedConv(w1, b1, dilation=d, padding=(k - 1) * d // 2).set_precision(prec)
                c2 = FusedConv(w2, b2, padding=(k - 1) // 2).set_precision(prec)
                y = c1.pair(c2, _nan_bordered(x))
                torch.cuda.synchronize()
                err = float(np.abs(y.cpu().numpy() - ref).max())
                worst = max(worst, err / scale)
                if verbose or err > 2e-5 * scale:
                    print(f"case {i:3d} {prec:5s} pair C={C} k={k} d={d} B={B} T={T} {_lib.last_kernel()} err={err:.2e} (scale {scale:.1f})")
                assert err <= 2e-5 * scale, ("pair", C, k, d, B, T, prec, _lib.last_kernel(), err)
                c1.close(); c2.close()
            continue
        transposed = rng.random() < 0.25
        pick = lambda *v: int(rng.choice(v))
        cin = pick(1, 2, 7, 16, 24, 32, 40, 64, 80, 100, 128, 192, 256, 300, 384, 512)
        cout = pick(1, 3, 16, 20, 32, 48, 64, 96, 128, 130, 256, 320, 512)
        big = rng.random() < 0.35
        B = int(rng.integers(6, 40)) if big else int(rng.integers(1, 4))
        T = int(rng.integers(60, 500)) if big else int(rng.integers(1, 70))
        pre = pick(_lib.FV_ACT_NONE, _lib.FV_ACT_SILU)
        post = pick(_lib.FV_ACT_NONE, _lib.FV_ACT_NONE, _lib.FV_ACT_TANH, _lib.FV_ACT_GELU)
        if transposed:
            k, u = [(2, 2), (4, 2), (8, 2), (16, 8), (8, 8), (10, 5), (8, 4), (4, 4), (3, 2), (7, 3), (6, 3)][int(rng.integers(0, 11))]
            pad = (k - u) // 2
            w = (rng.normal(size=(cin, cout, k)) / np.sqrt(cin * max(k // u, 1))).astype(np.float32)
            kw = dict(transposed=True, stride=u, padding=pad)
            ref_fn = lambda xa: orc.conv_transpose1d(xa, w, b, stride=u, padding=pad)
            desc = f"convT k={k} u={u} pad={pad}"
        else:
            k, d = pick(1, 1, 2, 3, 3, 4, 5, 7, 7, 11, 11, 13), pick(1, 1, 2, 3, 5)
            if k == 1:
                d = 1
            pad = (k - 1) * d // 2 if rng.random() < 0.7 else int(rng.integers(0, (k - 1) * d + 1))
            if T + 2 * pad < (k - 1) * d + 1:
                pad = (k - 1) * d
            w = (rng.normal(size=(cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
            kw = dict(dilation=d, padding=pad)
            ref_fn = lambda xa: orc.conv1d(xa, w, b, dilation=d, padding=pad)
            desc = f"conv k={k} d={d} pad={pad}"
        b = rng.normal(size=cout).astype(np.float32) if rng.random() < 0.8 else None
        x = (rng.normal(size=(B, cin, T)) * 1.5).astype(np.float32)
        ref = ref_fn(ACT[pre](x))
        res = rng.normal(size=ref.shape).astype(np.float32) if rng.random() < 0.5 else None
        if res is not None:
            ref = ref + res
        ref = ACT[post](ref)
        scale = max(1.0, float(np.abs(ref).max()))
        # both sides accumulate K = cin * k products in fp32 in different orders: the bound grows with sqrt(K) (seen: 2.25e-5
        # at K = 5632, the only case in 4000 above 2e-5)
        scale *= max(1.0, float(np.sqrt(cin * k / 1024.0)))
        for prec in ("f32", "f16x3"):
            conv = FusedConv(w, b, pre_act=pre, post_act=post, **kw).set_precision(prec)
            guard = torch.full((ref.size + 2048,), 9.5, device="cuda")   # 1024 floats of guard band either side of the output
            y = guard[1024:1024 + ref.size].view(*ref.shape)
            conv(_nan_bordered(x), None if res is None else _nan_bordered(res), y)   # inputs sit between NaNs: a stray read that is used shows
            torch.cuda.synchronize()
            assert bool((guard[:1024] == 9.5).all()) and bool((guard[1024 + ref.size:] == 9.5).all()), ("write outside the output", desc, cin, cout, B, T, prec)
            kern = _lib.last_kernel()
            err = float(np.abs(y.cpu().numpy() - ref).max())
            worst = max(worst, err / scale)
            if verbose or err > 2e-5 * scale:
                print(f"case {i:3d} {prec:5s} {desc} cin={cin} cout={cout} B={B} T={T} pre={pre} post={post} res={res is not None} "
                      f"bias={b is not None} {kern} err={err:.2e} (scale {scale:.1f})")
            assert y.shape == ref.shape, (y.shape, ref.shape)
            assert err <= 2e-5 * scale, (desc, cin, cout, B, T, prec, kern, err)
            conv.close()
    return worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("worst |d| / scale =", run(n, s))
