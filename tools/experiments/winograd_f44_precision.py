#!/usr/bin/env python
"""CPU experiment (build container only: imports the reference through tests/golden/gen_golden.py, like winograd_precision.py): what do Winograd
F(4,4) tap groups (conv_wino44_impl.h: quad lattice, groups of FOUR taps {0..3},{4..7},{8..11} with the taps past k zero, transformed weights rounded
to float32, float32 input / output transforms) cost in accuracy, and which interpolation points?  Layers with >= 64 channels and k = 7 / 11 take
F(4,4), the rest F(2,3); waveforms against a float64 forward.
    python tools/experiments/winograd_f44_precision.py [hifigan|bigvgan] [frames]
Results: see the end of the file."""
import os, sys, numpy as np, torch, torch.nn.functional as F
from fractions import Fraction as Fr
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import winograd_precision as wp
gg, syn = wp.gg, wp.syn

def cook_toom(points, m=4, r=4):
    n = m + r - 1
    a = [Fr(p) for p in points]; assert len(a) == n - 1
    AT = [[(a[j] ** i if j < n - 1 else Fr(int(i == m - 1))) for j in range(n)] for i in range(m)]
    G = []
    for j in range(n - 1):
        f = Fr(1)
        for l in range(n - 1):
            if l != j: f *= (a[j] - a[l])
        G.append([a[j] ** k / f for k in range(r)])
    G.append([Fr(int(k == r - 1)) for k in range(r)])
    # solve BT from the bilinear identity
    ATn = np.array(AT, dtype=float); Gn = np.array(G, dtype=float)
    rows = []; rhs = []
    for j in range(m):
        for k in range(r):
            for i in range(n):
                row = np.zeros((n, n))
                for p in range(n): row[p, i] = ATn[j, p] * Gn[p, k]
                rows.append(row.ravel()); rhs.append(1.0 if i == j + k else 0.0)
    BT = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0].reshape(n, n)
    res = np.abs(np.array(rows) @ BT.ravel() - np.array(rhs)).max()
    return ATn, Gn, BT, res

PTS = {"std": [Fr(1, 2), -Fr(1, 2), 1, -1, 2, -2], "0,±1,±2,1/2": [0, 1, -1, 2, -2, Fr(1, 2)], "0,±1,±1/2,2": [0, 1, -1, Fr(1, 2), -Fr(1, 2), 2],
       "0,±1,±2,3": [0, 1, -1, 2, -2, 3], "±3/4,±1,±3/2": [Fr(3, 4), -Fr(3, 4), 1, -1, Fr(3, 2), -Fr(3, 2)]}
MAT = {}
for kname, pts in PTS.items():
    MAT[kname] = cook_toom(pts)
    print(kname, "residual", MAT[kname][3])
np.set_printoptions(precision=4, suppress=True, linewidth=150)
print(MAT["std"][2]); print(MAT["std"][1]); print(MAT["std"][0])

SEL = {"pts": "std", "minC": 64}
_orig = wp._orig

def plan43(k):
    return {7: ([0, 4], []), 11: ([0, 4, 8], [])}.get(k)

def w43(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if isinstance(stride, tuple): stride = stride[0]
    if isinstance(padding, tuple): padding = padding[0]
    if isinstance(dilation, tuple): dilation = dilation[0]
    k = w.shape[-1]; pl = plan43(k)
    use43 = (wp.USE["on"] and x.dtype == torch.float32 and stride == 1 and groups == 1 and pl is not None and w.shape[1] >= SEL["minC"]
             and padding == (k - 1) * dilation // 2)
    if not use43:
        return wp.wino_conv1d(x, w, b, stride, padding, dilation, groups)
    AT, G, BT, _ = MAT[SEL["pts"]]
    D = dilation; B, C, T = x.shape
    Tp = -(-T // (4 * D)) * (4 * D)
    xp = F.pad(x, (padding, Tp - T + padding + 16 * D))
    nq = Tp // (4 * D); NP = nq * D
    n = torch.arange(NP + 4 * D)
    t0 = 4 * D * (n // D) + (n % D)
    X = [xp[..., t0 + j * D] for j in range(4)]
    def sh(a, s): return a[..., s:s + NP]
    def dval(i, g):   # x'[t0 + (4g + i) D]
        return sh(X[i % 4], (g + i // 4) * D)
    m = [torch.zeros(B, w.shape[0], NP) for _ in range(7)]
    S = [torch.zeros(B, w.shape[0], NP) for _ in range(4)]
    wd = np.concatenate([w.double().numpy(), np.zeros(w.shape[:2] + (1,))], -1)   # (the taps past k are zero)
    BTt = torch.tensor(BT, dtype=torch.float32)
    for j0 in pl[0]:
        g = j0 // 4
        gw = wd[..., j0:j0 + 4]                      # (o, c, 4)
        U = np.einsum("pk,ock->poc", G, gw).astype(np.float32)
        d = [dval(i, g) for i in range(7)]
        for p in range(7):
            V = sum(float(BT[p, i]) * d[i] for i in range(7) if abs(BT[p, i]) > 1e-12)   # fp32 transform
            m[p] = m[p] + torch.einsum("oc,bcn->bon", torch.from_numpy(U[p]), V)
    for j in pl[1]:
        g = j // 4
        for jj in range(4):
            S[jj] = S[jj] + torch.einsum("oc,bcn->bon", w[..., j], dval(3 + jj, g))
    y = torch.zeros(B, w.shape[0], Tp)
    nn_ = torch.arange(NP); tt = 4 * D * (nn_ // D) + (nn_ % D)
    for jj in range(4):
        yj = sum(float(AT[jj, p]) * m[p] for p in range(7) if abs(AT[jj, p]) > 1e-12) + S[jj]
        y[..., tt + jj * D] = yj
    y = y[..., :T]
    if b is not None: y = y + b[None, :, None]
    return y

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "hifigan"; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    torch.manual_seed(0)
    if which == "hifigan":
        cfg = dict(syn.HIFIGAN_V1_44K); sd = syn.hifigan_state_dict(cfg, 1); g = gg.HiFiGANGenerator(**cfg).eval()
    else:
        cfg = dict(syn.BIGVGAN_24K); sd = syn.bigvgan_state_dict(cfg, 1); g = gg.BigVGANGenerator(**cfg).eval()
    g.load_state_dict(gg._t(sd), strict=(which == "hifigan"))
    mel = torch.from_numpy(syn.synthetic_mel(1, cfg["num_mels"], frames, 7))
    with torch.no_grad():
        x = torch.randn(1, 64, 50)
        F.conv1d = w43; torch.nn.functional.conv1d = w43
        for k, D in ((7, 3), (11, 5), (11, 1), (7, 1)):
            w = torch.randn(64, 64, k) * 0.1
            wp.USE["on"] = False; a = _orig(x, w, None, 1, (k - 1) * D // 2, D)
            a64 = _orig(x.double(), w.double(), None, 1, (k - 1) * D // 2, D)
            wp.USE["on"] = True
            for pn in PTS:
                SEL["pts"] = pn
                c = w43(x, w, None, 1, (k - 1) * D // 2, D)
                print(f"layer k={k} D={D} {pn}: direct err {float((a-a64).abs().max()):.2e} f44 err {float((c - a64).abs().max()):.2e} (peak {float(a64.abs().max()):.2f})")
        wp.USE["on"] = False
        F.conv1d = _orig; torch.nn.functional.conv1d = _orig
        ref64 = g.double()(mel.double()).float(); g.float(); y32 = g(mel)
        F.conv1d = wp.wino_conv1d; torch.nn.functional.conv1d = wp.wino_conv1d
        wp.USE["on"] = True; y23 = g(mel)
        F.conv1d = w43; torch.nn.functional.conv1d = w43
        print(f"{which} {frames} frames peak {float(ref64.abs().max()):.3f}")
        print(f"  direct  max {float((y32-ref64).abs().max()):.3e}")
        print(f"  F(2,3)  max {float((y23-ref64).abs().max()):.3e}")
        for pn in PTS:
            SEL["pts"] = pn
            y = g(mel)
            print(f"  F(4,4) {pn}: max {float((y-ref64).abs().max()):.3e} rms {float((y-ref64).pow(2).mean().sqrt()):.3e}")
main()

# Results (HiFiGAN-V1 16 frames / BigVGAN 12 frames; max |waveform - float64 forward|):
#   direct fp32 sums 2.0e-6 / 3.2e-6,  F(2,3) 1.7e-6 / 3.0e-6,  F(4,3) (winograd_f43_precision.py) 2.0e-6 / 4.5e-6,
#   F(4,4) ±1/2, ±1, ±2, ∞ ("std", shipped) 2.3e-6 / 3.9e-6;  0, ±1, ±2, 1/2: 2.8e-6 / 3.6e-6;  0, ±1, ±1/2, 2: 3.2e-6 / 5.0e-6;
#   ±3/4, ±1, ±3/2: 3.3e-6 / 7.9e-6;  0, ±1, ±2, 3: 7.0e-6 / 1.4e-5.
# Single layers (64 channels, unit-variance input): 1.3 - 2.3 x the direct sum's error with the shipped points.  The symmetric set without 0 is as accurate as
# any and the cheapest to apply (even / odd parts shared by ±a: 21 instructions per lattice element).
