"""CPU: the arithmetic conv_wino_impl.h and conv_wino4_impl.h implement, restated in numpy (float64) against the oracle's direct conv — Winograd
F(2,3) tap groups {0,1,2}, {4,5,6}, {8,9,10} on the dilated pair lattice plus the single taps 3, 7 folded into accumulators m0 / m3, and F(4,3)
groups on the quad lattice split over two plane halves, each with the virtual-tap order, plane offsets and weight transform of the kernel /
conv_layer_create.  Exact in real arithmetic: float64 agrees to ~1e-11, so an indexing or sign error in the scheme cannot hide behind a tolerance."""
import numpy as np
import pytest

from oracle import oracle as orc


def virtual_taps(k):
    """(plane, shift in units of D, accumulator, weight row) per virtual tap, in the kernel's order; planes 0..3 = d0..d3, 4 = E, 5 = O."""
    ng, ns = (k + 1) // 4, (k - 3) // 4
    taps = [(p, 2 * g, p, ("g", g, p)) for g in range(ng) for p in range(4)]
    for s in range(ns):
        taps += [(5, 2 * s + 1, 0, ("s", s, +1.0)), (4, 2 * s + 2, 3, ("s", s, -1.0))]
    return taps


def transformed_weights(w, k):
    """(C_out, C_in, NV): what conv_layer_create packs for the kernel."""
    out = []
    for (_, _, _, src) in virtual_taps(k):
        if src[0] == "g":
            g0, g1, g2 = (w[..., 4 * src[1] + i] for i in range(3))
            out.append([g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2][src[2]])
        else:
            out.append(src[2] * w[..., 4 * src[1] + 3])
    return np.stack(out, axis=-1)


def wino_conv1d(x, w, d):
    B, C, T = x.shape
    k = w.shape[-1]
    pad = (k - 1) * d // 2
    nq = -(-T // (2 * d))                      # blocks of 2 D samples
    NP = nq * d                                # pair columns
    halo = 2 * d * ((k + 1) // 4 - 1) + d      # largest shift + the transform's neighbour
    xp = np.zeros((B, C, 2 * d * nq + 2 * halo + 2 * d + pad))
    xp[..., pad:pad + T] = x                   # x'[tau] = x[tau - pad]
    n = np.arange(NP + halo)
    t0 = 2 * d * (n // d) + n % d
    E, O = xp[..., t0], xp[..., t0 + d]
    E1, O1 = np.roll(E, -d, axis=-1), np.roll(O, -d, axis=-1)
    planes = [E - E1, O + E1, E1 - O, O - O1, E, O]
    ww = transformed_weights(w, k)
    m = [np.zeros((B, w.shape[0], NP)) for _ in range(4)]
    for v, (pl, sh, acc, _) in enumerate(virtual_taps(k)):
        m[acc] += np.einsum("oc,bcn->bon", ww[..., v], planes[pl][..., sh * d:sh * d + NP])
    y = np.zeros((B, w.shape[0], 2 * d * nq))
    tt = 2 * d * (np.arange(NP) // d) + np.arange(NP) % d
    y[..., tt] = m[0] + m[1] + m[2]
    y[..., tt + d] = m[1] - m[2] - m[3]
    return y[..., :T]


@pytest.mark.parametrize("k,d,T", [(3, 1, 17), (3, 5, 40), (7, 1, 64), (7, 3, 1), (7, 5, 333), (11, 1, 129), (11, 3, 50), (11, 5, 9), (11, 5, 700)])
def test_pair_lattice_winograd_equals_the_direct_conv(k, d, T):
    rng = np.random.default_rng(k * 100 + d * 10 + T)
    x = rng.normal(size=(2, 6, T))
    w = rng.normal(size=(5, 6, k))
    ref = orc.conv1d(x.astype(np.float32), w.astype(np.float32), None, dilation=d, padding=(k - 1) * d // 2)
    y = wino_conv1d(x.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64), d)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())       # the float32 oracle's own rounding
    # and exactly, against a float64 direct sum
    xp = np.pad(x.astype(np.float32).astype(np.float64), ((0, 0), (0, 0), ((k - 1) * d // 2,) * 2))
    direct = sum(np.einsum("oc,bct->bot", w.astype(np.float32).astype(np.float64)[..., j], xp[..., j * d:j * d + T]) for j in range(k))
    assert np.abs(y - direct).max() <= 1e-11


def test_products_per_output_pair():
    assert [len(virtual_taps(k)) for k in (3, 7, 11)] == [4, 10, 16]       # against 6 / 14 / 22 for the direct sum


# ---- conv_wino4_impl.h: F(4,3) tap groups on the dilated quad lattice, two plane halves per 32-row tile ----
def virtual_taps4(k, h):
    """(plane name, shift in units of D, accumulator 0..3 of half h, weight source) per virtual tap in the kernel's order (Wino4Geom::off_of / acc_of).
    Half 0 accumulates m0 m1 m2 S1, half 1 m5 m4 m3 S2."""
    ng, ns = (k + 1) // 4, (k - 3) // 4
    taps = [(("V", i if h == 0 else 5 - i), g, i, ("g", g, i if h == 0 else 5 - i)) for g in range(ng) for i in range(3)]
    for s in range(ns):
        if h == 0:
            taps += [(("X", 3), s, 0, ("s", s)), (("X", 0), s + 1, 3, ("s", s))]
        else:
            taps += [(("X", 2), s + 1, 0, ("s", s)), (("X", 1), s + 1, 3, ("s", s))]
    return taps


def transformed_weights4(w, k, h):
    out = []
    for (_, _, _, src) in virtual_taps4(k, h):
        if src[0] == "g":
            g0, g1, g2 = (w[..., 4 * src[1] + i] for i in range(3))
            out.append([g0 / 4, -(g0 + g1 + g2) / 6, -(g0 - g1 + g2) / 6, g0 / 24 + g1 / 12 + g2 / 6, g0 / 24 - g1 / 12 + g2 / 6, g2][src[2]])
        else:
            out.append(w[..., 4 * src[1] + 3])
    return np.stack(out, axis=-1)


def wino4_conv1d(x, w, d):
    B, C, T = x.shape
    k = w.shape[-1]
    pad = (k - 1) * d // 2
    nq = -(-T // (4 * d))                      # blocks of 4 D samples
    NP = nq * d                                # quad columns
    halo = d * ((k + 1) // 4 - 1) + d          # largest shift + the transform's neighbour
    xp = np.zeros((B, C, 4 * d * (nq + 1) + 4 * halo + pad))
    xp[..., pad:pad + T] = x                   # x'[tau] = x[tau - pad]
    n = np.arange(NP + halo)
    t0 = 4 * d * (n // d) + n % d
    X = [xp[..., t0 + j * d] for j in range(4)]
    x0, x1, x2, x3 = X
    x4, x5 = np.roll(x0, -d, axis=-1), np.roll(x1, -d, axis=-1)
    V = [4 * x0 - 5 * x2 + x4, (x4 - 4 * x2) + (x3 - 4 * x1), (x4 - 4 * x2) - (x3 - 4 * x1), (x4 - x2) + 2 * (x3 - x1), (x4 - x2) - 2 * (x3 - x1),
         4 * x1 - 5 * x3 + x5]
    planes = {("V", p): V[p] for p in range(6)} | {("X", j): X[j] for j in range(4)}
    acc = []
    for h in (0, 1):
        ww = transformed_weights4(w, k, h)
        m = [np.zeros((B, w.shape[0], NP)) for _ in range(4)]
        for v, (pl, sh, a, _) in enumerate(virtual_taps4(k, h)):
            m[a] += np.einsum("oc,bcn->bon", ww[..., v], planes[pl][..., sh * d:sh * d + NP])
        acc.append(m)
    # the kernel's output transform: what each half keeps (A in acc 0, B in acc 3) and what it passes to its partner (slots 0 / 1)
    (m0, m1, m2, s1), (m5, m4, m3, s2) = acc
    slot = [[m1 + m2, m1 - m2], [m3 + m4, 2 * (m3 - m4)]]
    A0, B0 = m0 + (m1 + m2), s1 + (m1 - m2)
    A1, B1 = m5 + 8 * (m3 - m4), s2 + 4 * (m3 + m4)
    y = np.zeros((B, w.shape[0], 4 * d * nq))
    tt = 4 * d * (np.arange(NP) // d) + np.arange(NP) % d
    y[..., tt] = A0 + slot[1][0]
    y[..., tt + d] = B0 + slot[1][1]
    y[..., tt + 3 * d] = A1 + slot[0][1]
    y[..., tt + 2 * d] = B1 + slot[0][0]
    return y[..., :T]


@pytest.mark.parametrize("k,d,T", [(3, 1, 17), (3, 5, 40), (7, 1, 64), (7, 3, 1), (7, 5, 333), (11, 1, 129), (11, 3, 50), (11, 5, 9), (11, 5, 700), (11, 1, 3)])
def test_quad_lattice_winograd_equals_the_direct_conv(k, d, T):
    rng = np.random.default_rng(k * 100 + d * 10 + T)
    x = rng.normal(size=(2, 6, T)).astype(np.float32).astype(np.float64)
    w = rng.normal(size=(5, 6, k)).astype(np.float32).astype(np.float64)
    ref = orc.conv1d(x.astype(np.float32), w.astype(np.float32), None, dilation=d, padding=(k - 1) * d // 2)
    y = wino4_conv1d(x, w, d)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())       # the float32 oracle's own rounding
    xp = np.pad(x, ((0, 0), (0, 0), ((k - 1) * d // 2,) * 2))
    direct = sum(np.einsum("oc,bct->bot", w[..., j], xp[..., j * d:j * d + T]) for j in range(k))
    assert np.abs(y - direct).max() <= 1e-10   # (exact in real arithmetic; the transform constants reach 8)


def test_products_per_output_quad():
    assert [len(virtual_taps4(k, 0)) + len(virtual_taps4(k, 1)) for k in (3, 7, 11)] == [6, 16, 26]   # against 12 / 28 / 44 for the direct sum
    assert all(len(virtual_taps4(k, 0)) == len(virtual_taps4(k, 1)) for k in (3, 7, 11))             # the two halves stay in step


# ---- conv_wino44_impl.h: F(4,4) tap groups (points ±1/2, ±1, ±2, ∞) on the quad lattice; seven planes over two halves, the ∞ plane shared by channel pairs ----
G44 = np.array([[16 / 45, 8 / 45, 4 / 45, 2 / 45], [-16 / 45, 8 / 45, -4 / 45, 2 / 45], [-2 / 9, -2 / 9, -2 / 9, -2 / 9], [2 / 9, -2 / 9, 2 / 9, -2 / 9],
                [1 / 45, 2 / 45, 4 / 45, 8 / 45], [-1 / 45, 2 / 45, -4 / 45, 8 / 45]])


def wino44_conv1d(x, w, d):
    B, C, T = x.shape
    k = w.shape[-1]
    assert C % 8 == 0
    pad = (k - 1) * d // 2
    ng = (k + 3) // 4
    wz = np.zeros(w.shape[:2] + (4 * ng,))
    wz[..., :k] = w                              # the taps past k are zero
    nq = -(-T // (4 * d))
    NP = nq * d
    halo = d * (ng - 1) + d
    xp = np.zeros((B, C, 4 * d * (nq + 1) + 4 * halo + pad))
    xp[..., pad:pad + T] = x
    n = np.arange(NP + halo)
    t0 = 4 * d * (n // d) + n % d
    x0, x1, x2, x3 = (xp[..., t0 + j * d] for j in range(4))
    x4, x5, x6 = (np.roll(v, -d, axis=-1) for v in (x0, x1, x2))
    V = []
    for a, (c0, c1) in ((0.5, (4.0, -5.0)), (1.0, (1.0, -4.25)), (2.0, (0.25, -1.25))):
        E, O = x4 + c1 * x2 + c0 * x0, x5 + c1 * x3 + c0 * x1
        V += [O + a * E, O - a * E]
    Vinf = (x6 - x0) + 5.25 * (x2 - x4)
    pair_of_channel = (np.arange(C) % 8) // 2     # channel pair inside its 8-channel block: half h takes the ∞ plane's products of pairs 2 h, 2 h + 1
    acc = []
    for h in (0, 1):
        m = [np.zeros((B, w.shape[0], NP)) for _ in range(4)]
        for g in range(ng):
            for i in range(3):
                U = np.einsum("t,oct->oc", G44[3 * h + i], wz[..., 4 * g:4 * g + 4])
                m[i] += np.einsum("oc,bcn->bon", U, V[3 * h + i][..., g * d:g * d + NP])
            if g < ng - 1:                       # U(∞) = the group's fourth tap: absent (zero) in the last group of k = 7 / 11
                mine = (pair_of_channel // 2) == h
                m[3] += np.einsum("oc,bcn->bon", wz[..., 4 * g + 3][:, mine], Vinf[:, mine, g * d:g * d + NP])
        acc.append(m)
    (mh, mmh, m1, ia), (mm1, m2, mm2, ib) = acc
    s, df = mh + mmh, mh - mmh                   # half 0 keeps y0, y1 and sends its parts of y2, y3
    keepA, sendA = [s + m1, 0.5 * df + m1], [0.25 * s + m1, 0.125 * df + m1 + ia]
    s, df = m2 + mm2, m2 - mm2                   # half 1 sends its parts of y0, y1 and keeps y2, y3
    sendB, keepB = [s + mm1, 2 * df - mm1], [4 * s + mm1, 8 * df - mm1 + ib]
    y = np.zeros((B, w.shape[0], 4 * d * nq))
    tt = 4 * d * (np.arange(NP) // d) + np.arange(NP) % d
    for j in range(2):
        y[..., tt + j * d] = keepA[j] + sendB[j]
        y[..., tt + (2 + j) * d] = keepB[j] + sendA[j]
    return y[..., :T]


@pytest.mark.parametrize("k,d,T", [(7, 1, 64), (7, 3, 1), (7, 5, 333), (11, 1, 129), (11, 3, 50), (11, 5, 9), (11, 5, 700), (11, 1, 3)])
def test_quad_lattice_f44_winograd_equals_the_direct_conv(k, d, T):
    rng = np.random.default_rng(k * 100 + d * 10 + T)
    x = rng.normal(size=(2, 8, T)).astype(np.float32).astype(np.float64)
    w = rng.normal(size=(5, 8, k)).astype(np.float32).astype(np.float64)
    y = wino44_conv1d(x, w, d)
    xp = np.pad(x, ((0, 0), (0, 0), ((k - 1) * d // 2,) * 2))
    direct = sum(np.einsum("oc,bct->bot", w[..., j], xp[..., j * d:j * d + T]) for j in range(k))
    assert y.shape == direct.shape
    assert np.abs(y - direct).max() <= 1e-10   # exact in real arithmetic
    ref = orc.conv1d(x.astype(np.float32), w.astype(np.float32), None, dilation=d, padding=(k - 1) * d // 2)
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_products_per_output_quad_f44():
    # per half and two channels: 3 planes x ng groups + the ∞ plane's (ng - 1) products on half of the channel pairs
    assert [2 * (3 * ng + (ng - 1) / 2) for ng in (2, 3)] == [13, 20]     # k = 7, 11: against 16 / 26 for F(4,3) and 28 / 44 for the direct sum
