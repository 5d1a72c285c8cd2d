"""GPU: the fused MFMA conv kernel (through the C ABI, fv_conv_*) against the CPU oracle on seeded inputs.
Tolerance: |d| <= 1e-4 absolute (north_star) — in practice ~1e-6 since the MFMA path is exact fp32."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev():
    assert torch.cuda.is_available(), "GPU test on a box without a GPU"
    return torch.device("cuda:0")


def _run(w, b, x, res=None, **kw):
    from vocoder_amd.engine import FusedConv
    conv = FusedConv(w, b, **kw)
    xt = torch.from_numpy(x).to(_dev())
    rt = None if res is None else torch.from_numpy(res).to(_dev())
    y = conv(xt, rt)
    torch.cuda.synchronize()
    return y.cpu().numpy()


def _last_kernel():
    from vocoder_amd import _lib
    return _lib.last_kernel()


def _check(y, ref, atol=1e-4):
    assert y.shape == ref.shape, (y.shape, ref.shape)
    err = np.abs(y - ref).max()
    scale = np.abs(ref).max()
    assert err <= atol, f"max|d|={err:.3e} (ref max {scale:.3f})"
    assert err <= 2e-5 * max(scale, 1.0), f"max|d|={err:.3e} looks too large for an exact-fp32 path (ref max {scale:.3f})"


CONV_CASES = [
    # (Cin, Cout, k, dil, B, T)
    (256, 256, 3, 1, 2, 200), (256, 256, 7, 3, 1, 130), (256, 256, 11, 5, 1, 97),
    (128, 128, 3, 3, 2, 300), (128, 128, 11, 1, 1, 517),
    (64, 64, 7, 5, 2, 700), (64, 64, 3, 5, 1, 1025),
    (32, 32, 11, 3, 2, 1500), (32, 32, 7, 1, 1, 640),
    (16, 16, 3, 1, 3, 2100), (16, 16, 11, 5, 1, 515),
    (80, 512, 7, 1, 2, 86),          # conv_pre of HiFiGAN-V1 (C_in not a multiple of 8 chunks? 80 = 10 chunks)
    (10, 64, 5, 1, 2, 33),           # odd C_in -> zero-padded channel chunk, generic (k=5) kernel
    (6, 4, 5, 2, 2, 61), (2, 2, 3, 2, 1, 9), (1, 3, 13, 1, 1, 40),
    (512, 512, 13, 1, 1, 50),        # firefly pre conv
    (128, 512, 1, 1, 2, 94), (512, 128, 1, 1, 2, 94),   # ConvNeXt pointwise
    # enough columns for the flattened (batch, time) tiles: even T stages column pairs, odd T single columns;
    # C_in = 100 ends in a partial 32-channel chunk
    (100, 512, 1, 1, 40, 94), (512, 128, 1, 1, 70, 47), (128, 256, 1, 1, 64, 50), (64, 1026, 1, 1, 33, 98),
    (96, 2048, 1, 1, 48, 94), (72, 1280, 1, 1, 60, 47),   # 256 x 64 tiles (an even number of 128-row blocks)
    (256, 512, 1, 1, 70, 94), (40, 768, 1, 1, 35, 63),     # 256 x 32 tiles
    # single-clip latency kernel (K split over the four waves of a 32-row tile): a partial 32-channel chunk (40, 72, 200
    # channels), three row blocks, ragged last column tile; lengths that take the 32 x 64 tile instead of 32 x 32
    (40, 64, 3, 1, 1, 200), (72, 96, 7, 3, 1, 150), (200, 200, 11, 5, 1, 97),
    (128, 128, 7, 3, 1, 3000), (64, 64, 11, 5, 1, 5000), (128, 128, 3, 1, 1, 2500),
]


@pytest.mark.parametrize("cin,cout,k,d,B,T", CONV_CASES)
def test_conv1d_silu_residual_matches_oracle(cin, cout, k, d, B, T):
    from vocoder_amd import _lib
    rng = np.random.default_rng(cin * 1000 + cout + k * 7 + d)
    x = rng.normal(size=(B, cin, T)).astype(np.float32)
    w = (rng.normal(size=(cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)   # asymmetric, transpose-detecting
    b = rng.normal(size=cout).astype(np.float32)
    pad = (k * d - d) // 2
    ref = orc.conv1d(orc.silu(x), w, b, dilation=d, padding=pad)
    res = rng.normal(size=ref.shape).astype(np.float32)
    y = _run(w, b, x, res, dilation=d, padding=pad, pre_act=_lib.FV_ACT_SILU)
    _check(y, ref + res)
    # plain conv, no activation / residual / bias
    y2 = _run(w, None, x, None, dilation=d, padding=pad)
    _check(y2, orc.conv1d(x, w, None, dilation=d, padding=pad))


def test_conv1d_post_activations():
    from vocoder_amd import _lib
    rng = np.random.default_rng(5)
    x = rng.normal(size=(2, 32, 300)).astype(np.float32)
    w = (rng.normal(size=(32, 32, 3)) / 8).astype(np.float32)
    b = rng.normal(size=32).astype(np.float32)
    lin = orc.conv1d(x, w, b, padding=1)
    _check(_run(w, b, x, padding=1, post_act=_lib.FV_ACT_SILU), orc.silu(lin))
    _check(_run(w, b, x, padding=1, post_act=_lib.FV_ACT_GELU), orc.gelu(lin))
    _check(_run(w, b, x, padding=1, post_act=_lib.FV_ACT_TANH), orc.tanh(lin))
    _check(_run(w, b, x, padding=1, pre_act=_lib.FV_ACT_LEAKY_RELU, act_slope=0.2),
           orc.conv1d(orc.leaky_relu(x, 0.2), w, b, padding=1))


CONVT_CASES = [
    # (Cin, Cout, k, u, B, T)
    (512, 256, 16, 8, 2, 86), (256, 128, 16, 8, 1, 100), (128, 64, 8, 2, 2, 300), (64, 32, 2, 2, 1, 700),
    (32, 16, 2, 2, 2, 1111), (128, 64, 4, 2, 1, 257), (64, 32, 4, 4, 1, 50), (8, 4, 4, 2, 2, 17), (4, 2, 7, 3, 1, 11),
    (16, 8, 3, 2, 1, 21),
]


@pytest.mark.parametrize("cin,cout,k,u,B,T", CONVT_CASES)
def test_conv_transpose1d_matches_oracle(cin, cout, k, u, B, T):
    from vocoder_amd import _lib
    rng = np.random.default_rng(cin + cout * 3 + k + u)
    x = rng.normal(size=(B, cin, T)).astype(np.float32)
    w = (rng.normal(size=(cin, cout, k)) / np.sqrt(cin * max(k // u, 1))).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    pad = (k - u) // 2
    ref = orc.conv_transpose1d(orc.silu(x), w, b, stride=u, padding=pad)
    y = _run(w, b, x, transposed=True, stride=u, padding=pad, pre_act=_lib.FV_ACT_SILU)
    _check(y, ref)


def test_golden_ops_through_hip():
    """The same golden vectors that pin the oracle (captured from torch / the reference), straight through HIP."""
    from conftest import load_golden
    g = load_golden("ops.npz")
    for k, d in ((3, 1), (7, 3), (11, 5)):
        y = _run(g[f"conv_k{k}d{d}_w"], g[f"conv_k{k}d{d}_b"], g[f"conv_k{k}d{d}_x"], dilation=d, padding=(k * d - d) // 2)
        _check(y, g[f"conv_k{k}d{d}_y"])
    for k, u in ((16, 8), (8, 2), (2, 2), (4, 4), (4, 2)):
        y = _run(g[f"convT_k{k}u{u}_w"], g[f"convT_k{k}u{u}_b"], g[f"convT_k{k}u{u}_x"], transposed=True, stride=u,
                 padding=(k - u) // 2)
        _check(y, g[f"convT_k{k}u{u}_y"])


def test_linearity_at_full_size():
    """Size-independent property at the BASELINE stage-1 shape (C=128, T=5504, B=4): conv(a*x1 + x2) - conv(x2)
    == a * (conv(x1) - bias-free) to fp32 round-off; checks the large-grid tiling paths the oracle is too slow for."""
    rng = np.random.default_rng(0)
    C, T, B, k, d = 128, 5504, 4, 7, 3
    w = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    x1 = rng.normal(size=(B, C, T)).astype(np.float32)
    x2 = rng.normal(size=(B, C, T)).astype(np.float32)
    pad = (k * d - d) // 2
    y1 = _run(w, None, x1, dilation=d, padding=pad)
    y2 = _run(w, None, x2, dilation=d, padding=pad)
    y12 = _run(w, None, (0.5 * x1 + x2).astype(np.float32), dilation=d, padding=pad)
    # three rounded results meet in the difference; the launch takes the F(4,4) Winograd kernel (conv_wino44_impl.h: ~1.5 x a direct sum's round-off)
    assert np.abs(y12 - (0.5 * y1 + y2)).max() <= 8e-6 * max(1.0, np.abs(y12).max())
    # spot-check a slab against the oracle
    ref = orc.conv1d(x1[:1, :, :700], w, None, dilation=d, padding=pad)
    _check(y1[:1, :, :600], ref[:, :, :600])
    assert np.abs(y1[:1, :, :600] - ref[:, :, :600]).max() <= 4e-6 * np.abs(ref).max()


def test_errors_are_reported():
    from vocoder_amd.engine import FusedConv, FishVocError
    w = np.zeros((4, 4, 3), np.float32)
    conv = FusedConv(w, None, padding=1)
    with pytest.raises(RuntimeError):
        conv(torch.zeros(1, 4, 8))            # CPU tensor: no fallback
    with pytest.raises(ValueError):
        conv(torch.zeros(1, 5, 8, device=_dev()))
    with pytest.raises(FishVocError):
        FusedConv(np.zeros((4, 4, 3), np.float32), None, padding=-1)


PAIR_CASES = [
    # (C, k, d, B, T) — T chosen to hit: several tiles, a ragged last tile, T smaller than one tile, T == 1
    (16, 3, 1, 2, 2100), (16, 7, 3, 1, 1030), (16, 11, 5, 2, 700), (16, 11, 1, 1, 502), (16, 3, 5, 1, 1),
    (64, 3, 1, 2, 400), (64, 3, 5, 1, 1033), (64, 3, 3, 1, 2),
    (128, 3, 1, 2, 300), (128, 3, 5, 1, 517), (128, 3, 3, 1, 3),     # waves along M, residual from HBM (round 3)
    (32, 3, 3, 2, 1000), (32, 7, 5, 1, 517), (32, 11, 1, 2, 300), (32, 11, 5, 1, 247), (32, 7, 1, 1, 5),
]


@pytest.mark.parametrize("C,k,d,B,T", PAIR_CASES)
def test_fused_resblock_pair_matches_oracle(C, k, d, B, T):
    """y = x + c2(silu(c1(silu(x)))) — one ResBlock1 iteration (reference hifigan.py:102-107) in one launch."""
    from vocoder_amd.engine import FusedConv
    rng = np.random.default_rng(C * 100 + k * 10 + d)
    x = rng.normal(size=(B, C, T)).astype(np.float32)
    w1 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    w2 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    b1 = rng.normal(size=C).astype(np.float32)
    b2 = rng.normal(size=C).astype(np.float32)
    xt = orc.conv1d(orc.silu(x), w1, b1, dilation=d, padding=(k * d - d) // 2)
    ref = x + orc.conv1d(orc.silu(xt), w2, b2, padding=(k - 1) // 2)
    c1 = FusedConv(w1, b1, dilation=d, padding=(k * d - d) // 2)
    c2 = FusedConv(w2, b2, padding=(k - 1) // 2)
    y = c1.pair(c2, torch.from_numpy(x).to(_dev()))
    torch.cuda.synchronize()
    _check(y.cpu().numpy(), ref)


PAIR_WINO_CASES = [(C, k, d) for C in (16, 32) for k in (3, 7, 11) for d in (1, 3, 5)]


@pytest.mark.parametrize("C,k,d,form", [(C, k, d, f) for (C, k, d) in PAIR_WINO_CASES for f in (("f44", "f23") if k >= 7 else ("f44",))])
def test_winograd_pair_matches_oracle_and_direct_pair(C, k, d, form, monkeypatch):
    """Every (C, k, dilation) variant of the Winograd fused pairs on ragged shapes — several tiles with a ragged last one, a row shorter than one tile,
    T = 1, odd T (no 8-byte stores), a length that ends on a partly valid quad — against the CPU oracle (reference hifigan.py:102-107), and against the
    direct-sum pair kernel selected through the C ABI (fv_conv_set_algorithm): same outputs to fp32 rounding, other sums.  form f44 (the default): k = 7 / 11
    on F(4,4) tap groups (pair_wino44_impl.h, round 5), k = 3 on F(2,3); f23 (FV_PAIR_WINO44=0): F(2,3) everywhere (pair_wino_impl.h)."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import FusedConv
    rng = np.random.default_rng(7 * C + 13 * k + d)
    w1 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    w2 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    b1 = rng.normal(size=C).astype(np.float32)
    b2 = rng.normal(size=C).astype(np.float32)
    if form == "f23":
        monkeypatch.setenv("FV_PAIR_WINO44", "0")
    _lib.reload_env()
    try:
        c1 = FusedConv(w1, b1, dilation=d, padding=(k * d - d) // 2)
        c2 = FusedConv(w2, b2, padding=(k - 1) // 2)
        want = "pair_wino44<" if (form == "f44" and k >= 7) else "pair_wino<"
        for B, T in ((2, 1000 + 37 * d), (1, 61), (3, 1), (1, 2 * 128 - k), (2, 1302), (1, 3)):
            x = rng.normal(size=(B, C, T)).astype(np.float32)
            xt = orc.conv1d(orc.silu(x), w1, b1, dilation=d, padding=(k * d - d) // 2)
            ref = x + orc.conv1d(orc.silu(xt), w2, b2, padding=(k - 1) // 2)
            xd = torch.from_numpy(x).to(_dev())
            y = c1.set_algorithm("auto").pair(c2, xd)
            torch.cuda.synchronize()
            assert _last_kernel().startswith(want), (_last_kernel(), want)
            _check(y.cpu().numpy(), ref)
            yd = c1.set_algorithm("direct").pair(c2, xd)
            torch.cuda.synchronize()
            assert _last_kernel().startswith("resblock_pair<"), _last_kernel()
            assert float((y - yd).abs().max()) <= 2e-5 * max(1.0, float(np.abs(ref).max()))
    finally:
        monkeypatch.delenv("FV_PAIR_WINO44", raising=False)
        _lib.reload_env()


@pytest.mark.parametrize("C,k,d", [(32, 7, 1), (32, 11, 3), (32, 11, 5), (16, 11, 1)])
def test_winograd_pair_reads_its_residual_from_a_4_byte_aligned_input(C, k, d):
    """pair_wino44 at C = 32 re-reads the residual x from global memory (round 6: the second V buffer took the raw tile's place in LDS; hifigan.py:107
    `x = xt + x`) — as 8-byte loads when T is even and x is 8-byte aligned, as dwords otherwise.  The same clip from a buffer that starts 4 bytes off an
    8-byte boundary must give the same bits (even T: the stores stay 8-byte, only the residual path changes), and both must match the oracle."""
    from vocoder_amd.engine import FusedConv
    rng = np.random.default_rng(11 * C + k + d)
    B, T = 2, 1302
    w1 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    w2 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    b1 = rng.normal(size=C).astype(np.float32)
    b2 = rng.normal(size=C).astype(np.float32)
    x = rng.normal(size=(B, C, T)).astype(np.float32)
    c1 = FusedConv(w1, b1, dilation=d, padding=(k * d - d) // 2)
    c2 = FusedConv(w2, b2, padding=(k - 1) // 2)
    xa = torch.from_numpy(x).to(_dev())
    flat = torch.zeros(B * C * T + 3, device=_dev())
    off = 1 if flat.data_ptr() % 8 == 0 else 2        # (the view below starts 4 bytes past an 8-byte boundary either way)
    if (flat.data_ptr() + 4 * off) % 8 == 0:
        off += 1
    xm = flat[off:off + B * C * T].view(B, C, T)
    xm.copy_(xa)
    assert xm.is_contiguous() and xm.data_ptr() % 8 == 4
    ya = c1.pair(c2, xa)
    ym = c1.pair(c2, xm)
    torch.cuda.synchronize()
    assert _last_kernel().startswith("pair_wino44<"), _last_kernel()
    assert torch.equal(ya, ym)
    xt = orc.conv1d(orc.silu(x), w1, b1, dilation=d, padding=(k * d - d) // 2)
    _check(ya.cpu().numpy(), x + orc.conv1d(orc.silu(xt), w2, b2, padding=(k - 1) // 2))


def test_fused_pair_rejects_unsupported_shapes():
    from vocoder_amd.engine import FusedConv, FishVocError
    w = np.zeros((64, 64, 5), np.float32)
    c = FusedConv(w, None, padding=2)
    with pytest.raises(FishVocError, match="unsupported pair"):
        c.pair(c, torch.zeros(1, 64, 32, device=_dev()))


# ---- the LDS-free persistent GEMM of the pointwise convs (gemm_pw.hip): every configuration, against the oracle -------------
PW_CASES = [
    # (Cin, Cout, B, T): even T -> 8-byte column pairs; odd T -> single columns; ragged rows / columns; more waves than tiles
    (512, 2048, 16, 94), (2048, 512, 16, 94), (128, 512, 40, 47), (256, 96, 9, 33), (64, 1026, 33, 98), (72, 200, 5, 7),
    (1024, 256, 3, 94), (64, 32, 2, 5),
]


def _reload_env_later():
    # registered BEFORE monkeypatch's own teardown runs?  No: finalizers run last-in-first-out and monkeypatch (a fixture
    # requested earlier) is torn down after this one — so drop the variable here ourselves, then re-read.
    import os
    from vocoder_amd import _lib
    os.environ.pop("FV_PW", None)
    _lib.reload_env()


@pytest.mark.parametrize("cfg", ["0", "1", None])
@pytest.mark.parametrize("cin,cout,B,T", PW_CASES)
def test_pointwise_gemm_every_configuration_matches_oracle(cin, cout, B, T, cfg, monkeypatch, request):
    """Linear -> (+bias) [-> +residual] [-> GELU] of the ConvNeXt block (convnext.py:130-141).  cfg = forced kernel
    configuration (FV_PW), None = the host's own choice (which may be the general conv kernel for tiny launches)."""
    from vocoder_amd import _lib
    if cfg is None:
        monkeypatch.delenv("FV_PW", raising=False)
    else:
        monkeypatch.setenv("FV_PW", cfg)
    _lib.reload_env()                       # the library caches its knobs: re-read now ...
    request.addfinalizer(_reload_env_later)  # ... and again once monkeypatch has restored the environment
    rng = np.random.default_rng(cin + 3 * cout + B + T)
    x = rng.normal(size=(B, cin, T)).astype(np.float32)
    w = (rng.normal(size=(cout, cin, 1)) / np.sqrt(cin)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(B, cout, T)).astype(np.float32)
    lin = orc.conv1d(x, w, b)
    y = _run(w, b, x, post_act=_lib.FV_ACT_GELU)
    if cfg is not None and cin >= 64:
        assert _lib.last_kernel().startswith("gemm_pw<"), _lib.last_kernel()
    _check(y, orc.gelu(lin))
    _check(_run(w, b, x, res), lin + res)


def test_fast_gelu_of_the_pointwise_gemm_against_the_exact_function():
    """gemm_pw.hip evaluates GELU with the erfc form of Abramowitz & Stegun 7.1.26 (16 VALU instructions instead of erff):
    pinned here against float64 erf over the whole useful range, bar |err| <= 1e-6 absolute (torch's own fp32 nn.GELU is
    off by up to 1.2e-6 over the same range)."""
    import math
    from vocoder_amd import _lib
    n = 64 * 94 * 2
    v = np.linspace(-12.0, 12.0, n).astype(np.float32)
    # identity weights: y[c] = x[c]; one input row carries the sweep, the others random values
    cin = 64
    x = np.zeros((2, cin, n // 2), np.float32)
    x[:, 0, :] = v.reshape(2, -1)
    x[:, 1:, :] = np.random.default_rng(0).normal(size=(2, cin - 1, n // 2)).astype(np.float32) * 3
    w = np.eye(cin, dtype=np.float32)[:, :, None]
    import os
    os.environ["FV_PW"] = "1"
    _lib.reload_env()
    try:
        y = _run(w, np.zeros(cin, np.float32), x, post_act=_lib.FV_ACT_GELU)
        assert _lib.last_kernel().startswith("gemm_pw<"), _lib.last_kernel()
    finally:
        del os.environ["FV_PW"]
        _lib.reload_env()
    x64 = x.astype(np.float64)
    erf = np.vectorize(math.erf)
    exact = 0.5 * x64 * (1.0 + erf(x64 / math.sqrt(2.0)))
    err = np.abs(y - exact)
    assert err.max() <= 1e-6, err.max()
    assert err[:, 0, :].max() <= 1e-6


WINO_CASES = [
    # (C, k, dil, B, T): ragged lengths (odd, below one block of 2 D samples, one sample), several row blocks, every tile variant
    (128, 11, 1, 1, 517), (128, 7, 3, 2, 300), (128, 3, 5, 1, 131), (256, 11, 5, 1, 97), (256, 3, 1, 2, 200), (192, 7, 5, 1, 1000),
    (64, 11, 3, 2, 700), (64, 7, 1, 1, 129), (64, 3, 3, 1, 64), (40, 7, 5, 1, 333), (128, 11, 5, 1, 9), (128, 7, 3, 1, 1),
    (32, 11, 5, 2, 900), (32, 7, 3, 1, 77), (96, 11, 1, 1, 255), (32, 3, 1, 1, 2),
]


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2])
def test_winograd_conv_matches_oracle(cfg, monkeypatch):
    """conv_wino_impl.h (F(2,3) tap groups; FV_WINO4=0 keeps the quad-lattice kernels out) through fv_conv_* with the kernel forced (FV_WINO=2) and every
    tile variant (FV_WINO_CFG): SiLU + bias + residual and the plain conv against the CPU oracle; the direct-sum kernel on the same layer for
    scale (both sit ~1e-6 from the oracle)."""
    from vocoder_amd import _lib
    monkeypatch.setenv("FV_WINO", "2")
    monkeypatch.setenv("FV_WINO4", "0")
    monkeypatch.setenv("FV_WINO_CFG", str(cfg))
    _lib.reload_env()
    try:
        for (c, k, d, B, T) in WINO_CASES:
            rng = np.random.default_rng(c * 1000 + k * 7 + d + T)
            x = rng.normal(size=(B, c, T)).astype(np.float32)
            w = (rng.normal(size=(c, c, k)) / np.sqrt(c * k)).astype(np.float32)
            b = rng.normal(size=c).astype(np.float32)
            pad = (k - 1) * d // 2
            ref = orc.conv1d(orc.silu(x), w, b, dilation=d, padding=pad)
            res = rng.normal(size=ref.shape).astype(np.float32)
            y = _run(w, b, x, res, dilation=d, padding=pad, pre_act=_lib.FV_ACT_SILU)
            assert _lib.last_kernel().startswith("conv_wino<"), (_lib.last_kernel(), c, k, d)
            _check(y, ref + res)
            y2 = _run(w, None, x, None, dilation=d, padding=pad)
            _check(y2, orc.conv1d(x, w, None, dilation=d, padding=pad))
    finally:
        monkeypatch.delenv("FV_WINO")
        monkeypatch.delenv("FV_WINO4")
        monkeypatch.delenv("FV_WINO_CFG")
        _lib.reload_env()


WINO4_CASES = [
    # (C, k, dil, B, T): whole 64-row blocks, k = 7 / 11; ragged lengths (odd, below one block of 4 D samples, one sample), one to four row blocks
    (128, 11, 1, 1, 517), (128, 7, 3, 2, 300), (256, 11, 5, 1, 97), (192, 7, 5, 1, 1000), (64, 11, 3, 2, 700), (64, 7, 1, 1, 129), (128, 11, 5, 1, 9),
    (128, 7, 3, 1, 1), (64, 11, 1, 3, 4099), (64, 7, 5, 1, 19), (128, 11, 3, 1, 12), (128, 7, 1, 2, 127), (256, 7, 1, 1, 33),
]


def test_winograd_f43_conv_matches_oracle(monkeypatch):
    """conv_wino4_impl.h — Winograd F(4,3) tap groups on the dilated quad lattice (26 / 16 matrix products per four outputs; the predecessor of the F(4,4)
    kernel, reached with FV_WINO44=0) — through fv_conv_* with the kernel forced (FV_WINO=2): SiLU + bias + residual, the plain conv and a leaky-ReLU
    conv with a SiLU behind it against the CPU oracle; the accumulate epilogue runs in the model tests (the MRF sum, test_gpu_models.py)."""
    from vocoder_amd import _lib
    monkeypatch.setenv("FV_WINO", "2")
    monkeypatch.setenv("FV_WINO44", "0")
    _lib.reload_env()
    try:
        for (c, k, d, B, T) in WINO4_CASES:
            rng = np.random.default_rng(c * 1000 + k * 7 + d + T)
            x = rng.normal(size=(B, c, T)).astype(np.float32)
            w = (rng.normal(size=(c, c, k)) / np.sqrt(c * k)).astype(np.float32)
            b = rng.normal(size=c).astype(np.float32)
            pad = (k - 1) * d // 2
            ref = orc.conv1d(orc.silu(x), w, b, dilation=d, padding=pad)
            res = rng.normal(size=ref.shape).astype(np.float32)
            y = _run(w, b, x, res, dilation=d, padding=pad, pre_act=_lib.FV_ACT_SILU)
            want = f"conv_wino4<k={k} d={d} tile=64x32q>"
            if _lib.last_kernel().startswith("conv_wino<"):
                pytest.skip("libfishvoc_hip.so was built without the A/B partner kernels (make ABPARTNERS=1 adds conv_wino4; the shipped build selects F(2,3) here)")
            assert _lib.last_kernel() == want, (_lib.last_kernel(), want)
            _check(y, ref + res)
            y2 = _run(w, None, x, None, dilation=d, padding=pad)
            assert _lib.last_kernel() == want
            _check(y2, orc.conv1d(x, w, None, dilation=d, padding=pad))
            if T in (517, 300, 700, 19):
                y3 = _run(w, b, x, None, dilation=d, padding=pad, pre_act=_lib.FV_ACT_LEAKY_RELU, post_act=_lib.FV_ACT_SILU, act_slope=0.1)
                assert _lib.last_kernel() == want
                _check(y3, orc.silu(orc.conv1d(np.where(x >= 0, x, np.float32(0.1) * x), w, b, dilation=d, padding=pad)))
    finally:
        monkeypatch.delenv("FV_WINO")
        monkeypatch.delenv("FV_WINO44")
        _lib.reload_env()


WINO44_CASES = [
    # (C, k, dil, B, T): whole 64-row tiles, k = 7 / 11; ragged lengths (odd, below one block of 4 D samples, one sample), one to four row blocks, both chunk sizes
    (128, 11, 1, 1, 517), (128, 7, 3, 2, 300), (256, 11, 5, 1, 97), (192, 7, 5, 1, 1000), (64, 11, 3, 2, 700), (64, 7, 1, 1, 129), (128, 11, 5, 1, 9),
    (128, 7, 3, 1, 1), (64, 11, 1, 3, 4099), (64, 7, 5, 1, 19), (128, 11, 3, 1, 12), (128, 7, 1, 2, 127), (256, 7, 1, 1, 33), (320, 11, 1, 1, 70),
]


@pytest.mark.parametrize("rows", [64, 128])
def test_winograd_f44_conv_matches_oracle(rows, monkeypatch):
    """conv_wino44_impl.h — Winograd F(4,4) tap groups on the dilated quad lattice (20 / 13 matrix products per four outputs; seven planes over two waves,
    the ∞ plane shared by channel pairs) — through fv_conv_* with the kernel forced (FV_WINO=2; FV_WINO44_ROWS: one or two 32-row tiles per wave): SiLU +
    bias + residual, the plain conv and a leaky-ReLU conv with a SiLU behind it against the CPU oracle."""
    from vocoder_amd import _lib
    monkeypatch.setenv("FV_WINO", "2")
    monkeypatch.setenv("FV_WINO44_ROWS", str(rows))   # (128 is the default where the layer has whole 128-row blocks)
    _lib.reload_env()
    try:
        for (c, k, d, B, T) in WINO44_CASES:
            rng = np.random.default_rng(c * 1000 + k * 7 + d + T)
            x = rng.normal(size=(B, c, T)).astype(np.float32)
            w = (rng.normal(size=(c, c, k)) / np.sqrt(c * k)).astype(np.float32)
            b = rng.normal(size=c).astype(np.float32)
            pad = (k - 1) * d // 2
            ref = orc.conv1d(orc.silu(x), w, b, dilation=d, padding=pad)
            res = rng.normal(size=ref.shape).astype(np.float32)
            y = _run(w, b, x, res, dilation=d, padding=pad, pre_act=_lib.FV_ACT_SILU)
            want = f"conv_wino44<k={k} d={d} tile={rows if c % 128 == 0 else 64}x32q>"
            assert _lib.last_kernel() in (want, want + " flat"), (_lib.last_kernel(), want)
            _check(y, ref + res)
            y2 = _run(w, None, x, None, dilation=d, padding=pad)
            assert _lib.last_kernel() in (want, want + " flat")
            _check(y2, orc.conv1d(x, w, None, dilation=d, padding=pad))
            if T in (517, 300, 700, 19):
                y3 = _run(w, b, x, None, dilation=d, padding=pad, pre_act=_lib.FV_ACT_LEAKY_RELU, post_act=_lib.FV_ACT_SILU, act_slope=0.1)
                assert _lib.last_kernel() in (want, want + " flat")
                _check(y3, orc.silu(orc.conv1d(np.where(x >= 0, x, np.float32(0.1) * x), w, b, dilation=d, padding=pad)))
    finally:
        monkeypatch.delenv("FV_WINO")
        monkeypatch.delenv("FV_WINO44_ROWS")
        _lib.reload_env()


FLAT_CASES = [
    # (C, k, dil, B, T): batches whose clips are not a whole number of 32-column tiles — the flattened column axis needs fewer tiles than per-clip tiling
    (256, 11, 1, 4, 688), (256, 7, 1, 3, 688), (128, 7, 3, 3, 300), (128, 11, 5, 5, 260), (128, 11, 1, 2, 516), (384, 7, 5, 6, 130), (128, 11, 3, 7, 40),
    (128, 11, 1, 4, 4), (256, 7, 3, 5, 8), (128, 7, 5, 3, 1), (128, 11, 5, 9, 23),   # clips shorter than a tile, than one 4 D block, one sample
    # ... and shapes the flattened instances do not cover (64-row workgroups; rows that are no whole quads at D = 1): per-clip tiles under either setting
    (64, 11, 5, 5, 260), (192, 7, 5, 6, 130), (128, 11, 1, 2, 517),
]


@pytest.mark.parametrize("c,k,d,B,T", FLAT_CASES)
def test_winograd_f44_flattened_columns_equal_per_clip_tiles(c, k, d, B, T, monkeypatch):
    """conv_wino44 over ONE axis of all clips' quad columns (each clip followed by NG D columns of its own halo; conv_layer.hip) against the same kernel
    tiling every clip on its own (FV_WINO44_FLAT=0): bit for bit — an output's sum does not depend on the tile it sits in — and against the CPU oracle;
    with and without the residual (the row-split 16-byte epilogue at D = 1, the lean one at D > 1)."""
    from vocoder_amd import _lib
    rng = np.random.default_rng(c * 1000 + k * 7 + d + T + B)
    x = rng.normal(size=(B, c, T)).astype(np.float32)
    w = (rng.normal(size=(c, c, k)) / np.sqrt(c * k)).astype(np.float32)
    b = rng.normal(size=c).astype(np.float32)
    pad = (k - 1) * d // 2
    ref = orc.conv1d(orc.silu(x), w, b, dilation=d, padding=pad)
    res = rng.normal(size=ref.shape).astype(np.float32)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FV_WINO", "2")
        monkeypatch.setenv("FV_WINO44_FLAT", mode)
        _lib.reload_env()
        try:
            y = _run(w, b, x, res, dilation=d, padding=pad, pre_act=_lib.FV_ACT_SILU)
            name = _lib.last_kernel()
            assert name.startswith("conv_wino44<"), name
            covered = c % 128 == 0 and (d != 1 or T % 4 == 0)   # (128-row workgroups; D = 1: the row-split epilogue's whole quads)
            assert name.endswith(" flat") == (mode == "1" and covered), (name, mode)
            y2 = _run(w, b, x, None, dilation=d, padding=pad, pre_act=_lib.FV_ACT_SILU, post_act=_lib.FV_ACT_SILU)
            got[mode] = (y, y2)
        finally:
            monkeypatch.delenv("FV_WINO")
            monkeypatch.delenv("FV_WINO44_FLAT")
            _lib.reload_env()
    _check(got["1"][0], ref + res)
    _check(got["1"][1], orc.silu(ref))
    assert np.array_equal(got["1"][0], got["0"][0])
    assert np.array_equal(got["1"][1], got["0"][1])


LAT_CASES = [
    # (C, k, dil, B, T): single clips / small batches below the Winograd gate; ragged lengths; both tile widths (16 / 32 pairs)
    (256, 11, 1, 1, 688), (256, 11, 5, 1, 97), (256, 7, 3, 1, 344), (256, 3, 1, 2, 200), (128, 11, 1, 1, 5504), (128, 11, 3, 1, 517),
    (128, 7, 5, 1, 2048), (128, 3, 5, 1, 131), (64, 11, 5, 1, 11008), (64, 7, 1, 1, 129), (64, 3, 3, 1, 64), (192, 7, 5, 1, 1000),
    (96, 11, 1, 1, 255), (32, 11, 5, 1, 900), (32, 7, 3, 1, 77), (32, 3, 1, 1, 2), (128, 11, 5, 1, 9), (128, 7, 3, 1, 1),
]


@pytest.mark.parametrize("form", ["f44", "f23"])
def test_winograd_latency_conv_matches_oracle(form, monkeypatch):
    """conv_wino_lat44_impl.h (k = 7 / 11 on whole 32-row blocks: F(4,4) tap groups on the quad lattice, the default) and conv_wino_lat_impl.h (k = 3;
    everything under FV_LAT_WINO44=0: F(2,3)) — the Winograd convs of launches too small to fill the chip (the reference's single-utterance forward,
    test.py:88-90): 16-row tiles on 16x16x4 MFMAs, K split over the four waves.  SiLU + bias + residual, the plain conv, and
    c1's fused post-activation against the CPU oracle on ragged shapes; the direct split-K kernels on the same layer via fv_conv_set_algorithm."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import FusedConv
    monkeypatch.setenv("FV_LAT_WINO44", "1" if form == "f44" else "0")
    _lib.reload_env()
    try:
        for (c, k, d, B, T) in LAT_CASES:
            rng = np.random.default_rng(c * 1000 + k * 7 + d + T)
            x = rng.normal(size=(B, c, T)).astype(np.float32)
            w = (rng.normal(size=(c, c, k)) / np.sqrt(c * k)).astype(np.float32)
            b = rng.normal(size=c).astype(np.float32)
            pad = (k - 1) * d // 2
            ref = orc.conv1d(orc.silu(x), w, b, dilation=d, padding=pad)
            res = rng.normal(size=ref.shape).astype(np.float32)
            y = _run(w, b, x, res, dilation=d, padding=pad, pre_act=_lib.FV_ACT_SILU)
            want = "conv_wino_lat44<" if (form == "f44" and k >= 7 and c % 32 == 0) else "conv_wino_lat<"
            assert _lib.last_kernel().startswith(want), (_lib.last_kernel(), c, k, d, B, T)
            _check(y, ref + res)
            y2 = _run(w, None, x, None, dilation=d, padding=pad)
            assert _lib.last_kernel().startswith(want), _lib.last_kernel()
            _check(y2, orc.conv1d(x, w, None, dilation=d, padding=pad))
            y3 = _run(w, b, x, None, dilation=d, padding=pad, pre_act=_lib.FV_ACT_SILU, post_act=_lib.FV_ACT_SILU)
            _check(y3, orc.silu(ref))
            conv = FusedConv(w, b, dilation=d, padding=pad, pre_act=_lib.FV_ACT_SILU).set_algorithm("direct")
            yd = conv(torch.from_numpy(x).to(_dev()), torch.from_numpy(res).to(_dev()))
            torch.cuda.synchronize()
            assert _lib.last_kernel().startswith("conv_mfma<"), _lib.last_kernel()
            assert np.abs(yd.cpu().numpy() - y).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    finally:
        monkeypatch.delenv("FV_LAT_WINO44")
        _lib.reload_env()


@pytest.mark.parametrize("c,k,d,T", [(128, 11, 1, 5504), (256, 7, 3, 688), (64, 11, 5, 3000), (128, 7, 1, 2049)])
def test_winograd_latency_conv_with_heavy_tailed_weights_against_the_float64_sum(c, k, d, T):
    """conv_wino_lat44_impl.h on a layer shaped like a TRAINED one (gains over 100 x, |w| outliers, channels of unequal scale) against the float64
    direct sum, next to the direct-sum kernel of the same layer: the bars of the batch kernels' test above.  Reference: hifigan.py:101-108."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import FusedConv
    rng = np.random.default_rng(c * 31 + k * 7 + d)
    w = _heavy_tailed_weight(rng, c, k)
    b = rng.normal(size=c).astype(np.float32)
    x = (rng.normal(size=(1, c, T)) * np.exp(rng.uniform(np.log(0.3), np.log(3.0), size=(1, c, 1)))).astype(np.float32)
    ref = _conv1d_f64(_silu64(x), w, b, d)
    scale = float(np.abs(ref).max())
    xd = torch.from_numpy(x).to(_dev())
    conv = FusedConv(w, b, dilation=d, padding=(k - 1) * d // 2, pre_act=_lib.FV_ACT_SILU)
    yw = conv(xd).cpu().numpy()
    assert _lib.last_kernel().startswith("conv_wino_lat44<"), _lib.last_kernel()
    yd = conv.set_algorithm("direct")(xd).cpu().numpy()
    assert _lib.last_kernel().startswith("conv_mfma<"), _lib.last_kernel()
    ew, ed = float(np.abs(yw - ref).max()), float(np.abs(yd - ref).max())
    print(f"heavy-tailed latency C={c} k={k} d={d}: scale {scale:.2f}  winograd {ew:.2e}  direct {ed:.2e}  ratio {ew / max(ed, 1e-12):.2f}")
    assert ew <= 2e-5 * max(scale, 1.0), (ew, scale)
    assert ew <= 4.0 * ed + 1e-6 * max(scale, 1.0), (ew, ed)


SPLITK_DIRECT_CASES = [
    # (cin, cout, k, stride / None, B, T): the few-tap convs of a single-clip forward — conv_pre (k = 7), the polyphase upsamplers (two taps per phase,
    # one for k = stride) — at ragged lengths, channel counts that leave a part-filled last chunk, one and two clips
    (80, 512, 7, None, 1, 86), (80, 96, 7, None, 2, 33), (100, 64, 7, None, 1, 5), (512, 256, 16, 8, 1, 86), (256, 128, 16, 8, 1, 200), (72, 40, 16, 8, 2, 31),
    (64, 32, 4, 2, 1, 700), (32, 16, 4, 2, 1, 1500), (48, 24, 2, 2, 1, 257),
]


@pytest.mark.parametrize("cin,cout,k,u,B,T", SPLITK_DIRECT_CASES)
def test_split_k_launches_with_operands_straight_from_global_memory(cin, cout, k, u, B, T, monkeypatch):
    """conv_mfma_splitk_direct_kernel (conv_mfma_impl.h, round 5): the few-tap split-K launches of a single clip without the LDS staging — against the CPU
    oracle, and bit for bit against the LDS-staged kernel (FV_SPLITK_DIRECT=0): the same products in the same order.  Reference: hifigan.py:164-187, 226-231."""
    from vocoder_amd import _lib
    rng = np.random.default_rng(cin * 131 + cout * 7 + k + T)
    x = rng.normal(size=(B, cin, T)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    if u is None:
        w = (rng.normal(size=(cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
        kw = dict(padding=(k - 1) // 2, pre_act=_lib.FV_ACT_NONE)
        ref = orc.conv1d(x, w, b, padding=(k - 1) // 2)
    else:
        w = (rng.normal(size=(cin, cout, k)) / np.sqrt(cin * k / u)).astype(np.float32)
        kw = dict(transposed=True, stride=u, padding=(k - u) // 2, pre_act=_lib.FV_ACT_SILU)
        ref = orc.conv_transpose1d(orc.silu(x), w, b, stride=u, padding=(k - u) // 2)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FV_SPLITK_DIRECT", mode)
        _lib.reload_env()
        try:
            outs[mode] = _run(w, b, x, None, **kw)
            name = _lib.last_kernel()
            assert "splitK" in name, name
            ks = int(name.split(" k=")[1].split()[0])   # taps of the GEMM view (polyphase: k / stride, or k scattered)
            assert name.endswith(" direct") == (mode == "1" and ks in (1, 2, 7)), (name, mode)
            print(mode, name)
        finally:
            monkeypatch.delenv("FV_SPLITK_DIRECT")
            _lib.reload_env()
    _check(outs["1"], ref)
    assert np.array_equal(outs["1"], outs["0"])


def test_winograd_conv_offsets_beyond_2_gib(monkeypatch):
    """One batch item of 2.2 GB (C = 128, T = 4.3 M samples: byte offsets past 2^31, under the 4 GiB addressing span conv_layer_run
    enforces): the Winograd kernel (staging offsets, SGPR row offsets of its epilogue) against the direct-sum kernel on the same
    device tensors — first, middle and last columns of the first and last rows."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import FusedConv
    C, T, k, d = 128, 4_300_000, 11, 5
    g = torch.Generator(device="cuda:0").manual_seed(3)
    x = torch.randn(1, C, T, device=_dev(), generator=g)
    res = torch.randn(1, C, T, device=_dev(), generator=g)
    w = torch.randn(C, C, k, generator=torch.Generator().manual_seed(4)) / (C * k) ** 0.5
    conv = FusedConv(w, torch.zeros(C), dilation=d, padding=(k - 1) * d // 2, pre_act=_lib.FV_ACT_SILU)
    try:
        y1 = conv(x, res)
        assert _lib.last_kernel().startswith("conv_wino44<"), _lib.last_kernel()
        monkeypatch.setenv("FV_WINO", "0")
        _lib.reload_env()
        y0 = conv(x, res)
        assert _lib.last_kernel().startswith("conv_mfma<"), _lib.last_kernel()
        torch.cuda.synchronize()
        for sl in (slice(0, 4096), slice(T // 2 - 2048, T // 2 + 2048), slice(T - 4096, T)):
            for rows in (slice(0, 2), slice(C - 2, C)):
                dmax = float((y1[0, rows, sl] - y0[0, rows, sl]).abs().max())
                assert dmax <= 2e-5, (sl, rows, dmax)
        assert float((y1 - y0).abs().max()) <= 2e-5
    finally:
        monkeypatch.delenv("FV_WINO", raising=False)
        _lib.reload_env()


# ---- heavy-tailed weights: what trained weight-norm layers look like, not what a Gaussian initialiser draws ----
def _heavy_tailed_weight(rng, c, k):
    """A folded weight-norm layer w = g * v / ||v|| (hifigan.py:31) with the per-output-channel gain g log-uniform over 100 x and
    0.4 % of the taps 30 x larger than their neighbours: the transforms of the Winograd kernels mix taps, so an outlier is added to
    and subtracted from small values before the matrix product."""
    v = rng.normal(size=(c, c, k))
    hit = rng.random(size=v.shape) < 0.004
    v = np.where(hit, 30.0 * v, v)
    g = np.exp(rng.uniform(np.log(0.03), np.log(3.0), size=(c, 1, 1)))
    return (g * v / np.sqrt((v * v).sum(axis=(1, 2), keepdims=True))).astype(np.float32)


def _conv1d_f64(x, w, b, d):
    """float64 direct sum of a 'same' dilated Conv1d (hifigan.py:36-57): the arbiter between two fp32 summation orders."""
    B, C, T = x.shape
    k = w.shape[2]
    pad = (k - 1) * d // 2
    xp = np.zeros((B, C, T + 2 * pad))
    xp[:, :, pad:pad + T] = x
    y = np.zeros((B, w.shape[0], T))
    for j in range(k):
        y += np.einsum("oi,bit->bot", w[:, :, j].astype(np.float64), xp[:, :, j * d:j * d + T])
    return y + b.astype(np.float64)[None, :, None]


def _silu64(x):
    x = x.astype(np.float64)
    return x / (1.0 + np.exp(-x))


HEAVY_CASES = [
    # (C, k, dil, B, T, kernel family the Winograd setting must dispatch)
    (128, 11, 1, 2, 517, "conv_wino44<"), (128, 7, 3, 1, 700, "conv_wino44<"), (64, 11, 5, 2, 900, "conv_wino44<"), (256, 7, 1, 1, 260, "conv_wino44<"),
    (256, 3, 1, 2, 200, "conv_wino<"), (32, 11, 3, 2, 1500, "conv_wino<"),
]


@pytest.mark.parametrize("c,k,d,B,T,family", HEAVY_CASES)
def test_winograd_convs_with_heavy_tailed_weights_against_the_float64_sum(c, k, d, B, T, family, monkeypatch):
    """The F(4,4) / F(2,3) tap-group kernels (conv_wino44_impl.h, conv_wino_impl.h) on a layer shaped like a TRAINED one — gains spread over
    100 x, |w| outliers, input channels of unequal scale — against the float64 direct sum, next to the direct-sum kernel of the same layer:
    bar 2e-5 of the output's scale (the model bar is 1e-4 absolute on a waveform in (-1, 1)), and the Winograd error within 4 x the direct
    kernel's.  Reference: hifigan.py:101-108."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import FusedConv
    rng = np.random.default_rng(c * 31 + k * 7 + d)
    w = _heavy_tailed_weight(rng, c, k)
    b = rng.normal(size=c).astype(np.float32)
    x = (rng.normal(size=(B, c, T)) * np.exp(rng.uniform(np.log(0.3), np.log(3.0), size=(1, c, 1)))).astype(np.float32)
    ref = _conv1d_f64(_silu64(x), w, b, d)
    scale = float(np.abs(ref).max())
    xd = torch.from_numpy(x).to(_dev())
    monkeypatch.setenv("FV_WINO", "2")
    _lib.reload_env()
    try:
        conv = FusedConv(w, b, dilation=d, padding=(k - 1) * d // 2, pre_act=_lib.FV_ACT_SILU)
        yw = conv.set_algorithm("winograd")(xd).cpu().numpy()
        assert _lib.last_kernel().startswith(family), _lib.last_kernel()
        yd = conv.set_algorithm("direct")(xd).cpu().numpy()
        assert _lib.last_kernel().startswith("conv_mfma<"), _lib.last_kernel()
    finally:
        monkeypatch.delenv("FV_WINO")
        _lib.reload_env()
    ew, ed = float(np.abs(yw - ref).max()), float(np.abs(yd - ref).max())
    print(f"heavy-tailed C={c} k={k} d={d}: scale {scale:.2f}  winograd {ew:.2e}  direct {ed:.2e}  ratio {ew / max(ed, 1e-12):.2f}")
    assert ew <= 2e-5 * max(scale, 1.0), (ew, scale)
    assert ew <= 4.0 * ed + 1e-6 * max(scale, 1.0), (ew, ed)


@pytest.mark.parametrize("C,k,d", [(16, 3, 1), (16, 11, 5), (32, 7, 3), (32, 11, 1), (64, 3, 3)])
def test_winograd_pairs_with_heavy_tailed_weights_against_the_float64_sum(C, k, d):
    """The fused (c1, c2) Winograd pairs (pair_wino_impl.h) on heavy-tailed layers against x + c2(silu(c1(silu(x)))) in float64, next to the
    direct-sum pair kernel.  Reference: hifigan.py:102-107."""
    from vocoder_amd.engine import FusedConv
    rng = np.random.default_rng(C * 17 + k * 5 + d)
    w1, w2 = _heavy_tailed_weight(rng, C, k), _heavy_tailed_weight(rng, C, k)
    b1, b2 = rng.normal(size=C).astype(np.float32), rng.normal(size=C).astype(np.float32)
    B, T = 2, 2100
    x = (rng.normal(size=(B, C, T)) * np.exp(rng.uniform(np.log(0.3), np.log(3.0), size=(1, C, 1)))).astype(np.float32)
    nb = min(B, 2)
    xt = _conv1d_f64(_silu64(x[:nb]), w1, b1, d)
    ref = x[:nb].astype(np.float64) + _conv1d_f64(_silu64(xt), w2, b2, 1)
    scale = float(np.abs(ref).max())
    c1 = FusedConv(w1, b1, dilation=d, padding=(k * d - d) // 2)
    c2 = FusedConv(w2, b2, padding=(k - 1) // 2)
    xd = torch.from_numpy(x).to(_dev())
    yw = c1.set_algorithm("auto").pair(c2, xd)[:nb].cpu().numpy()
    assert _last_kernel().startswith(("pair_wino<", "pair_wino44<")), _last_kernel()
    yd = c1.set_algorithm("direct").pair(c2, xd)[:nb].cpu().numpy()
    assert _last_kernel().startswith("resblock_pair<"), _last_kernel()
    ew, ed = float(np.abs(yw - ref).max()), float(np.abs(yd - ref).max())
    print(f"heavy-tailed pair C={C} k={k} d={d}: scale {scale:.2f}  winograd {ew:.2e}  direct {ed:.2e}  ratio {ew / max(ed, 1e-12):.2f}")
    assert ew <= 2e-5 * max(scale, 1.0), (ew, scale)
    assert ew <= 4.0 * ed + 1e-6 * max(scale, 1.0), (ew, ed)


@pytest.mark.parametrize("cin,cout,B,T", [(512, 256, 2, 86), (512, 256, 32, 86), (256, 128, 1, 100), (256, 128, 3, 517), (64, 32, 1, 7), (512, 256, 1, 3)])
def test_stride8_upsampler_quad_stores_equal_the_single_stores(cin, cout, B, T, monkeypatch):
    """the 16-byte-store epilogue of the stride-8 polyphase upsamplers (conv_mfma_impl.h conv_epilogue, p.vec_store) against the general
    epilogue (FV_VEC_STORE=0) BIT for bit — ragged t_in, quads cropped by the transposed conv's padding at t < 0 and t >= Tout, the 128 x 96 / 128 x 128 /
    narrow / small-launch tiles — with y 16-byte aligned and offset by 4 bytes (which must fall back to the general epilogue); both against the oracle.
    Reference: ups[i] of hifigan.py:175-187,231 (k = 16, stride 8, padding 4)."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import FusedConv
    k, u = 16, 8
    rng = np.random.default_rng(cin + B * 7 + T)
    x = rng.normal(size=(B, cin, T)).astype(np.float32)
    w = (rng.normal(size=(cin, cout, k)) / np.sqrt(cin * 2)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    ref = orc.conv_transpose1d(orc.silu(x), w, b, stride=u, padding=(k - u) // 2)
    xd = torch.from_numpy(x).to(_dev())
    outs, kernels = {}, {}
    try:
        for vec in ("1", "0"):
            monkeypatch.setenv("FV_VEC_STORE", vec)
            _lib.reload_env()
            conv = FusedConv(w, b, transposed=True, stride=u, padding=(k - u) // 2, pre_act=_lib.FV_ACT_SILU)
            n = int(np.prod(ref.shape))
            for off in (0, 1):
                buf = torch.full((n + 8,), float("nan"), device=_dev())
                assert buf.data_ptr() % 16 == 0
                y = conv(xd, out=buf[off:off + n].view(*ref.shape))
                torch.cuda.synchronize()
                outs[(vec, off)] = y.cpu().numpy().copy()
                kernels[(vec, off)] = _lib.last_kernel()
                assert bool(torch.isnan(buf[:off]).all()) and bool(torch.isnan(buf[off + n:]).all())   # nothing outside the output
    finally:
        monkeypatch.delenv("FV_VEC_STORE", raising=False)
        _lib.reload_env()
    assert len(set(kernels.values())) == 1, kernels
    _check(outs[("1", 0)], ref)
    for key, val in outs.items():
        assert np.array_equal(val, outs[("0", 0)]), (key, kernels[key], float(np.abs(val - outs[("0", 0)]).max()))
