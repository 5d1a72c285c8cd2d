"""GPU: the opt-in f16x3 precision mode (split-fp16 MFMA, vocoder_amd/csrc/conv_f16x3_impl.h) against the CPU oracle.
Same tolerance as the fp32 path: |d| <= 1e-4 absolute (north_star); the split keeps ~22 mantissa bits per product, so the
observed error must also stay in the fp32-roundoff class (<= 2e-5 of the output scale)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev():
    assert torch.cuda.is_available(), "GPU test on a box without a GPU"
    return torch.device("cuda:0")


def _conv(w, b, x, res, precision, **kw):
    from vocoder_amd import _lib
    from vocoder_amd.engine import FusedConv
    conv = FusedConv(w, b, **kw).set_precision(precision)
    xt = torch.from_numpy(x).to(_dev())
    rt = None if res is None else torch.from_numpy(res).to(_dev())
    y = conv(xt, rt)
    torch.cuda.synchronize()
    return y.cpu().numpy(), _lib.last_kernel()


CASES = [
    # (Cin, Cout, k, dil, B, T)
    (128, 128, 11, 1, 2, 517), (128, 128, 7, 3, 1, 300), (128, 128, 3, 5, 2, 1000),
    (256, 256, 11, 5, 1, 97), (256, 256, 3, 1, 2, 688), (256, 256, 7, 1, 1, 130),
    (64, 64, 11, 3, 2, 700), (64, 64, 7, 5, 1, 1025), (64, 64, 3, 1, 3, 259),
    (48, 96, 7, 1, 2, 200),      # C_in not a multiple of 16 chunks? 48 = 3 chunks; M = 96 -> partial m-block
    (40, 64, 3, 3, 1, 77),       # C_in = 40 -> zero-padded half chunk
    (32, 32, 11, 1, 2, 1500), (32, 32, 7, 5, 1, 255), (32, 32, 3, 3, 3, 513), (64, 32, 11, 3, 1, 300), (32, 48, 7, 1, 2, 90),   # 32 x 256 tiles
]


@pytest.mark.parametrize("cin,cout,k,d,B,T", CASES)
def test_f16x3_conv_matches_oracle(cin, cout, k, d, B, T):
    from vocoder_amd import _lib
    rng = np.random.default_rng(cin * 1000 + cout + k * 7 + d)
    x = (rng.normal(size=(B, cin, T)) * 3.0).astype(np.float32)
    w = (rng.normal(size=(cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    pad = (k * d - d) // 2
    ref = orc.conv1d(orc.silu(x), w, b, dilation=d, padding=pad)
    res = rng.normal(size=ref.shape).astype(np.float32)
    y, kern = _conv(w, b, x, res, "f16x3", dilation=d, padding=pad, pre_act=_lib.FV_ACT_SILU)
    assert kern.startswith("conv_f16x3"), kern
    ref = ref + res
    err = np.abs(y - ref).max()
    scale = max(np.abs(ref).max(), 1.0)
    assert err <= 1e-4 and err <= 2e-5 * scale, f"max|d|={err:.3e} scale={scale:.2f} ({kern})"
    # and it must be at least as close to the oracle as 4x the exact-fp32 kernel's own error (fp32-class accuracy)
    y32, kern32 = _conv(w, b, x, res, "f32", dilation=d, padding=pad, pre_act=_lib.FV_ACT_SILU)
    assert kern32.startswith(("conv_mfma", "conv_wino")), kern32   # (direct sums, or Winograd tap groups where the shape has them)
    err32 = np.abs(y32 - ref).max()
    assert err <= max(4 * err32, 2e-6 * scale), f"f16x3 {err:.3e} vs f32 {err32:.3e}"


def test_f16x3_wide_dynamic_range():
    """Activations spanning 1e-6 .. 1e3 and weights spanning 2^-20 of their maximum: the split must degrade gracefully
    (absolute error bounded by the fp32-roundoff of the large terms)."""
    from vocoder_amd import _lib
    rng = np.random.default_rng(11)
    cin = cout = 64
    k, d, T = 7, 1, 400
    x = (rng.normal(size=(1, cin, T)) * np.exp(rng.uniform(np.log(1e-6), np.log(1e3), size=(1, cin, T)))).astype(np.float32)
    w = (rng.normal(size=(cout, cin, k)) * np.exp2(rng.uniform(-20, 0, size=(cout, cin, k)))).astype(np.float32)
    pad = (k - 1) // 2
    y, kern = _conv(w, None, x, None, "f16x3", dilation=d, padding=pad)
    assert kern.startswith("conv_f16x3"), kern
    ref64 = orc.conv1d(x.astype(np.float64).astype(np.float32), w, None, dilation=d, padding=pad)
    mag = orc.conv1d(np.abs(x), np.abs(w), None, dilation=d, padding=pad)   # sum |x w|
    rel = np.abs(y - ref64) / np.maximum(mag, 1e-30)
    assert rel.max() < 2e-6, rel.max()


def test_f16x3_falls_back_to_f32_kernels_where_not_covered():
    from vocoder_amd import _lib
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1, 16, 100)).astype(np.float32)
    w = rng.normal(size=(16, 16, 3)).astype(np.float32) * 0.1
    y, kern = _conv(w, None, x, None, "f16x3", padding=1)
    assert kern.startswith("conv_mfma"), kern
    np.testing.assert_allclose(y, orc.conv1d(x, w, None, padding=1), atol=1e-5)


def test_f16x3_hifigan_model_matches_oracle():
    """Whole generator in f16x3 mode vs the oracle (tolerance 1e-4 on the waveform, north_star)."""
    from vocoder_amd import synthetic
    from vocoder_amd.modules.generators.hifigan import HiFiGANGenerator
    cfg = dict(synthetic.HIFIGAN_V1_44K)
    sd = synthetic.hifigan_state_dict(cfg, seed=0)
    mel = synthetic.synthetic_mel(1, cfg["num_mels"], 12, seed=1)
    ref = orc.hifigan_forward(sd, cfg, mel)
    gen = HiFiGANGenerator(**cfg).eval()
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    gen.precision = "f16x3"
    y = gen.to(_dev())(torch.from_numpy(mel).to(_dev())).cpu().numpy()
    assert np.abs(y - ref).max() <= 1e-4, np.abs(y - ref).max()
    prof = gen.engine(_dev()).profile(torch.from_numpy(mel).to(_dev()))
    assert any(r["kernel"].startswith("conv_f16x3") for r in prof), [r["kernel"] for r in prof][:5]


def test_f16x3_full_size_engine_is_deterministic_and_close_to_f32():
    """BASELINE-size clips (T_mel = 86, 8 clips): the f16x3 engine must give bit-identical waveforms run after run (a
    compiler-scheduled VALU -> SDWA hazard once made the activation split timing-dependent) and stay within 1e-4 of the
    exact-fp32 engine — also through the no-activation (c2) and accumulate paths that only the engine exercises."""
    from vocoder_amd import _lib, synthetic
    from vocoder_amd.engine import Engine, upsampler_config
    cfg = dict(synthetic.HIFIGAN_V1_44K)
    sd = synthetic.hifigan_state_dict(cfg, seed=0)
    mel = torch.from_numpy(synthetic.synthetic_mel(8, cfg["num_mels"], 86, seed=5)).to(_dev())
    e32 = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd)
    ref = e32(mel).clone()
    e16 = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd, precision="f16x3")
    y0 = e16(mel).clone()
    torch.cuda.synchronize()
    assert float((y0 - ref).abs().max()) <= 1e-4
    for _ in range(12):
        y = e16(mel)
        torch.cuda.synchronize()
        assert torch.equal(y, y0)


POINTWISE = [(128, 512, 2, 94), (512, 128, 3, 94), (512, 2048, 1, 50), (2048, 512, 2, 33), (256, 512, 4, 94), (64, 128, 2, 700)]


@pytest.mark.parametrize("cin,cout,B,T", POINTWISE)
def test_f16x3_pointwise_conv_matches_oracle(cin, cout, B, T):
    """ConvNeXt pointwise GEMMs (k = 1): flattened (batch, time) columns, GELU epilogue, residual."""
    from vocoder_amd import _lib
    rng = np.random.default_rng(cin + cout + B)
    x = (rng.normal(size=(B, cin, T)) * 2.0).astype(np.float32)
    w = (rng.normal(size=(cout, cin, 1)) / np.sqrt(cin)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    ref = orc.gelu(orc.conv1d(x, w, b))
    y, kern = _conv(w, b, x, None, "f16x3", post_act=_lib.FV_ACT_GELU)
    assert kern.startswith("conv_f16x3<k=1"), kern
    scale = max(np.abs(ref).max(), 1.0)
    err = np.abs(y - ref).max()
    assert err <= 1e-4 and err <= 2e-5 * scale, f"max|d|={err:.3e} ({kern})"
    res = rng.normal(size=ref.shape).astype(np.float32)
    ref2 = orc.conv1d(x, w, b) + res
    y2, _ = _conv(w, b, x, res, "f16x3")
    assert np.abs(y2 - ref2).max() <= 2e-5 * max(np.abs(ref2).max(), 1.0)


def test_f16x3_vocos_and_convnext_match_oracle():
    """Whole ConvNeXt / Vocos forwards in f16x3 mode (pointwise GEMMs with layer scale + in-place residual)."""
    from vocoder_amd import _lib, synthetic as syn
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config
    cfg = dict(backbone=dict(input_channels=80, depths=[1, 1, 2, 1], dims=[128, 256, 512, 1024], kernel_size=7),
               head=dict(dim=1024, n_fft=1024, hop_length=256, win_length=1024, padding="same"))
    sd = syn.vocos_state_dict(cfg, seed=3)
    mel = syn.synthetic_mel(3, 80, 10, seed=9)
    ref = orc.vocos_forward(sd, cfg, mel)
    eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]),
                 state_dict=sd, precision="f16x3")
    x = torch.from_numpy(mel).to(_dev())
    y = eng(x).cpu().numpy()
    assert np.abs(y - ref).max() <= 1e-4 * float(np.abs(ref).max()), (np.abs(y - ref).max(), np.abs(ref).max())   # of the waveform's own peak
    prof = eng.profile(x)
    assert any(r["kernel"].startswith("conv_f16x3<k=1") for r in prof), [r["kernel"] for r in prof][:8]


def test_f16x3_bigvgan_and_refinegan_goldens():
    """The other generator families in f16x3 mode against the reference captures: BigVGAN (split kernels for the AMPBlock convs,
    which see no pre-activation) and RefineGAN (leaky-ReLU pre-activations are not implemented by the split kernels, so those
    layers must quietly stay on the fp32 kernels)."""
    from conftest import load_golden
    from vocoder_amd import _lib, synthetic as syn
    from vocoder_amd.engine import Engine, refinegan_config, upsampler_config
    dev = _dev()
    g = load_golden("bigvgan_tiny.npz")
    sd = syn.bigvgan_state_dict(g["cfg"], g["seed"])
    eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**g["cfg"]), state_dict=sd, precision="f16x3")
    y = eng(torch.from_numpy(g["mel"]).to(dev)).cpu().numpy()
    assert np.abs(y - g["out"]).max() <= 1e-4, np.abs(y - g["out"]).max()
    g = load_golden("refinegan_tiny.npz")
    cfg = g["cfg"]
    sd = syn.refinegan_state_dict(cfg, g["seed"])
    B, _, T = g["mel"].shape
    noise = np.concatenate([n.reshape(-1) for n in syn.refinegan_noise(cfg, B, T, seed=int(g["noise_seed"]))])
    eng = Engine(_lib.FV_MODEL_REFINEGAN, refine=refinegan_config(**cfg), state_dict=sd, precision="f16x3")
    y = eng(torch.from_numpy(g["mel"]).to(dev), None, torch.from_numpy(g["template"]).to(dev),
            torch.from_numpy(noise).to(dev)).cpu().numpy()
    assert np.abs(y - g["out"]).max() <= 1e-4, np.abs(y - g["out"]).max()


def test_f16x3_range_contract_is_loud():
    """|activation| >= 65504 is outside the f16x3 contract (include/fishvoc.h fv_precision): the fp16 high plane becomes inf and
    the affected outputs non-finite — never a silently wrong finite number; the fp32 mode handles the same input."""
    rng = np.random.default_rng(2)
    x = rng.normal(size=(1, 64, 300)).astype(np.float32)
    x[0, 5, 100] = 1.0e5
    w = (rng.normal(size=(64, 64, 3)) / 14.0).astype(np.float32)
    y16, kern = _conv(w, None, x, None, "f16x3", padding=1)
    assert kern.startswith("conv_f16x3"), kern
    assert not np.isfinite(y16[0, :, 99:102]).all()
    keep = np.ones(300, bool)
    keep[99:102] = False
    ref = orc.conv1d(x, w, None, padding=1)
    assert np.isfinite(y16[0][:, keep]).all() and np.abs(y16[0][:, keep] - ref[0][:, keep]).max() <= 1e-4
    y32, _ = _conv(w, None, x, None, "f32", padding=1)
    # 1e5-scale values: fp32 roundoff relative to the largest output (the Winograd tap groups round the outlier's neighbours at its ulp)
    assert np.isfinite(y32).all() and np.abs(y32 - ref).max() <= 3e-6 * np.abs(ref).max(), (np.abs(y32 - ref).max(), np.abs(ref).max())


PAIR_CASES = [(256, 11, 5, 1, 688), (256, 3, 1, 2, 100), (256, 7, 3, 1, 87), (128, 11, 1, 2, 300), (128, 11, 5, 1, 517), (128, 7, 3, 2, 200), (128, 3, 1, 1, 1000), (128, 3, 5, 3, 97),
              (64, 11, 3, 2, 700), (64, 7, 1, 1, 255), (64, 3, 3, 2, 129), (64, 11, 5, 1, 40), (128, 7, 5, 1, 5),
              (32, 11, 5, 2, 1000), (32, 7, 3, 1, 374), (32, 3, 1, 3, 383), (32, 11, 1, 1, 9), (32, 7, 5, 2, 2049),
              # C = 16: two output samples per MFMA row (pair16_f16x3.hip); every (k, d), tile edges, tiny clips
              (16, 11, 1, 2, 2000), (16, 11, 3, 1, 1504), (16, 11, 5, 2, 500), (16, 7, 1, 1, 506), (16, 7, 3, 2, 1018),
              (16, 7, 5, 1, 2), (16, 3, 1, 3, 510), (16, 3, 3, 1, 4096), (16, 3, 5, 2, 38)]


@pytest.mark.parametrize("C,k,d,B,T", PAIR_CASES)
def test_f16x3_wide_fused_pair_matches_oracle(C, k, d, B, T):
    """y = x + c2(silu(c1(silu(x)))) in one launch on the split-fp16 path (fv_conv_pair_forward, pair_f16x3_impl.h):
    ragged tile edges, clips shorter than a tile, every (k, dilation) the ResBlocks use."""
    if os.environ.get("FV_NO_F16X3_PAIRS"):
        pytest.skip("the diagnostic switch FV_NO_F16X3_PAIRS turns these kernels off")
    from vocoder_amd import _lib
    from vocoder_amd.engine import FusedConv
    rng = np.random.default_rng(C + 13 * k + d)
    x = (rng.normal(size=(B, C, T)) * 2.0).astype(np.float32)
    w1 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k) * 1.3).astype(np.float32)
    w2 = (rng.normal(size=(C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    b1 = rng.normal(size=C).astype(np.float32) * 0.1
    b2 = rng.normal(size=C).astype(np.float32) * 0.1
    h = orc.conv1d(orc.silu(x), w1, b1, dilation=d, padding=(k * d - d) // 2)
    ref = x + orc.conv1d(orc.silu(h), w2, b2, dilation=1, padding=(k - 1) // 2)
    c1 = FusedConv(w1, b1, dilation=d, padding=(k * d - d) // 2).set_precision("f16x3")
    c2 = FusedConv(w2, b2, padding=(k - 1) // 2).set_precision("f16x3")
    y = c1.pair(c2, torch.from_numpy(x).to(_dev()))
    torch.cuda.synchronize()
    assert _lib.last_kernel().startswith("pair_f16x3"), _lib.last_kernel()
    err = np.abs(y.cpu().numpy() - ref).max()
    scale = max(np.abs(ref).max(), 1.0)
    assert err <= 1e-4 and err <= 2e-5 * scale, f"max|d|={err:.3e} scale={scale:.2f}"


def test_f16x3_pair16_needs_an_even_length_and_falls_back_to_fp32_otherwise():
    if os.environ.get("FV_NO_F16X3_PAIRS"):
        pytest.skip("the diagnostic switch FV_NO_F16X3_PAIRS turns this kernel off")
    from vocoder_amd import _lib
    from vocoder_amd.engine import FusedConv
    rng = np.random.default_rng(5)
    w = (rng.normal(size=(16, 16, 7)) / 10.0).astype(np.float32)
    c1 = FusedConv(w, None, dilation=3, padding=9).set_precision("f16x3")
    c2 = FusedConv(w, None, padding=3).set_precision("f16x3")
    for T, want in ((300, "pair_f16x3"), (301, "pair_wino")):   # (the exact-fp32 pair kernel of this shape: pair_wino_impl.h)
        x = rng.normal(size=(2, 16, T)).astype(np.float32)
        y = c1.pair(c2, torch.from_numpy(x).to(_dev()))
        torch.cuda.synchronize()
        assert _lib.last_kernel().startswith(want), (T, _lib.last_kernel())
        h = orc.conv1d(orc.silu(x), w, None, dilation=3, padding=9)
        ref = x + orc.conv1d(orc.silu(h), w, None, dilation=1, padding=3)
        assert np.abs(y.cpu().numpy() - ref).max() <= 1e-5


CONVT = [(512, 256, 16, 8, 2, 86), (256, 128, 16, 8, 1, 300), (128, 64, 8, 2, 2, 500), (64, 64, 4, 2, 1, 333), (128, 64, 16, 8, 1, 1)]


@pytest.mark.parametrize("cin,cout,k,u,B,T", CONVT)
def test_f16x3_conv_transpose_matches_oracle(cin, cout, k, u, B, T):
    """The up-sampling ConvTranspose1d (polyphase form: rows = (c_out, phase), scatter store) on the split-fp16 path."""
    from vocoder_amd import _lib
    rng = np.random.default_rng(cin + cout + k + u)
    x = (rng.normal(size=(B, cin, T)) * 2.0).astype(np.float32)
    w = (rng.normal(size=(cin, cout, k)) / np.sqrt(cin * k / u)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    ref = orc.conv_transpose1d(orc.silu(x), w, b, stride=u, padding=(k - u) // 2)
    y, kern = _conv(w, b, x, None, "f16x3", transposed=True, stride=u, padding=(k - u) // 2, pre_act=_lib.FV_ACT_SILU)
    assert kern.startswith("conv_f16x3"), kern
    assert y.shape == ref.shape
    err = np.abs(y - ref).max()
    assert err <= 1e-4 and err <= 2e-5 * max(np.abs(ref).max(), 1.0), f"max|d|={err:.3e} ({kern})"
