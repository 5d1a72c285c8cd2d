"""GPU: whole-generator parity through fv_forward against (a) the golden fixtures captured from the reference and
(b) the CPU oracle on seeded inputs.  Bar: |d| <= 1e-4 on the waveform (BASELINE.json north_star)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc
from vocoder_amd import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 1e-4


def _peak_tol(ref):
    """Vocos / ISTFT-head waveforms are small with the synthetic head weights (peaks of 0.02 ... 0.07 on the reference captures),
    so an absolute 1e-4 would let an error of 0.5 % of the signal pass: their bar is 1e-4 of the expected waveform's own peak
    (2e-6 absolute at a peak of 0.02).  Reference: vocos.py:57-69."""
    return TOL * float(np.abs(ref).max())


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _hifigan_engine(cfg, sd, kind=None):
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, upsampler_config
    return Engine(kind or _lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd)


def _fwd(eng, x):
    y = eng(torch.from_numpy(x).to(_dev()))
    torch.cuda.synchronize()
    return y.cpu().numpy()


@pytest.mark.parametrize("name", ["hifigan_tiny.npz", "hifigan_narrow.npz", "hifigan_tiny_t1.npz", "hifigan_v1_t12.npz",
                                  "hifigan_v1_t86.npz"])
def test_hifigan_golden(name):
    g = load_golden(name)
    sd = syn.hifigan_state_dict(g["cfg"], g["seed"])
    y = _fwd(_hifigan_engine(g["cfg"], sd), g["mel"])
    assert y.shape == g["out"].shape
    err = np.abs(y - g["out"]).max()
    assert err <= TOL, f"waveform max|d| = {err:.3e} vs reference golden"


def test_hifigan_one_second_reference_clip_alone_and_inside_a_full_batch():
    """BASELINE config[0] / [1]: the reference's own output for a full V1 generator on a 1 s clip (T_mel = 86, hifigan.py:226-249)
    against the HIP path run alone (latency kernels) and with the same clip placed at three positions of a B = 32 batch of other
    clips (throughput kernels, flattened / batched tiles)."""
    g = load_golden("hifigan_v1_t86.npz")
    sd = syn.hifigan_state_dict(g["cfg"], g["seed"])
    eng = _hifigan_engine(g["cfg"], sd)
    assert g["mel"].shape == (1, 80, 86) and g["out"].shape == (1, 1, 44032)
    y1 = _fwd(eng, g["mel"])
    assert np.abs(y1 - g["out"]).max() <= TOL
    mel = syn.synthetic_mel(32, 80, 86, seed=99)
    for pos in (0, 13, 31):
        mel[pos] = g["mel"][0]
    y = _fwd(eng, mel)
    for pos in (0, 13, 31):
        err = np.abs(y[pos] - g["out"][0]).max()
        assert err <= TOL, f"clip at batch position {pos}: max|d| = {err:.3e} vs reference golden"
    # every other item against the oracle at full length would take minutes; two of them do
    for i in (7, 30):
        ref = orc.hifigan_forward(sd, g["cfg"], mel[i:i + 1])
        assert np.abs(y[i] - ref[0]).max() <= TOL


def test_hifigan_v1_vs_oracle_seeded_batch():
    cfg = dict(syn.HIFIGAN_V1_44K)
    sd = syn.hifigan_state_dict(cfg, seed=11)
    mel = syn.synthetic_mel(2, 80, 9, seed=77)     # ragged-ish small T, B=2
    ref = orc.hifigan_forward(sd, cfg, mel)
    y = _fwd(_hifigan_engine(cfg, sd), mel)
    err = np.abs(y - ref).max()
    assert err <= TOL, f"max|d| = {err:.3e}"


def test_hifigan_full_size_properties():
    """BASELINE config[1] size (B=32, T_mel=86): checks that need no oracle at that size —
    batch independence (item i of the batch == the same clip run alone), determinism, range of tanh."""
    cfg = dict(syn.HIFIGAN_V1_44K)
    sd = syn.hifigan_state_dict(cfg, seed=0)
    eng = _hifigan_engine(cfg, sd)
    mel = syn.synthetic_mel(32, 80, 86, seed=1234)
    y = _fwd(eng, mel)
    assert y.shape == (32, 1, 86 * 512)
    assert np.isfinite(y).all() and np.abs(y).max() <= 1.0
    y_again = _fwd(eng, mel)
    assert np.array_equal(y, y_again)
    for i in (0, 17, 31):
        yi = _fwd(eng, mel[i:i + 1])
        # a single clip dispatches the latency (split-K) kernels: same arithmetic, different summation order
        assert np.abs(yi[0] - y[i]).max() <= 2e-5
    # first clip against the oracle on a prefix: the receptive field is finite (one-sided reach of the whole stack is
    # ~9 mel frames, dominated by the k=11, d=5 ResBlocks of stage 0), so a T=28 prefix run agrees with the full run
    # on the first 8 frames.
    ref = orc.hifigan_forward(sd, cfg, mel[:1, :, :28])
    assert np.abs(ref[0, 0, :8 * 512] - y[0, 0, :8 * 512]).max() <= TOL


@pytest.mark.parametrize("tag,spec,factory", [
    ("leaky_relu", ("leaky_relu", 0.1), lambda nn, partial: partial(nn.LeakyReLU, 0.1)), ("relu", "relu", lambda nn, partial: nn.ReLU),
    ("gelu", "gelu", lambda nn, partial: nn.GELU), ("tanh", "tanh", lambda nn, partial: nn.Tanh),
    ("identity", "identity", lambda nn, partial: nn.Identity)])
def test_hifigan_post_activation_goldens_through_the_module(tag, spec, factory):
    """`post_activation` of the reference ctor (hifigan.py:150,213,245; VERDICT r4 missing 1): the drop-in module built with the same factory
    the reference was captured with — nn.LeakyReLU(0.1) is what classic HiFi-GAN checkpoints need — against the reference capture, the
    oracle on a longer ragged clip, and a module the engine has no kernel form for."""
    from functools import partial
    from torch import nn
    from vocoder_amd.modules.generators import HiFiGANGenerator
    g = load_golden("hifigan_post_activation.npz")
    sd = syn.hifigan_state_dict(g["cfg"], g["seed"])
    gen = HiFiGANGenerator(**g["cfg"], post_activation=factory(nn, partial)).eval()
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    gen = gen.to(_dev())
    y = gen(torch.from_numpy(g["mel"]).to(_dev())).cpu().numpy()
    err = np.abs(y - g[f"out_{tag}"]).max()
    assert err <= TOL, f"post_activation={tag}: max|d| = {err:.3e} vs reference golden"
    mel = syn.synthetic_mel(3, g["cfg"]["num_mels"], 171, seed=8)
    ref = orc.hifigan_forward(sd, dict(g["cfg"], post_activation=spec), mel)
    assert np.abs(gen(torch.from_numpy(mel).to(_dev())).cpu().numpy() - ref).max() <= TOL
    with pytest.raises(NotImplementedError, match="post_activation"):
        HiFiGANGenerator(**g["cfg"], post_activation=nn.Softplus)


def test_zero_initialised_post_activation_means_the_reference_default():
    """ADVICE r5 (low), ABI 5: `fv_upsampler_config.post_activation == 0` — what a C caller's zero-initialised struct carries — is the reference
    default nn.SiLU (hifigan.py:150), not Identity; Identity is FV_POST_ACT_IDENTITY (-1).  Three engines on the same weights."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, upsampler_config
    g = load_golden("hifigan_post_activation.npz")
    sd = syn.hifigan_state_dict(g["cfg"], g["seed"])
    mel = syn.synthetic_mel(2, g["cfg"]["num_mels"], 33, seed=4)
    x = torch.from_numpy(mel).to(_dev())
    ys = {}
    for tag, act in (("zero", _lib.FV_POST_ACT_DEFAULT), ("silu", _lib.FV_ACT_SILU), ("identity", _lib.FV_POST_ACT_IDENTITY)):
        ys[tag] = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**g["cfg"], post_activation=act), state_dict=sd)(x).cpu().numpy()
    assert _lib.FV_POST_ACT_DEFAULT == 0 and np.array_equal(ys["zero"], ys["silu"])
    assert np.abs(ys["zero"] - orc.hifigan_forward(sd, g["cfg"], mel)).max() <= TOL
    assert np.abs(ys["identity"] - orc.hifigan_forward(sd, dict(g["cfg"], post_activation="identity"), mel)).max() <= TOL
    assert np.abs(ys["identity"] - ys["zero"]).max() > 1e-3


def test_istft_head_center_padding_golden_and_oracle():
    """ISTFTHead(padding="center") (vocos.py:19-38; VERDICT r4 missing 2): torch.istft(center=True)'s own output, (T - 1) * hop samples, through the
    head module; Vocos with a centre-padded head at a real resolution against the oracle; a single frame raises (no samples, as torch)."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config
    from vocoder_amd.modules.generators.vocos import ISTFTHead
    g = load_golden("istft_head_center.npz")
    cfg = g["cfg"]
    sd = syn.istft_head_state_dict(cfg, g["seed"])
    head = ISTFTHead(**cfg).eval()
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    y = head.to(_dev())(torch.from_numpy(g["x"]).to(_dev())).cpu().numpy()
    assert y.shape == g["wave"].shape
    assert np.abs(y - g["wave"]).max() <= _peak_tol(g["wave"]), np.abs(y - g["wave"]).max()
    with pytest.raises((ValueError, RuntimeError)):
        head(torch.from_numpy(g["x"][:, :, :1].copy()).to(_dev()))
    vc = dict(backbone=dict(input_channels=80, depths=[1, 2], dims=[96, 192], drop_path_rate=0.0, kernel_size=7),
              head=dict(dim=192, n_fft=1024, hop_length=256, win_length=1024, padding="center"))
    vsd = syn.vocos_state_dict(vc, seed=4)
    mel = syn.synthetic_mel(3, 80, 37, seed=12)
    ref = orc.vocos_forward(vsd, vc, mel)
    eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**vc["backbone"]), head=istft_head_config(**vc["head"]), state_dict=vsd)
    yv = _fwd(eng, mel)
    assert yv.shape == ref.shape == (3, 1, 36 * 256)
    assert np.abs(yv - ref).max() <= _peak_tol(ref), (np.abs(yv - ref).max(), np.abs(ref).max())


def test_strict_loading_errors():
    from vocoder_amd.engine import FishVocError
    g = load_golden("hifigan_tiny.npz")
    sd = syn.hifigan_state_dict(g["cfg"], g["seed"])
    bad = dict(sd)
    bad.pop("ups.0.bias")
    with pytest.raises(FishVocError, match="missing weight 'ups.0.bias'"):
        _hifigan_engine(g["cfg"], bad)
    extra = dict(sd)
    extra["bogus.weight"] = np.zeros(3, np.float32)
    with pytest.raises(FishVocError, match="unexpected key"):
        _hifigan_engine(g["cfg"], extra)
    wrong = dict(sd)
    wrong["conv_pre.bias"] = np.zeros(5, np.float32)
    with pytest.raises(FishVocError, match="shape"):
        _hifigan_engine(g["cfg"], wrong)
    with pytest.raises(FishVocError, match="hop_length must be"):
        _hifigan_engine(dict(g["cfg"], hop_length=17), sd)
    with pytest.raises(FishVocError, match="noise_convs"):
        _hifigan_engine(dict(g["cfg"], use_template=True), sd)      # template generator needs the noise_convs weights


def test_bigvgan_golden_and_oracle():
    from vocoder_amd import _lib
    g = load_golden("bigvgan_tiny.npz")
    sd = syn.bigvgan_state_dict(g["cfg"], g["seed"])
    y = _fwd(_hifigan_engine(g["cfg"], sd, _lib.FV_MODEL_BIGVGAN), g["mel"])
    err = np.abs(y - g["out"]).max()
    assert err <= TOL, f"max|d| = {err:.3e} vs (stand-in backed) golden"
    cfg = dict(syn.BIGVGAN_24K)
    sd = syn.bigvgan_state_dict(cfg, seed=2)
    mel = syn.synthetic_mel(1, 80, 7, seed=5)
    ref = orc.bigvgan_forward(sd, cfg, mel)
    y = _fwd(_hifigan_engine(cfg, sd, _lib.FV_MODEL_BIGVGAN), mel)
    err = np.abs(y - ref).max()
    assert err <= TOL, f"max|d| = {err:.3e} vs oracle"


@pytest.mark.parametrize("name,kw", [("bigvgan_24k_t6.npz", {}), ("bigvgan_snake_post.npz", {"post_beta": False})])
def test_bigvgan_full_width_and_snake_post_goldens(name, kw):
    """Engine and drop-in module against the reference captures: the full-width config[2] generator, and activation=Snake
    (activation_post only; the AMPBlocks keep SnakeBeta and their beta keys — reference bigvgan.py:330,335-337)."""
    from vocoder_amd import _lib
    from vocoder_amd.modules.generators import bigvgan as bv
    g = load_golden(name)
    sd = syn.bigvgan_state_dict(g["cfg"], g["seed"], **kw)
    y = _fwd(_hifigan_engine(g["cfg"], sd, _lib.FV_MODEL_BIGVGAN), g["mel"])
    assert np.abs(y - g["out"]).max() <= TOL
    gen = bv.BigVGANGenerator(**g["cfg"], activation=bv.Snake if kw else bv.SnakeBeta).eval()
    keys = set(gen.state_dict().keys())
    assert ("activation_post.act.beta" in keys) == (not kw) and "resblocks.0.activations.0.act.beta" in keys
    missing, unexpected = gen.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("filter") for k in missing), (missing, unexpected)
    y2 = gen.to(_dev())(torch.from_numpy(g["mel"]).to(_dev()))
    assert np.abs(y2.cpu().numpy() - g["out"]).max() <= TOL


def test_vocos_full_depth_golden_and_oracle():
    """BASELINE config[3] at its real depth [3, 3, 27, 3] / dims [128 .. 1024] (vocos.yaml:4-8): reference capture (B=2, T=10)
    and the oracle on another seed."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config
    g = load_golden("vocos_24k_t10.npz")
    cfg = g["cfg"]
    assert cfg["backbone"]["depths"] == [3, 3, 27, 3]
    sd = syn.vocos_state_dict(cfg, g["seed"])
    eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]),
                 state_dict=sd)
    y = _fwd(eng, g["mel"])
    assert np.abs(y - g["out"]).max() <= _peak_tol(g["out"]), (np.abs(y - g["out"]).max(), np.abs(g["out"]).max())
    mel = syn.synthetic_mel(2, 80, 10, seed=61)
    ref = orc.vocos_forward(sd, cfg, mel)
    y = _fwd(eng, mel)
    assert np.abs(y - ref).max() <= _peak_tol(ref), (np.abs(y - ref).max(), np.abs(ref).max())


def test_convnext_and_vocos():
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config
    g = load_golden("convnext_small.npz")
    sd = syn.convnext_state_dict(g["cfg"], g["seed"])
    eng = Engine(_lib.FV_MODEL_CONVNEXT, backbone=convnext_config(**g["cfg"]), state_dict=sd)
    y = _fwd(eng, g["mel"])
    err = np.abs(y - g["out"]).max()
    assert err <= TOL, f"convnext max|d| = {err:.3e}"

    g = load_golden("istft_head.npz")
    sd = syn.istft_head_state_dict(g["cfg"], g["seed"])
    eng = Engine(_lib.FV_MODEL_ISTFT_HEAD, head=istft_head_config(**g["cfg"]), state_dict=sd)
    y = _fwd(eng, g["x"])
    err = np.abs(y[:, 0] - g["wave"]).max()
    assert err <= _peak_tol(g["wave"]), f"istft head max|d| = {err:.3e} (peak {np.abs(g['wave']).max():.3f})"

    g = load_golden("vocos_tiny.npz")
    sd = syn.vocos_state_dict(g["cfg"], g["seed"])
    eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**g["cfg"]["backbone"]),
                 head=istft_head_config(**g["cfg"]["head"]), state_dict=sd)
    y = _fwd(eng, g["mel"])
    err = np.abs(y - g["out"]).max()
    assert err <= _peak_tol(g["out"]), f"vocos max|d| = {err:.3e} (peak {np.abs(g['out']).max():.3f})"


def test_vocos_24k_vs_oracle():
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config
    cfg = dict(backbone=dict(input_channels=80, depths=[1, 1, 2, 1], dims=[128, 256, 512, 1024], kernel_size=7),
               head=dict(dim=1024, n_fft=1024, hop_length=256, win_length=1024, padding="same"))
    sd = syn.vocos_state_dict(cfg, seed=3)
    mel = syn.synthetic_mel(2, 80, 10, seed=9)
    ref = orc.vocos_forward(sd, cfg, mel)
    eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]),
                 state_dict=sd)
    y = _fwd(eng, mel)
    err = np.abs(y - ref).max()
    assert err <= _peak_tol(ref), f"vocos-24k max|d| = {err:.3e} (peak {np.abs(ref).max():.3f})"


def test_graph_replay_and_branch_streams_match_first_eager_call():
    """Calls 1-2 with the same buffers run eagerly (branch streams), call 3+ replays the captured hipGraph; new data written
    into the same input buffer must flow through the replay."""
    cfg = dict(syn.HIFIGAN_V1_44K)
    sd = syn.hifigan_state_dict(cfg, seed=2)
    eng = _hifigan_engine(cfg, sd)
    mel_a = torch.from_numpy(syn.synthetic_mel(3, 80, 10, seed=1)).to(_dev())
    mel_b = torch.from_numpy(syn.synthetic_mel(3, 80, 10, seed=2)).to(_dev())
    x = mel_a.clone()
    out = torch.empty((3, 1, 10 * 512), device=_dev())
    eng(x, out)
    torch.cuda.synchronize()
    first = out.clone()
    for _ in range(4):
        eng(x, out)
    torch.cuda.synchronize()
    assert torch.equal(out, first)
    x.copy_(mel_b)
    eng(x, out)
    torch.cuda.synchronize()
    ref_b = orc.hifigan_forward(sd, cfg, mel_b.cpu().numpy())
    assert np.abs(out.cpu().numpy() - ref_b).max() <= TOL
    ref_a = orc.hifigan_forward(sd, cfg, mel_a.cpu().numpy())
    assert np.abs(first.cpu().numpy() - ref_a).max() <= TOL
    # on a user-provided non-default stream as well
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            eng(x, out)
    s.synchronize()
    assert np.abs(out.cpu().numpy() - ref_b).max() <= TOL


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_branch_mean_in_the_next_upsampler_is_bit_identical_to_the_ordered_accumulate(prec, monkeypatch):
    """With branch streams the three ResBlock branches keep their own outputs and the next stage's upsampler conv (the last
    stage: mean_of_three_kernel) forms ((y0 + y1) + y2) / 3; on one stream the branches accumulate into the stage output in
    order.  Same additions, same order: the two engines must agree to the last bit, single clip and full batch."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, upsampler_config
    for kind, cfg, sdf, shapes in ((_lib.FV_MODEL_HIFIGAN, dict(syn.HIFIGAN_V1_44K), syn.hifigan_state_dict, ((1, 86), (32, 86), (5, 33))),
                                   (_lib.FV_MODEL_BIGVGAN, dict(syn.BIGVGAN_24K), syn.bigvgan_state_dict, ((1, 94), (9, 40)))):
        sd = sdf(cfg, 3)
        monkeypatch.setenv("FV_SINGLE_STREAM", "1")
        chain = Engine(kind, ups=upsampler_config(**cfg), state_dict=sd, precision=prec)
        monkeypatch.delenv("FV_SINGLE_STREAM")
        tree = Engine(kind, ups=upsampler_config(**cfg), state_dict=sd, precision=prec)
        for B, T in shapes:
            mel = torch.from_numpy(syn.synthetic_mel(B, cfg["num_mels"], T, seed=B + T)).to(_dev())
            ya = chain(mel).clone()
            yb = tree(mel).clone()
            yb2 = tree(mel).clone()   # (second call: captured)
            yb3 = tree(mel).clone()   # (third: replayed)
            torch.cuda.synchronize()
            assert torch.equal(ya, yb) and torch.equal(ya, yb2) and torch.equal(ya, yb3), (kind, prec, B, T, float((ya - yb).abs().max()))


def test_long_clip_and_odd_lengths_vs_oracle():
    """A 35 s clip at hop 16 (T_mel = 1501, prime-ish) and a 2-frame clip through the tiny config: exercises many tiles,
    ragged last tiles in every stage and the fused-pair kernels' halo logic at both ends."""
    g = load_golden("hifigan_tiny.npz")
    cfg = g["cfg"]
    sd = syn.hifigan_state_dict(cfg, 12)
    eng = _hifigan_engine(cfg, sd)
    for B, T in ((1, 1501), (3, 2), (2, 257)):
        mel = syn.synthetic_mel(B, cfg["num_mels"], T, seed=T)
        ref = orc.hifigan_forward(sd, cfg, mel)
        y = _fwd(eng, mel)
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() <= TOL, (B, T, float(np.abs(y - ref).max()))


def test_narrow_stage_configs_use_fused_pairs_and_match():
    """C0 = 256 -> stages of 128, 64, 32, 16 channels with k in (3, 7, 11): the 32/16-channel stages run the fused
    (c1, c2) pair kernels (16x16x4 and 32x32x2 MFMA variants)."""
    cfg = dict(hop_length=32, upsample_rates=[2, 2, 2, 4], upsample_kernel_sizes=[4, 4, 4, 8],
               resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=24,
               upsample_initial_channel=256, use_template=False, pre_conv_kernel_size=7, post_conv_kernel_size=7)
    sd = syn.hifigan_state_dict(cfg, 21)
    mel = syn.synthetic_mel(2, 24, 70, seed=3)
    ref = orc.hifigan_forward(sd, cfg, mel)
    y = _fwd(_hifigan_engine(cfg, sd), mel)
    assert np.abs(y - ref).max() <= TOL


def test_template_branch_golden_and_module():
    """use_template=True: forward(x, template) with the strided noise_convs (reference hifigan.py:226-234)."""
    from vocoder_amd.modules.generators import HiFiGANGenerator
    from vocoder_amd.engine import FishVocError
    g = load_golden("hifigan_template.npz")
    sd = syn.hifigan_state_dict(g["cfg"], g["seed"])
    gen = HiFiGANGenerator(**g["cfg"])
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    gen = gen.eval().cuda()
    y = gen(torch.from_numpy(g["mel"]).cuda(), template=torch.from_numpy(g["template"]).cuda())
    assert np.abs(y.cpu().numpy() - g["out"]).max() <= TOL
    with pytest.raises(TypeError):
        gen(torch.from_numpy(g["mel"]).cuda())
    eng = _hifigan_engine(g["cfg"], sd)
    with pytest.raises(FishVocError, match="needs a template"):
        eng(torch.from_numpy(g["mel"]).cuda())
    # full V1 config with a template vs the oracle
    cfg = dict(syn.HIFIGAN_V1_44K, use_template=True)
    sd = syn.hifigan_state_dict(cfg, 5)
    mel = syn.synthetic_mel(2, 80, 7, seed=8)
    tmpl = np.random.default_rng(1).normal(0, 0.5, size=(2, 1, 7 * 512)).astype(np.float32)
    ref = orc.hifigan_forward(sd, cfg, mel, template=tmpl)
    y = _hifigan_engine(cfg, sd)(torch.from_numpy(mel).cuda(), None, torch.from_numpy(tmpl).cuda())
    torch.cuda.synchronize()
    assert np.abs(y.cpu().numpy() - ref).max() <= TOL


def test_bigvgan_template_branch_golden_module_oracle_and_batch_invariance():
    """BigVGANGenerator(use_template=True) — the reference ctor DEFAULT (bigvgan.py:267): x = x + noise_convs[i](template) after every
    upsampler (bigvgan.py:300-330,359-360).  VERDICT r5 missing 2: the engine ran this forward with nothing checking it.  Reference capture
    (B = 3, T = 13) through the drop-in module and the raw engine; the 24 kHz configuration (strided noise convs of 64 / 8 / 2 / 1) at ragged
    T, B > 1 against the oracle; a clip alone bit-identical to the clip inside the batch in batch-invariant mode."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, FishVocError, upsampler_config
    from vocoder_amd.modules.generators import BigVGANGenerator
    dev = _dev()
    g = load_golden("bigvgan_template.npz")
    assert g["cfg"]["use_template"] is True
    sd = syn.bigvgan_state_dict(g["cfg"], g["seed"])
    gen = BigVGANGenerator(**g["cfg"])
    missing, unexpected = gen.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("filter") for k in missing), (missing, unexpected)
    gen = gen.eval().to(dev)
    xt, tt = torch.from_numpy(g["mel"]).to(dev), torch.from_numpy(g["template"]).to(dev)
    y = gen(xt, template=tt)
    assert y.shape == g["out"].shape
    assert np.abs(y.cpu().numpy() - g["out"]).max() <= TOL
    with pytest.raises(TypeError):
        gen(xt)
    eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**g["cfg"]), state_dict=sd)
    with pytest.raises(FishVocError, match="needs a template"):
        eng(xt)
    assert np.abs(eng(xt, None, tt).cpu().numpy() - g["out"]).max() <= TOL
    # the template must matter (a branch that silently dropped it would still be finite)
    assert float((eng(xt, None, torch.zeros_like(tt)) - y).abs().max()) > 1e-3
    # BASELINE config[2]'s generator with the template branch on, ragged T, B = 3, against the oracle
    cfg = dict(syn.BIGVGAN_24K, use_template=True)
    sd = syn.bigvgan_state_dict(cfg, 4)
    B, T = 3, 37
    mel = syn.synthetic_mel(B, cfg["num_mels"], T, seed=12)
    tmpl = syn.synthetic_template(B, T, cfg["hop_length"], seed=13)
    ref = orc.bigvgan_forward(sd, cfg, mel, template=tmpl)
    eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=sd)
    xt, tt = torch.from_numpy(mel).to(dev), torch.from_numpy(tmpl).to(dev)
    y = eng(xt, None, tt).clone()
    y2 = eng(xt, None, tt).clone()       # captured
    y3 = eng(xt, None, tt).clone()       # replayed
    assert torch.equal(y, y2) and torch.equal(y, y3)
    assert np.abs(y.cpu().numpy() - ref).max() <= TOL, np.abs(y.cpu().numpy() - ref).max()
    eng.set_batch_invariant(True)
    yb = eng(xt, None, tt).clone()
    assert np.abs(yb.cpu().numpy() - ref).max() <= TOL
    for i in range(B):
        yi = eng(xt[i:i + 1].contiguous(), None, tt[i:i + 1].contiguous())
        assert torch.equal(yi[0], yb[i]), f"clip {i} alone differs from the clip inside the batch (batch-invariant mode)"


@pytest.mark.parametrize("name", ["refinegan_tiny.npz", "refinegan_rates.npz", "refinegan_slope.npz"])
def test_refinegan_golden(name):
    """RefineGAN through the engine (fv_forward_refinegan) and through the drop-in module, against the reference capture."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, refinegan_config
    from vocoder_amd.modules.generators.refinegan import RefineGANGenerator
    g = load_golden(name)
    cfg = g["cfg"]
    sd = syn.refinegan_state_dict(cfg, g["seed"])
    B, _, T = g["mel"].shape
    noise = np.concatenate([n.reshape(-1) for n in syn.refinegan_noise(cfg, B, T, seed=int(g["noise_seed"]))])
    eng = Engine(_lib.FV_MODEL_REFINEGAN, refine=refinegan_config(**cfg), state_dict=sd)
    assert eng.noise_elems(B, T) == noise.size
    dev = _dev()
    y = eng(torch.from_numpy(g["mel"]).to(dev), None, torch.from_numpy(g["template"]).to(dev),
            torch.from_numpy(noise).to(dev)).cpu().numpy()
    # the captures peak at ~0.1: the bar is 1e-4 of the expected waveform's own peak, as for Vocos (refinegan.py:287-323)
    tol = min(TOL, _peak_tol(g["out"]))
    err = np.abs(y - g["out"]).max()
    assert err <= tol, f"{name}: max|d| = {err:.3e} (bar {tol:.1e} = 1e-4 of the peak {np.abs(g['out']).max():.3f})"
    gen = RefineGANGenerator(**cfg).eval()
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    gen = gen.to(dev)
    y2 = gen(torch.from_numpy(g["mel"]).to(dev), torch.from_numpy(g["template"]).to(dev), torch.from_numpy(noise).to(dev))
    assert np.abs(y2.cpu().numpy() - g["out"]).max() <= tol
    y3 = gen(torch.from_numpy(g["mel"]).to(dev), torch.from_numpy(g["template"]).to(dev))   # own torch.randn draws
    assert y3.shape == y2.shape and bool(torch.isfinite(y3).all()) and float((y3 - y2).abs().max()) > 0


def test_refinegan_default_config_vs_oracle():
    """The reference's default widths / rates (start_channels=16, (2,2,8,8)/(8,8,2,2), hop 256) on a short clip."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, refinegan_config
    cfg = dict(syn.REFINEGAN_44K)
    sd = syn.refinegan_state_dict(cfg, seed=4)
    B, T = 2, 5
    mel = syn.synthetic_mel(B, cfg["num_mels"], T, seed=2)
    tmpl = syn.synthetic_template(B, T, cfg["hop_length"], seed=3)
    noise = syn.refinegan_noise(cfg, B, T, seed=5)
    ref = orc.refinegan_forward(sd, cfg, mel, tmpl, noise)
    eng = Engine(_lib.FV_MODEL_REFINEGAN, refine=refinegan_config(**cfg), state_dict=sd)
    dev = _dev()
    y = eng(torch.from_numpy(mel).to(dev), None, torch.from_numpy(tmpl).to(dev),
            torch.from_numpy(np.concatenate([n.reshape(-1) for n in noise])).to(dev)).cpu().numpy()
    assert np.abs(y - ref).max() <= TOL, np.abs(y - ref).max()
    with pytest.raises(Exception):
        eng(torch.from_numpy(mel).to(dev))   # template + noise are mandatory


@pytest.mark.parametrize("dims,T,B", [([24, 100], 3, 2), ([72, 200, 520], 33, 1), ([130], 1, 3),
                                      # widths that leave whole 64-channel chunks of the kernel's 256 / 512 / 1024 register
                                      # budget unused (Firefly's 384 did: uninitialised registers reached the variance)
                                      ([320], 12, 2), ([384], 40, 3), ([640, 192], 64, 4), ([448, 896], 5, 1)])
def test_convnext_odd_widths_and_short_clips_vs_oracle(dims, T, B):
    """Channel counts that are not multiples of the 64-row LDS chunks of the tiled dwconv+LN kernel, clips shorter than its
    32-column tile (and than the 7-tap halo)."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config
    cfg = dict(input_channels=11, depths=[1] * len(dims), dims=dims, kernel_size=7)
    sd = syn.convnext_state_dict(cfg, seed=21)
    x = syn.synthetic_mel(B, 11, T, seed=4)
    ref = orc.convnext_forward(sd, cfg, x)
    eng = Engine(_lib.FV_MODEL_CONVNEXT, backbone=convnext_config(**cfg), state_dict=sd)
    y = _fwd(eng, x)
    err = np.abs(y - ref).max()
    assert err <= 2e-5 * max(np.abs(ref).max(), 1.0), f"max|d| = {err:.3e}"


def test_refinegan_single_frame_and_rejected_lengths():
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, refinegan_config
    cfg = dict(sampling_rate=16000, hop_length=16, downsample_rates=(2, 2, 2, 2), upsample_rates=(2, 2, 2, 2),
               leaky_relu_slope=0.2, num_mels=12, start_channels=4)
    sd = syn.refinegan_state_dict(cfg, seed=2)
    eng = Engine(_lib.FV_MODEL_REFINEGAN, refine=refinegan_config(**cfg), state_dict=sd)
    dev = _dev()
    mel = syn.synthetic_mel(1, 12, 1, seed=1)
    tmpl = syn.synthetic_template(1, 1, 16, seed=2)
    noise = syn.refinegan_noise(cfg, 1, 1, seed=3)
    ref = orc.refinegan_forward(sd, cfg, mel, tmpl, noise)
    y = eng(torch.from_numpy(mel).to(dev), None, torch.from_numpy(tmpl).to(dev),
            torch.from_numpy(np.concatenate([n.reshape(-1) for n in noise])).to(dev)).cpu().numpy()
    assert np.abs(y - ref).max() <= TOL
    with pytest.raises(ValueError):   # wrong noise length
        eng(torch.from_numpy(mel).to(dev), None, torch.from_numpy(tmpl).to(dev), torch.zeros(5, device=dev))


def test_engine_lifecycle_releases_device_memory_and_empty_batch():
    """fv_create / fv_forward / fv_destroy in a loop must give back everything it allocated (weights, packed planes,
    graphs, branch streams); an empty batch returns an empty waveform like the reference's convs do."""
    cfg, sd = dict(syn.HIFIGAN_V1_44K), syn.hifigan_state_dict(syn.HIFIGAN_V1_44K, 0)
    x = torch.from_numpy(syn.synthetic_mel(2, 80, 20, 5)).to(_dev())

    stream = torch.cuda.Stream()   # one caller stream for every cycle (on the legacy default stream each engine would
                                   # draw a fresh side stream from torch's lazily created pool of 32)

    def cycle():
        eng = _hifigan_engine(cfg, sd)
        with torch.cuda.stream(stream):
            for _ in range(3):      # eager, capture, replay
                y = eng(x)
            empty = eng(x[:0])
        torch.cuda.synchronize()
        assert tuple(empty.shape) == (0, 1, 20 * 512) and empty.dtype == torch.float32
        eng.close()
        return y

    ref = cycle().clone()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(6):
        assert torch.equal(cycle(), ref)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 32 << 20, f"{(free0 - free1) >> 20} MiB lost over 6 create/destroy cycles (one engine holds ~170 MiB)"


def test_two_engines_on_two_host_threads_match_single_threaded_results():
    """include/fishvoc.h: one engine per caller thread is re-entrant (distinct workspaces and streams)."""
    import threading
    cfg = dict(hop_length=16, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4], resblock_kernel_sizes=[3, 7, 11],
               resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=20, upsample_initial_channel=64, use_template=False)
    sds = [syn.hifigan_state_dict(cfg, s) for s in (1, 2)]
    xs = [torch.from_numpy(syn.synthetic_mel(3, 20, 40 + 7 * i, 11 + i)).to(_dev()) for i in range(2)]
    engines = [_hifigan_engine(cfg, sd) for sd in sds]
    want = [eng(x).clone() for eng, x in zip(engines, xs)]
    torch.cuda.synchronize()
    got, errs = [None, None], []

    def work(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(20):
                    y = engines[i](xs[i])
                s.synchronize()
            got[i] = y
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_one_engine_shared_by_two_threads_and_streams_is_serialised():
    """One engine owns one workspace and one set of branch streams: forwards from two host threads on two torch streams are
    serialised by the handle (host lock + event chain), so they cannot corrupt each other."""
    import threading
    cfg = dict(hop_length=16, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4], resblock_kernel_sizes=[3, 7, 11],
               resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=20, upsample_initial_channel=64, use_template=False)
    eng = _hifigan_engine(cfg, syn.hifigan_state_dict(cfg, 5))
    xs = [torch.from_numpy(syn.synthetic_mel(4, 20, 150 + 40 * i, 31 + i)).to(_dev()) for i in range(2)]
    want = [eng(x).clone() for x in xs]
    torch.cuda.synchronize()
    bad, errs = [0, 0], []

    def work(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(25):
                    y = eng(xs[i])
                    s.synchronize()
                    bad[i] += int(not torch.equal(y, want[i]))
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert bad == [0, 0], bad


_RES = {"24000_256_1024": dict(num_mels=100, n_fft=1024, hop_length=256, win_length=1024),
        "44100_512_2048": dict(num_mels=128, n_fft=2048, hop_length=512, win_length=2048)}


@pytest.mark.parametrize("name,depths,dims,res", [
    ("vocos-small", [8], [512], "24000_256_1024"),                              # configs/model/generator/vocos-small.yaml
    ("vocos-small", [8], [512], "44100_512_2048"),
    ("vocos", [1, 1, 2, 1], [128, 256, 512, 1024], "44100_512_2048"),            # vocos.yaml widths at the 44.1 kHz resolution
    ("vocos-huge", [1, 1, 1, 1], [352, 704, 1408, 2816], "24000_256_1024"),      # vocos-huge.yaml widths (depths cut for the oracle)
    ("vocos-small-vae decoder", [2, 1], [512, 1024], "24000_256_1024"),
])
def test_shipped_vocos_configs_vs_oracle(name, depths, dims, res):
    """The widths / resolutions the reference ships YAMLs for (not only the benchmark config) against the CPU oracle."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config
    r = _RES[res]
    cfg = dict(backbone=dict(input_channels=r["num_mels"], depths=depths, dims=dims, kernel_size=7),
               head=dict(dim=dims[-1], n_fft=r["n_fft"], hop_length=r["hop_length"], win_length=r["win_length"], padding="same"))
    sd = syn.vocos_state_dict(cfg, seed=len(name))
    mel = syn.synthetic_mel(2, r["num_mels"], 9, seed=2)
    ref = orc.vocos_forward(sd, cfg, mel)
    for prec in ("f32", "f16x3"):
        eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]),
                     state_dict=sd, precision=prec)
        y = _fwd(eng, mel)
        assert y.shape == ref.shape == (2, 1, 9 * r["hop_length"])
        err = np.abs(y - ref).max()
        assert err <= _peak_tol(ref), f"{name} @ {res} ({prec}): max|d| = {err:.3e} (ref max {np.abs(ref).max():.3f})"


def test_shipped_hifigan_vae_decoder_config_vs_oracle():
    """configs/model/generator/hifigan-vae.yaml decoder: hop 640 = 8 * 5 * 4 * 2 * 2 (a stride-5 and a stride-4 stage),
    512 input features."""
    cfg = dict(hop_length=640, upsample_rates=[8, 5, 4, 2, 2], upsample_kernel_sizes=[16, 10, 8, 4, 4],
               resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=512,
               upsample_initial_channel=512, use_template=False, pre_conv_kernel_size=7, post_conv_kernel_size=7)
    sd = syn.hifigan_state_dict(cfg, 2)
    mel = syn.synthetic_mel(2, 512, 3, seed=8) * 0.2
    ref = orc.hifigan_forward(sd, cfg, mel)
    y = _fwd(_hifigan_engine(cfg, sd), mel)
    # k - stride is odd at the stride-5 stage (padding (10 - 5) // 2 = 2): the reference emits 5 T + 1 samples there, so
    # 640 T + 16 in total (checked against the reference itself: 1936 for T = 3, oracle within 6e-8 of it)
    assert y.shape == ref.shape == (2, 1, 3 * 640 + 16)
    assert np.abs(y - ref).max() <= TOL, np.abs(y - ref).max()


def test_istft_head_with_a_hop_that_does_not_divide_n_fft_vs_oracle():
    """resolution/24000_2048_3072.yaml: hop 2048, n_fft = win = 3072 (1.5 hops per frame)."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, istft_head_config
    cfg = dict(dim=64, n_fft=3072, hop_length=2048, win_length=3072, padding="same")
    sd = syn.istft_head_state_dict(cfg, 1)
    x = (np.random.default_rng(3).normal(size=(2, 64, 5)) * 0.5).astype(np.float32)
    ref = orc.istft_head_forward(sd, cfg, x)
    eng = Engine(_lib.FV_MODEL_ISTFT_HEAD, head=istft_head_config(**cfg), state_dict=sd)
    y = _fwd(eng, x)
    assert y.shape == (2, 1, 5 * 2048) and ref.shape == (2, 5 * 2048)   # ISTFTHead returns (B, T); the engine adds the channel axis (unify.py:30-31)
    err = np.abs(y[:, 0] - ref).max()
    assert err <= _peak_tol(ref), f"max|d| = {err:.3e} (ref max {np.abs(ref).max():.3f})"


def test_bigvgan_long_clip_covers_the_interior_snake_tiles_vs_oracle():
    """The anti-aliased snake takes a clamp-free path for tiles whose 1024 samples + halo lie inside the clip: stages of
    3000 and 6000 samples have both kinds (the goldens and the 7-frame BigVGAN-24k case above only have edge tiles)."""
    from vocoder_amd import _lib
    cfg = dict(hop_length=4, upsample_rates=[2, 2], upsample_kernel_sizes=[4, 4], resblock_kernel_sizes=[3, 7],
               resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]], num_mels=12, upsample_initial_channel=32, use_template=False,
               pre_conv_kernel_size=7, post_conv_kernel_size=7)
    sd = syn.bigvgan_state_dict(cfg, seed=11)
    mel = syn.synthetic_mel(2, 12, 1500, seed=6)
    ref = orc.bigvgan_forward(sd, cfg, mel)
    for prec in ("f32", "f16x3"):
        from vocoder_amd.engine import Engine, upsampler_config
        eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=sd, precision=prec)
        y = _fwd(eng, mel)
        assert y.shape == ref.shape == (2, 1, 6000)
        assert np.abs(y - ref).max() <= TOL, (prec, np.abs(y - ref).max())


def _large_alpha_bigvgan(c0, seed):
    """A one-stage BigVGAN whose SnakeBeta parameters look like TRAINED log-scale ones (bigvgan.py:121-135: alpha = exp(p), p up to ~3) instead
    of the N(0, 0.3) the synthetic state dicts draw: log-alpha, log-beta ~ N(2.5, 0.5) (alpha up to ~50), and a conv_pre gain that puts |u| ~ 5
    in front of the first activation — |alpha u| in the hundreds of radians, where the phase of sin^2 decides the result."""
    cfg = dict(hop_length=2, upsample_rates=[2], upsample_kernel_sizes=[4], resblock_kernel_sizes=[3, 7, 11],
               resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=20, upsample_initial_channel=c0, use_template=False,
               pre_conv_kernel_size=7, post_conv_kernel_size=7)
    sd = syn.bigvgan_state_dict(cfg, seed=seed)
    rng = np.random.default_rng(seed + 1000)
    for k in sd:
        if k.endswith(".act.alpha") or k.endswith(".act.beta"):
            sd[k] = rng.normal(2.5, 0.5, size=sd[k].shape).astype(np.float32)
    for name, gain in (("conv_pre", 1.6), ("ups.0", 1.6), ("conv_post", 0.35)):   # |u| rms ~4.5, max ~18; waveform peak ~0.9 (tanh not saturated)
        sd[f"{name}.parametrizations.weight.original0"] = sd[f"{name}.parametrizations.weight.original0"] * np.float32(gain)
    return cfg, sd


@pytest.mark.parametrize("c0,path", [(64, "amp_conv"), (256, "aa_snake")])
def test_bigvgan_trained_scale_snake_parameters_vs_oracle(c0, path):
    """VERDICT r4 item 2 / ADVICE r3: the snake runs on v_cos_f32 with a compensated phase (small_kernels.hip snake2, amp_conv.hip); this drives it
    at |alpha u| of 100 - 400 rad through both forms — C = 32 (k = 3 convs fused with their activation: amp_conv; k = 7 / 11: aa_snake + conv) and
    C = 128 (aa_snake + Winograd convs) — against the oracle's sinf(fl(u alpha)) (bigvgan.py:133-135), 19 activations deep."""
    from vocoder_amd import _lib
    cfg, sd = _large_alpha_bigvgan(c0, seed=c0)
    mel = syn.synthetic_mel(2, 20, 700, seed=3)
    col = {}
    ref = orc.bigvgan_forward(sd, cfg, mel, collect=col)
    eng = _hifigan_engine(cfg, sd, _lib.FV_MODEL_BIGVGAN)
    prof = eng.profile(torch.from_numpy(mel).to(_dev()), repeats=1)
    assert any(r["kernel"].startswith(path) for r in prof), [r["kernel"] for r in prof]
    y = _fwd(eng, mel)
    err = np.abs(y - ref).max()
    amax = max(float(np.exp(v).max()) for k, v in sd.items() if k.endswith(".act.alpha"))
    print(f"trained-scale snake C0={c0}: max alpha {amax:.1f}, |u| into the first activation up to {np.abs(col['ups.0']).max():.1f}, "
          f"waveform peak {np.abs(ref).max():.3f}, max|d| = {err:.2e}")
    assert amax > 30.0
    assert err <= TOL, f"max|d| = {err:.3e} vs oracle"


def test_differential_fuzz_of_random_configurations():
    """tools/fuzz_{hifigan,vocos,refinegan,conv,logmel,sequence,firefly}.py: random (but seeded) generator configurations, batch sizes and clip lengths
    through the engine in both precisions against the oracle, plus graph capture / replay identity."""
    import importlib.util, os
    tools = os.path.join(os.path.dirname(__file__), "..", "tools")
    for name, kw in (("fuzz_hifigan", dict(n_cases=6, seed=11)), ("fuzz_hifigan", dict(n_cases=2, seed=12, large=True)),
                     ("fuzz_hifigan", dict(n_cases=3, seed=16, model="bigvgan")),   # case 2 draws use_template=True
                     ("fuzz_vocos", dict(n_cases=6, seed=13)), ("fuzz_refinegan", dict(n_cases=4, seed=14)),
                     ("fuzz_conv", dict(n_cases=40, seed=16)), ("fuzz_logmel", dict(n_cases=8, seed=17)),
                     ("fuzz_sequence", dict(n_calls=40, seed=18)), ("fuzz_firefly", dict(n_cases=4, seed=19))):
        spec = importlib.util.spec_from_file_location(name, os.path.join(tools, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        assert mod.run(verbose=False, **kw) <= 2e-4   # each tool asserts its own per-case bound (log-mel: 2e-4 on log values)


def test_template_branch_with_a_long_first_stage_stride_vs_oracle():
    """use_template=True with rates (2, 8, 8, 4): the first noise conv has stride 256 and 512 taps (more than 64 KiB of LDS
    in the strided-conv kernel; found by tools/fuzz_hifigan.py)."""
    cfg = dict(hop_length=512, upsample_rates=[2, 8, 8, 4], upsample_kernel_sizes=[4, 16, 16, 8], resblock_kernel_sizes=[3],
               resblock_dilation_sizes=[[1, 3, 5]], num_mels=20, upsample_initial_channel=32, use_template=True,
               pre_conv_kernel_size=7, post_conv_kernel_size=7)
    sd = syn.hifigan_state_dict(cfg, 4)
    mel = syn.synthetic_mel(2, 20, 5, seed=3)
    tmpl = syn.synthetic_template(2, 5, 512, seed=4)
    ref = orc.hifigan_forward(sd, cfg, mel, template=tmpl)
    eng = _hifigan_engine(cfg, sd)
    y = eng(torch.from_numpy(mel).to(_dev()), None, torch.from_numpy(tmpl).to(_dev()))
    torch.cuda.synchronize()
    assert np.abs(y.cpu().numpy() - ref).max() <= TOL


def test_bigvgan_and_vocos_at_the_baseline_batch_sizes_vs_oracle():
    """BASELINE config[2] (BigVGAN-24k, B = 64) and config[3] (Vocos-24k, depths [3,3,27,3], B = 128) at their stated sizes,
    94 frames = 1 s: finite, deterministic, every checked item equal to the same clip run alone, and the first / last items
    against the CPU oracle run on the whole 1 s clip."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config
    cfg = dict(syn.BIGVGAN_24K)
    sdb = syn.bigvgan_state_dict(cfg, 0)
    eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=sdb)
    cfgv = dict(syn.VOCOS_24K)
    sdv = syn.vocos_state_dict(cfgv, 0)
    engv = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfgv["backbone"]), head=istft_head_config(**cfgv["head"]),
                  state_dict=sdv)
    for e, B, oracle in ((eng, 64, lambda m: orc.bigvgan_forward(sdb, cfg, m)), (engv, 128, lambda m: orc.vocos_forward(sdv, cfgv, m))):
        mel = syn.synthetic_mel(B, 80, 94, seed=B)
        y = _fwd(e, mel)
        assert y.shape == (B, 1, 94 * 256) and np.isfinite(y).all()
        assert np.array_equal(y, _fwd(e, mel))
        # BigVGAN waveforms are O(1) (tanh): absolute bar; Vocos ones are small with the synthetic head: bar relative to the peak
        scale = 1.0 if e is eng else float(np.abs(y).max())
        for i in (0, B // 2, B - 1):
            yi = _fwd(e, mel[i:i + 1])
            assert np.abs(yi[0] - y[i]).max() <= 3e-5 * scale, (i, np.abs(yi[0] - y[i]).max(), scale)
        for i in (0, B - 1):
            ref = oracle(mel[i:i + 1])
            err = np.abs(ref[0] - y[i]).max()
            assert err <= TOL * scale, f"B={B} item {i}: max|d| = {err:.3e} vs oracle on the full clip (scale {scale:.3f})"
        e.close()


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_bigvgan_and_vocos_benchmark_sizes_properties(prec):
    """BASELINE config[2] / config[3] geometry at a batch that uses the full-size tiles (BigVGAN-24k B=12, Vocos-24k B=40,
    94 frames): finite, deterministic, and every batch item equal to the same clip run alone — the pointwise convs of the
    ConvNeXt trunk tile the flattened (batch, time) axis, so items share tiles."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config
    cfg = dict(syn.BIGVGAN_24K)
    eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0), precision=prec)
    cfgv = dict(syn.VOCOS_24K)
    engv = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfgv["backbone"]), head=istft_head_config(**cfgv["head"]),
                  state_dict=syn.vocos_state_dict(cfgv, 0), precision=prec)
    for e, B, tol in ((eng, 12, 3e-5), (engv, 40, 3e-5)):
        mel = syn.synthetic_mel(B, 80, 94, seed=B)
        y = _fwd(e, mel)
        assert y.shape == (B, 1, 94 * 256) and np.isfinite(y).all()
        assert np.array_equal(y, _fwd(e, mel))
        scale = max(1.0, np.abs(y).max())
        for i in (0, B // 2, B - 1):
            yi = _fwd(e, mel[i:i + 1])
            assert np.abs(yi[0] - y[i]).max() <= tol * scale, (i, np.abs(yi[0] - y[i]).max(), scale)
        e.close()


def test_no_writes_outside_the_output_and_the_declared_workspace():
    """Guard bands around the caller's output tensor and around exactly fv_workspace_bytes of workspace stay untouched, and a
    workspace poisoned with NaNs before every call changes nothing (no kernel reads workspace it did not write in this
    forward) — HiFiGAN with and without template, BigVGAN, Vocos, Firefly, RefineGAN, log-mel; both precisions; odd sizes."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import (Engine, convnext_config, istft_head_config, logmel_config, refinegan_config,
                                    upsampler_config)
    dev = _dev()
    G = 4096   # guard floats on each side
    hcfg = dict(hop_length=16, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4], resblock_kernel_sizes=[3, 7, 11],
                resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=20, upsample_initial_channel=128, use_template=False)
    vcfg = dict(backbone=dict(input_channels=20, depths=[1, 2], dims=[64, 192], kernel_size=7),
                head=dict(dim=192, n_fft=64, hop_length=16, win_length=64, padding="same"))
    fcfg = dict(backbone=dict(input_channels=20, depths=[1], dims=[64], kernel_size=7), head=dict(hcfg, num_mels=64))
    rcfg = dict(sampling_rate=16000, hop_length=16, downsample_rates=(2, 2, 4), upsample_rates=(4, 2, 2), leaky_relu_slope=0.2,
                num_mels=20, start_channels=4)
    mcfg = dict(sample_rate=16000, n_fft=64, win_length=64, hop_length=16, n_mels=20)

    def engines(prec):
        yield "hifigan", Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**hcfg), state_dict=syn.hifigan_state_dict(hcfg, 1), precision=prec), 20, None
        tc = dict(hcfg, use_template=True)
        yield "hifigan+template", Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**tc), state_dict=syn.hifigan_state_dict(tc, 1), precision=prec), 20, "template"
        yield "bigvgan", Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**hcfg), state_dict=syn.bigvgan_state_dict(hcfg, 1), precision=prec), 20, None
        yield "vocos", Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**vcfg["backbone"]), head=istft_head_config(**vcfg["head"]),
                              state_dict=syn.vocos_state_dict(vcfg, 1), precision=prec), 20, None
        yield "firefly", Engine(_lib.FV_MODEL_FIREFLY, backbone=convnext_config(**fcfg["backbone"]), ups=upsampler_config(**fcfg["head"]),
                                state_dict=syn.firefly_state_dict(fcfg, 1), precision=prec), 20, None
        yield "refinegan", Engine(_lib.FV_MODEL_REFINEGAN, refine=refinegan_config(**rcfg), state_dict=syn.refinegan_state_dict(rcfg, 1), precision=prec), 20, "refine"
        yield "logmel", Engine(_lib.FV_MODEL_LOGMEL, mel=logmel_config(**mcfg), state_dict={}, precision=prec), 1, "wave"

    for prec in ("f32", "f16x3"):
        for name, eng, cin, extra in engines(prec):
            ran = 0
            for B, T in ((1, 1), (3, 37), (2, 130)):
                if extra == "wave":
                    T = 16 * T + 48
                xbuf = torch.full((B * cin * T + 2 * G,), float("nan"), device=dev)   # NaNs either side of the input as well:
                x = xbuf[G:G + B * cin * T].view(B, cin, T)                            # a stray read that is used would show
                x.copy_(torch.randn(B, cin, T, device=dev) * (0.1 if extra == "wave" else 1.0))
                L = eng.output_length(T)
                if L <= 0:
                    continue
                n_out = B * eng.out_channels * L
                obuf = torch.full((n_out + 2 * G,), 7.25, device=dev)
                out = obuf[G:G + n_out].view(B, eng.out_channels, L)
                need = (eng.workspace_bytes(B, T) + 3) // 4
                wbuf = torch.full((need + 2 * G,), -3.5, device=dev)
                eng._ws = wbuf[G:G + need]
                kw = {}
                if extra == "template":
                    kw["template"] = torch.randn(B, 1, L, device=dev) * 0.3
                if extra == "refine":
                    kw["template"] = torch.randn(B, 1, L, device=dev) * 0.3
                    kw["noise"] = torch.randn(eng.noise_elems(B, T), device=dev)
                try:
                    eng._ws.zero_()
                    eng(x, out, **kw)
                    torch.cuda.synchronize()
                    first = out.clone()
                    for _ in range(3):   # eager, capture, replay — each time over a workspace full of NaNs: nothing may be
                        eng._ws.fill_(float("nan"))   # read before this forward wrote it (halo columns, padded rows, ...)
                        eng(x, out, **kw)
                        torch.cuda.synchronize()
                        assert torch.equal(out, first), (name, prec, B, T, "result depends on stale workspace contents")
                except ValueError:
                    continue             # lengths RefineGAN cannot join
                assert eng._ws.data_ptr() == wbuf[G:].data_ptr(), "the engine replaced a workspace of the size it asked for"
                assert bool((obuf[:G] == 7.25).all()) and bool((obuf[G + n_out:] == 7.25).all()), (name, prec, B, T, "output guard")
                assert bool((wbuf[:G] == -3.5).all()) and bool((wbuf[G + need:] == -3.5).all()), (name, prec, B, T, "workspace guard")
                assert bool(torch.isfinite(out).all()), (name, prec, B, T)
                ran += 1
            assert ran >= 2, (name, prec, ran)
            eng.close()


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_benchmark_size_kernels_do_not_read_stale_workspace(prec):
    """The full-size tile shapes and fused pair kernels (HiFiGAN-V1 B=8, BigVGAN-24k B=4, Vocos-24k B=16, 86 / 94 frames):
    the same forward over a zeroed and over a NaN-filled workspace must give identical, finite waveforms."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config
    dev = _dev()
    h, b, v = dict(syn.HIFIGAN_V1_44K), dict(syn.BIGVGAN_24K), dict(syn.VOCOS_24K)
    cases = [(Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**h), state_dict=syn.hifigan_state_dict(h, 0), precision=prec), 8, 86),
             (Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**b), state_dict=syn.bigvgan_state_dict(b, 0), precision=prec), 4, 94),
             (Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**v["backbone"]), head=istft_head_config(**v["head"]),
                     state_dict=syn.vocos_state_dict(v, 0), precision=prec), 16, 94)]
    for eng, B, T in cases:
        x = torch.from_numpy(syn.synthetic_mel(B, 80, T, seed=B)).to(dev)
        need = (eng.workspace_bytes(B, T) + 3) // 4
        eng._ws = torch.zeros(need, device=dev)
        first = eng(x).clone()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(first).all())
        for _ in range(3):
            eng._ws.fill_(float("nan"))
            y = eng(x)
            torch.cuda.synchronize()
            assert torch.equal(y, first)
        eng.close()


@pytest.mark.parametrize("model,prec", [("bigvgan", "f32"), ("bigvgan", "f16x3"), ("hifigan", "f32"), ("hifigan", "f16x3")])
def test_single_clip_forward_is_bitwise_repeatable_across_branch_streams(model, prec):
    """A single-clip forward runs its three MRF branches on three streams (eager here: graph replay off).  25 repeats must be
    bit-identical: kernels that consume anything they did not write themselves (registers, LDS, another stream's buffer)
    show up as a few hundred samples that change from run to run.  (Round 2: a faster aa_snake variant made 9 % of the
    f16x3 BigVGAN single-clip forwards differ by up to 3e-2 around multiples of 256 samples; reverted, this test guards it.)"""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, upsampler_config
    if model == "bigvgan":
        cfg = dict(syn.BIGVGAN_24K)
        eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0), precision=prec)
        T = 94
    else:
        cfg = dict(syn.HIFIGAN_V1_44K)
        eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=syn.hifigan_state_dict(cfg, 0), precision=prec)
        T = 86
    eng.set_graph_replay(False)
    mel = torch.from_numpy(syn.synthetic_mel(1, 80, T, seed=12)).to(_dev())
    out = torch.empty((1, 1, eng.output_length(T)), device=_dev())
    eng(mel, out)
    torch.cuda.synchronize()
    ref = out.clone()
    differing = 0
    for _ in range(25):
        eng(mel, out)
        torch.cuda.synchronize()
        differing += int(not torch.equal(out, ref))
    eng.close()
    assert differing == 0, f"{differing} of 25 repeats differ from the first run"


def test_hifigan_batch_256_the_eight_gpu_global_batch_on_one_gpu():
    """BASELINE config[4]'s global batch (256 one-second clips; 32 per GPU over 8 GPUs in the reference's launch,
    configs/trainer/default.yaml:6-9) pushed through ONE engine: finite, deterministic, items equal to the same clip run alone,
    and the reference's own 1 s capture (hifigan_v1_t86.npz, hifigan.py:226-249) placed at the last position."""
    g = load_golden("hifigan_v1_t86.npz")
    sd = syn.hifigan_state_dict(g["cfg"], g["seed"])
    eng = _hifigan_engine(g["cfg"], sd)
    B = 256
    mel = syn.synthetic_mel(B, 80, 86, seed=4321)
    mel[B - 1] = g["mel"][0]
    x = torch.from_numpy(mel).to(_dev())
    y_t = eng(x)
    torch.cuda.synchronize()
    y = y_t.cpu().numpy()
    assert y.shape == (B, 1, 44032) and np.isfinite(y).all() and np.abs(y).max() <= 1.0
    assert torch.equal(eng(x), y_t)                      # run to run
    for i in (0, 128, 255):
        yi = _fwd(eng, mel[i:i + 1])
        assert np.abs(yi[0] - y[i]).max() <= 2e-5, (i, np.abs(yi[0] - y[i]).max())
    err = np.abs(y[B - 1] - g["out"][0]).max()
    assert err <= TOL, f"item 255 vs the reference capture: max|d| = {err:.3e}"
    # the same 256 clips as eight contiguous shards of 32 (what the eight ranks compute): bit for bit the global batch
    for r in (0, 3, 7):
        ys = eng(x[32 * r:32 * (r + 1)])
        torch.cuda.synchronize()
        assert torch.equal(ys, y_t[32 * r:32 * (r + 1)]), r
    eng.close()


def test_bench_runs_its_collectives_on_a_one_rank_rccl_group():
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (backend nccl = RCCL): the weights go through
    broadcast_state_dict, the timed with_collectives loop runs broadcast -> forward -> all_gather on the process group, and the
    collected batch equals the direct forward bit for bit.  A plain `python bench.py` (what the driver runs) builds the same
    1-rank group itself."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--no-alt-precision"]
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    for launcher in (["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                      "--master-port", str(port)], []):
        r = subprocess.run([sys.executable] + launcher + [os.path.join(repo, "bench.py")] + common, capture_output=True, text=True,
                           timeout=900, env=env, cwd=repo)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        j = json.loads(lines[0])
        assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and "rccl_error" not in j, j.get("rccl_error")
        c = j["with_collectives"]
        assert c["backend"] == "rccl" and c["collected_finite"] and c["rank0_shard_equals_direct_forward"], c
        assert c["collected_shape"] == [32, 1, 44032] and c["all_gather_bytes_per_step"] == 32 * 44032 * 4
        assert j["output_finite"] and j["ms_per_step_eager"] > 0 and j["mixed_shapes"]["output_finite"]


@pytest.fixture
def direct_sums(monkeypatch):
    """Bit-for-bit comparisons between launch sequences hold between kernels that add the same products in the same order: keep the
    Winograd F(2,3) convs (conv_wino_impl.h, same results to ~1e-6 but other sums) out of both sides."""
    from vocoder_amd import _lib
    monkeypatch.setenv("FV_WINO", "0")
    _lib.reload_env()
    yield
    monkeypatch.delenv("FV_WINO")
    _lib.reload_env()


@pytest.mark.parametrize("maxk", ["3", "11"])
def test_bigvgan_fused_amp_convs_equal_the_activation_plus_conv_launches(maxk, monkeypatch, direct_sums):
    """amp_conv.hip (AMPBlock conv with its anti-aliased SnakeBeta fused in front, bigvgan.py:235-245) against the aa_snake + conv
    launches it replaces: the same arithmetic in the same order, so full-size tiles agree bit for bit; every kernel size (the
    default fuses k = 3 only), sequence ends inside a tile (replicate padding of both FIRs, zero padding of the conv), ragged T;
    and both against the CPU oracle."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, upsampler_config
    cfg = dict(syn.BIGVGAN_24K)
    sd = syn.bigvgan_state_dict(cfg, 3)
    monkeypatch.setenv("FV_AMP_MAXK", maxk)
    monkeypatch.setenv("FV_AMP_MAXC", "64")   # (the default fuses at C = 32 only since round 6, LOG R6.13: both widths of amp_conv stay covered here)
    fused = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=sd)
    monkeypatch.setenv("FV_NO_AMP_FUSION", "1")
    plain = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=sd)
    monkeypatch.delenv("FV_NO_AMP_FUSION")
    x = torch.from_numpy(syn.synthetic_mel(40, 80, 31, seed=8)).to(_dev())     # enough columns for the tiled conv kernels
    ya, yb = fused(x), plain(x)
    torch.cuda.synchronize()
    prof = fused.profile(x, repeats=1)
    assert any(r["kernel"].startswith("amp_conv<k=3") and "C=64" in r["kernel"] for r in prof)
    assert any(r["kernel"].startswith("amp_conv<k=3") and "C=32" in r["kernel"] for r in prof)
    assert any(r["kernel"].startswith("amp_conv<k=11") for r in prof) == (maxk == "11")
    assert torch.equal(ya, yb), float((ya - yb).abs().max())
    for B, T in ((2, 7), (1, 1), (3, 19)):                                       # short clips: both sequence ends inside one tile
        mel = syn.synthetic_mel(B, 80, T, seed=T)
        ref = orc.bigvgan_forward(sd, cfg, mel)
        for eng in (fused, plain):
            err = np.abs(_fwd(eng, mel) - ref).max()
            assert err <= TOL, (B, T, err)
    fused.close()
    plain.close()


def test_round3_fusions_are_bit_identical_to_the_launches_they_replace(monkeypatch, direct_sums):
    """Round-3 engine paths against the launch sequences they replace, on a batch large enough to take them (C = 128 pairs need
    tiles >= 2 x CUs): k = 3 (c1, c2) pairs at C = 128 (FV_PAIR_MAXC=64 keeps them per layer) and the last stage's branch mean formed
    inside conv_post (FV_NO_POST_SUM3=1 keeps mean_of_three_kernel).  Same k-step order / same additions: bit for bit."""
    cfg = dict(syn.HIFIGAN_V1_44K)
    sd = syn.hifigan_state_dict(cfg, seed=5)
    x = torch.from_numpy(syn.synthetic_mel(12, 80, 86, seed=77)).to(_dev())
    base = _hifigan_engine(cfg, sd)
    y = base(x)
    torch.cuda.synchronize()
    prof = base.profile(x, repeats=1)
    assert any(r["kernel"].startswith("resblock_pair<k=3") and "C=128" in r["kernel"] for r in prof), [r["kernel"] for r in prof][:12]
    for var, val in (("FV_PAIR_MAXC", "64"), ("FV_NO_POST_SUM3", "1")):
        monkeypatch.setenv(var, val)
        eng = _hifigan_engine(cfg, sd)
        monkeypatch.delenv(var)
        y2 = eng(x)
        torch.cuda.synchronize()
        assert torch.equal(y, y2), (var, float((y - y2).abs().max()))
        eng.close()
    base.close()


@pytest.mark.parametrize("model", ["hifigan", "bigvgan"])
def test_winograd_convs_against_the_direct_sums_and_the_oracle(model, monkeypatch):
    """conv_wino4_impl.h / conv_wino_impl.h (Winograd F(4,3) / F(2,3) tap groups for the dilated k = 7, 11 / k = 3 convs of launches that fill the
    chip) in the whole forward: the B = 32 / 64 step takes it (profile), equals the direct-sum engine within 2e-5 of full scale (the north_star bar is 1e-4; both
    are ~2e-6 from a float64 forward, tools/experiments/winograd_precision.py), is deterministic, and the same engine forced onto
    the Winograd kernels for short clips (FV_WINO=2) stays within the bar of the CPU oracle."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, upsampler_config
    if model == "hifigan":
        cfg = dict(syn.HIFIGAN_V1_44K)
        sd = syn.hifigan_state_dict(cfg, seed=9)
        mk = lambda: _hifigan_engine(cfg, sd)   # noqa: E731
        B, T, ref_fn = 32, 86, orc.hifigan_forward
    else:
        cfg = dict(syn.BIGVGAN_24K)
        sd = syn.bigvgan_state_dict(cfg, 9)
        mk = lambda: Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=sd)   # noqa: E731
        B, T, ref_fn = 64, 47, orc.bigvgan_forward
    x = torch.from_numpy(syn.synthetic_mel(B, 80, T, seed=55)).to(_dev())
    try:
        eng = mk()
        y = eng(x).clone()
        torch.cuda.synchronize()
        prof = eng.profile(x, repeats=1)
        wino = [r["kernel"] for r in prof if r["kernel"].startswith(("conv_wino<", "conv_wino4<", "conv_wino44<"))]
        assert any("k=11" in k for k in wino) and any("k=7" in k for k in wino), [r["kernel"] for r in prof][:20]
        assert torch.equal(eng(x), y)
        monkeypatch.setenv("FV_WINO", "0")
        _lib.reload_env()
        direct = mk()   # (a fresh engine: the first one replays its captured launch sequence for this input)
        y0 = direct(x).clone()
        torch.cuda.synchronize()
        assert not any(r["kernel"].startswith(("conv_wino<", "conv_wino4<", "conv_wino44<")) for r in direct.profile(x, repeats=1))
        direct.close()
        d = float((y - y0).abs().max())
        assert 0 < d <= 2e-5, d
        monkeypatch.setenv("FV_WINO", "2")
        _lib.reload_env()
        for b, t in ((2, 9), (1, 1), (3, 24)):
            mel = syn.synthetic_mel(b, 80, t, seed=t)
            ref = ref_fn(sd, cfg, mel)
            err = np.abs(_fwd(eng, mel) - ref).max()
            assert err <= TOL, (b, t, err)
        assert any(r["kernel"].startswith(("conv_wino<", "conv_wino4<", "conv_wino44<")) for r in eng.profile(torch.from_numpy(syn.synthetic_mel(1, 80, 9, seed=1)).to(_dev()), repeats=1))
        eng.close()
    finally:
        monkeypatch.delenv("FV_WINO", raising=False)
        _lib.reload_env()


def test_ragged_shards_against_the_global_batch_and_the_batch_invariant_switch():
    """Utterance sharding with a ragged split (37 one-second clips over 8 ranks: 5,5,5,5,5,4,4,4 — vocoder_amd.sharding.shard_slice; the
    reference's 8-device launch is configs/trainer/default.yaml:6-9).  The default engine chooses direct sums or Winograd tap groups per
    LAUNCH (>= one workgroup per CU), so a 4-clip shard and the 37-clip batch may add a clip's products in another order: the outputs agree to
    <= 2e-5 of full scale (the parity bar is 1e-4).  With fv_set_batch_invariant every kernel choice follows from the layer shape alone and
    the shards equal the global batch — and a clip run alone — BIT FOR BIT; fv_set_conv_algorithm(direct) + batch-invariant likewise."""
    from vocoder_amd.sharding import shard_sizes, shard_slice
    cfg = dict(syn.HIFIGAN_V1_44K)
    sd = syn.hifigan_state_dict(cfg, seed=21)
    B, W = 37, 8
    assert shard_sizes(B, W) == [5, 5, 5, 5, 5, 4, 4, 4]
    x = torch.from_numpy(syn.synthetic_mel(B, 80, 86, seed=99)).to(_dev())
    eng = _hifigan_engine(cfg, sd)
    y = eng(x).clone()
    torch.cuda.synchronize()
    worst = 0.0
    for r in (0, 4, 5, 7):
        sl = shard_slice(B, W, r)
        ys = eng(x[sl])
        torch.cuda.synchronize()
        worst = max(worst, float((ys - y[sl]).abs().max()))
    assert worst <= 2e-5, worst
    for algo in ("auto", "direct"):
        eng.set_conv_algorithm(algo)
        eng.set_batch_invariant(True)
        yi = eng(x).clone()
        torch.cuda.synchronize()
        assert float((yi - y).abs().max()) <= 2e-5
        for r in range(W):
            sl = shard_slice(B, W, r)
            ys = eng(x[sl])
            torch.cuda.synchronize()
            assert torch.equal(ys, yi[sl]), (algo, r, float((ys - yi[sl]).abs().max()))
        y1 = eng(x[17:18])
        torch.cuda.synchronize()
        assert torch.equal(y1, yi[17:18]), algo
        prof = [r["kernel"] for r in eng.profile(x[:4], repeats=1)]
        assert not any("splitK" in k for k in prof), prof[:10]
        assert any(k.startswith(("conv_wino<", "conv_wino4<", "conv_wino44<")) for k in prof) == (algo == "auto"), prof[:10]
        eng.set_batch_invariant(False)
    eng.close()


# ---- long clips: time tiles with halo recompute, run as a batch (engine.hip run_model; SURVEY §5 long-context row, VERDICT r4 missing 3) ----
def _tile_layout(T, L, halo):
    """The engine's plan (engine.hip TilePlan): n tiles of L frames, regular starts i * (L - 2 halo), the last pulled back to end at T;
    tile i contributes the frames [lo_i, hi_i)."""
    stride = L - 2 * halo
    n = max(2, -(-(T - 2 * halo) // stride))
    start = [min(i * stride, T - L) for i in range(n)]
    lo = [0 if i == 0 else (i - 1) * stride + L - halo for i in range(n)]
    hi = [T if i + 1 == n else i * stride + L - halo for i in range(n)]
    return start, lo, hi


@pytest.mark.parametrize("model", ["hifigan", "hifigan_template", "bigvgan", "bigvgan_template", "vocos", "firefly"])
def test_time_tiled_forward_matches_the_whole_clip_and_the_oracle(model, monkeypatch):
    """FV_TILE_FRAMES forces the long-clip path (a batch of overlapping time tiles whose halo outputs are discarded) on clips that also run
    whole: tiled == whole to the last-bit differences of the Winograd lattices' anchoring (<= 2e-5), both within the parity bar of the
    oracle; ragged last tile, B = 2."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config
    tiny = dict(hop_length=16, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4], resblock_kernel_sizes=[3, 7, 11],
                resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=20, upsample_initial_channel=64, use_template=False,
                pre_conv_kernel_size=7, post_conv_kernel_size=7)
    T, B = 389, 2
    tmpl = None
    if model in ("hifigan", "hifigan_template"):
        cfg = dict(tiny, use_template=model.endswith("template"))
        sd = syn.hifigan_state_dict(cfg, seed=5)
        mk = lambda: Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd)   # noqa: E731
        mel = syn.synthetic_mel(B, 20, T, seed=9)
        if cfg["use_template"]:
            tmpl = syn.synthetic_template(B, T, 16, seed=3)
        ref = orc.hifigan_forward(sd, cfg, mel, template=tmpl)
    elif model in ("bigvgan", "bigvgan_template"):
        cfg = dict(tiny, use_template=model.endswith("template"))
        sd = syn.bigvgan_state_dict(cfg, seed=6)
        mk = lambda: Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=sd)   # noqa: E731
        mel = syn.synthetic_mel(B, 20, T, seed=9)
        if cfg["use_template"]:
            tmpl = syn.synthetic_template(B, T, 16, seed=3)
        ref = orc.bigvgan_forward(sd, cfg, mel, template=tmpl)
    else:
        bb = dict(input_channels=20, depths=[1, 2], dims=[32, 64], drop_path_rate=0.0, kernel_size=7)
        if model == "vocos":
            cfg = dict(backbone=bb, head=dict(dim=64, n_fft=64, hop_length=16, win_length=64, padding="same"))
            sd = syn.vocos_state_dict(cfg, seed=7)
            mk = lambda: Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**bb), head=istft_head_config(**cfg["head"]), state_dict=sd)   # noqa: E731
            ref = orc.vocos_forward(sd, cfg, mel := syn.synthetic_mel(B, 20, T, seed=9))
        else:
            cfg = dict(backbone=bb, head=dict(tiny, num_mels=64))
            sd = syn.firefly_state_dict(cfg, seed=8)
            mk = lambda: Engine(_lib.FV_MODEL_FIREFLY, backbone=convnext_config(**bb), ups=upsampler_config(**cfg["head"]), state_dict=sd)   # noqa: E731
            ref = orc.firefly_forward(sd, cfg, mel := syn.synthetic_mel(B, 20, T, seed=9))
    tol = min(TOL, _peak_tol(ref)) if model == "vocos" else TOL
    dev = _dev()
    xt = torch.from_numpy(mel).to(dev)
    tt = None if tmpl is None else torch.from_numpy(tmpl).to(dev)
    whole = mk()
    y_whole = whole(xt, None, tt).cpu().numpy()
    ws_whole = whole.workspace_bytes(B, T)
    monkeypatch.setenv("FV_TILE_FRAMES", "150")     # read at fv_create
    tiled = mk()
    monkeypatch.delenv("FV_TILE_FRAMES")
    assert tiled.workspace_bytes(B, T) != ws_whole           # the tile plan is in force
    y_tiled = tiled(xt, None, tt).cpu().numpy()
    y_again = tiled(xt, None, tt).cpu().numpy()               # (second call: captured graph)
    assert y_tiled.shape == y_whole.shape == ref.shape
    assert np.array_equal(y_tiled, y_again)
    assert np.abs(y_whole - ref).max() <= tol
    assert np.abs(y_tiled - ref).max() <= tol, np.abs(y_tiled - ref).max()
    assert np.abs(y_tiled - y_whole).max() <= 2e-5 * max(1.0, float(np.abs(ref).max()))
    prof = tiled.profile(xt, repeats=1) if tmpl is None else []
    assert tmpl is not None or {"gather_tiles", "scatter_tiles"} <= {r["kernel"] for r in prof}


def test_generators_whose_output_is_not_frames_times_hop_are_not_tiled(monkeypatch):
    """a ConvTranspose1d with odd (kernel - rate) yields T u + 1 samples per stage, so fv_output_length(T) != T * hop — the
    tile gather / scatter address rows at exactly frames x hop and would shift every tile after the first.  Such a generator must run
    whole even when FV_TILE_FRAMES asks for tiles (the plan declines), and still agree with the oracle."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, upsampler_config
    cfg = dict(hop_length=16, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[7, 4, 5], resblock_kernel_sizes=[3, 7],
               resblock_dilation_sizes=[[1, 3, 5]] * 2, num_mels=20, upsample_initial_channel=64, use_template=False,
               pre_conv_kernel_size=7, post_conv_kernel_size=7)
    sd = syn.hifigan_state_dict(cfg, seed=5)
    B, T = 2, 389
    mel = syn.synthetic_mel(B, 20, T, seed=9)
    ref = orc.hifigan_forward(sd, cfg, mel)
    assert ref.shape[-1] != T * 16                              # (4 T + 1) * 2 * 2 + 1 samples
    dev = _dev()
    whole = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd)
    monkeypatch.setenv("FV_TILE_FRAMES", "150")
    tiled = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd)
    monkeypatch.delenv("FV_TILE_FRAMES")
    assert tiled.output_length(T) == ref.shape[-1]
    assert tiled.workspace_bytes(B, T) == whole.workspace_bytes(B, T)       # no tile plan
    xt = torch.from_numpy(mel).to(dev)
    y = tiled(xt).cpu().numpy()
    assert y.shape == ref.shape and np.abs(y - ref).max() <= TOL
    assert "gather_tiles" not in {r["kernel"] for r in tiled.profile(xt, repeats=1)}
    assert np.array_equal(y, whole(xt).cpu().numpy())


def test_tile_batches_past_the_grid_z_limit(monkeypatch):
    """gather / scatter put (clip, tile) in gridDim.z (limit 65 535): 6 000 clips x 12 tiles = 72 000 items now go in z chunks; and
    a plan past 2^24 tiles is refused with a message instead of running on an empty workspace."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, FishVocError, convnext_config
    cfg = dict(input_channels=4, depths=[1], dims=[8], drop_path_rate=0.0, kernel_size=7)
    sd = syn.convnext_state_dict(cfg, 2)
    monkeypatch.setenv("FV_TILE_FRAMES", "8")       # the plan raises it to 4 x the reach (8 frames): tiles of 32 frames, stride 16
    tiled = Engine(_lib.FV_MODEL_CONVNEXT, backbone=convnext_config(**cfg), state_dict=sd)
    monkeypatch.delenv("FV_TILE_FRAMES")
    whole = Engine(_lib.FV_MODEL_CONVNEXT, backbone=convnext_config(**cfg), state_dict=sd)
    B, T = 6000, 200
    assert tiled.workspace_bytes(1, T) != whole.workspace_bytes(1, T)
    dev = _dev()
    x7 = torch.from_numpy(syn.synthetic_mel(7, 4, T, seed=3)).to(dev)
    x = x7.repeat(B // 7 + 1, 1, 1)[:B].contiguous()
    y, yw = tiled(x), whole(x)
    torch.cuda.synchronize()
    assert float((y - yw).abs().max()) <= 2e-5                  # every (clip, tile) item, the ones past z = 65 535 included
    ref = orc.convnext_forward(sd, cfg, x7.cpu().numpy())
    assert np.abs(y[B - 7 - B % 7:B - B % 7].cpu().numpy() - ref).max() <= TOL
    del x, y, yw
    # 130 clips x 131 071 tiles > 2^24: refused by name (the check precedes the workspace test)
    with pytest.raises(FishVocError, match="time tiles"):
        tiled(torch.zeros((130, 4, 2 ** 21), device=dev))


def test_thirty_minute_clip_runs_as_time_tiles_bit_identical_to_a_run_of_one_tiles_frames():
    """A 30-minute clip at 44.1 kHz (T_mel = 155 040 -> 79.4 M samples; one stage-4 tensor would be 5 GB, past the 4 GiB addressing span the
    per-layer kernels enforce, conv_layer.hip) through fv_forward: three tiles of 65 536 frames run as one batch.  In batch-invariant mode the
    samples a tile contributes are BIT-identical to a stand-alone forward of that tile's own frames (a shorter, overlapping run); a window of
    the clip's middle and its first frames agree with the oracle."""
    cfg = dict(syn.HIFIGAN_V1_44K)
    sd = syn.hifigan_state_dict(cfg, seed=0)
    eng = _hifigan_engine(cfg, sd)
    eng.set_batch_invariant(True)
    T = 155_040
    mel = syn.synthetic_mel(1, 80, T, seed=77)
    xt = torch.from_numpy(mel).to(_dev())
    assert eng.output_length(T) * 16 * 4 > 2 ** 32            # the clip's last-stage tensor does not fit the addressing span
    y = eng(xt)
    torch.cuda.synchronize()
    assert y.shape == (1, 1, T * 512)
    assert bool(torch.isfinite(y).all()) and float(y.abs().max()) <= 1.0
    L = 65536
    halo = 17   # the plan: 2^29 / 8192 = 65 536 frames per tile; halo = the engine's reach bound (15 + 2 frames for V1: engine.hip ups_reach)
    start, lo, hi = _tile_layout(T, L, halo)
    assert len(start) == 3
    i = 1
    yi = eng(xt[:, :, start[i]:start[i] + L].contiguous())
    a, b = (lo[i] - start[i]) * 512, (hi[i] - start[i]) * 512
    assert torch.equal(yi[0, 0, a:b], y[0, 0, lo[i] * 512:hi[i] * 512]), "tile 1's samples differ from a stand-alone run of its frames"
    # the seams: the last frame of tile 0 and the first of tile 1 against a whole run of the frames around the seam
    s = lo[1]
    ys = eng(xt[:, :, s - 40:s + 40].contiguous())
    assert float((ys[0, 0, 20 * 512:60 * 512] - y[0, 0, (s - 20) * 512:(s + 20) * 512]).abs().max()) <= 2e-5
    del yi, ys
    yh = y.cpu().numpy()
    ref0 = orc.hifigan_forward(sd, cfg, mel[:, :, :28])
    assert np.abs(ref0[0, 0, :8 * 512] - yh[0, 0, :8 * 512]).max() <= TOL
    m = T // 2
    refm = orc.hifigan_forward(sd, cfg, mel[:, :, m - 24:m + 32])
    assert np.abs(refm[0, 0, 24 * 512:32 * 512] - yh[0, 0, m * 512:(m + 8) * 512]).max() <= TOL
    refe = orc.hifigan_forward(sd, cfg, mel[:, :, T - 28:])
    assert np.abs(refe[0, 0, -8 * 512:] - yh[0, 0, -8 * 512:]).max() <= TOL


@pytest.mark.parametrize("model", ["bigvgan", "vocos", "firefly", "refinegan"])
def test_batch_invariant_mode_holds_beyond_hifigan(model):
    """fv_set_batch_invariant promises kernel choices from the layer shape alone; pinned so far for HiFiGAN only.  BigVGAN (amp_conv /
    aa_snake / Winograd), Vocos (gemm_pw / dwconv_ln / ISTFT), Firefly and RefineGAN: every clip of a batch equals the same clip run alone and
    inside another batch composition, BIT for bit."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config, refinegan_config, upsampler_config
    dev = _dev()
    tmpl = noise_for = None
    if model == "bigvgan":
        cfg = dict(syn.BIGVGAN_24K)
        eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 3))
        B, T, cin = 20, 47, 80
    elif model == "vocos":
        cfg = dict(syn.VOCOS_24K)
        eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]),
                     state_dict=syn.vocos_state_dict(cfg, 3))
        B, T, cin = 24, 94, 80
    elif model == "firefly":
        cfg = dict(syn.FIREFLY_BASE_44K)
        eng = Engine(_lib.FV_MODEL_FIREFLY, backbone=convnext_config(**cfg["backbone"]), ups=upsampler_config(**cfg["head"]),
                     state_dict=syn.firefly_state_dict(cfg, 3))
        B, T, cin = 12, 43, 128
    else:
        cfg = dict(syn.REFINEGAN_44K)
        eng = Engine(_lib.FV_MODEL_REFINEGAN, refine=refinegan_config(**cfg), state_dict=syn.refinegan_state_dict(cfg, 3))
        B, T, cin = 6, 20, 128
        tmpl = torch.from_numpy(syn.synthetic_template(B, T, cfg["hop_length"], seed=2)).to(dev)
        per = syn.refinegan_noise(cfg, B, T, seed=4)
        noise_for = lambda idx: torch.from_numpy(np.concatenate([n[idx].reshape(-1) for n in per])).to(dev)   # noqa: E731
    eng.set_batch_invariant(True)
    x = torch.from_numpy(syn.synthetic_mel(B, cin, T, seed=31)).to(dev)

    def run(idx):
        idx = list(idx)
        xi = x[idx].contiguous()
        if noise_for is None:
            return eng(xi).clone()
        return eng(xi, None, tmpl[idx].contiguous(), noise_for(idx)).clone()

    y = run(range(B))
    assert bool(torch.isfinite(y).all())
    for i in (0, B // 2, B - 1):
        assert torch.equal(run([i])[0], y[i]), f"{model}: clip {i} alone differs from the same clip inside the batch of {B}"
    sub = [B - 1, 1, B // 2]
    ys = run(sub)
    for j, i in enumerate(sub):
        assert torch.equal(ys[j], y[i]), f"{model}: clip {i} in a batch of 3 differs from the batch of {B}"


def test_batch_invariant_mode_refuses_the_launch_dependent_pointwise_fallback():
    """the pointwise GEMM (gemm_pw.hip) addresses a launch with 32-bit byte offsets; past 4 GiB the default engine moves the layer to the k = 1
    conv kernel, which forms OTHER sums (measured: equal bits at K = 512, different at K = 2048) — a batch-size-dependent choice, so batch-invariant
    mode refuses the launch instead (FV_ERR_UNSUPPORTED: "split the batch").  Reference: convnext.py:130-141."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, FishVocError, convnext_config
    cfg = dict(input_channels=16, depths=[1], dims=[64], drop_path_rate=0.0, kernel_size=7)
    eng = Engine(_lib.FV_MODEL_CONVNEXT, backbone=convnext_config(**cfg), state_dict=syn.convnext_state_dict(cfg, 2))
    B, T = 45_000, 94                                   # hidden layer: 45 000 x 256 x 94 x 4 B = 4.3 GB
    x = torch.zeros((B, 16, T), device=_dev())
    x[:, :, :] = torch.from_numpy(syn.synthetic_mel(1, 16, T, seed=5)).to(_dev())
    eng.set_batch_invariant(True)
    with pytest.raises(FishVocError, match="batch-invariant mode"):
        eng(x)
    y_small = eng(x[:8].contiguous()).clone()
    eng.set_batch_invariant(False)
    y = eng(x)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all())
    assert float((y[:8] - y_small).abs().max()) <= 2e-5 and float((y[-1] - y_small[0]).abs().max()) <= 2e-5


@pytest.mark.gpu
def test_graft_entry_build_then_smoke_in_one_process():
    """`__graft_entry__.build()` followed by `smoke()` in ONE interpreter — the order in which the library and torch come into the process must not matter
    (round 5: with the library opened first, its HIP runtime was a second copy next to torch's and found no device)."""
    import subprocess, sys, os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], capture_output=True, text=True, cwd=repo, timeout=900)
    assert r.returncode == 0 and "smoke OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_engine_built_on_all_host_cores_equals_the_serial_build(monkeypatch):
    """fv_finalize folds / packs / uploads the layers on the host's cores (engine.hip run_jobs; LOG R6.3: the one-shot caller of test.py:31-38 waits for
    engine creation).  A layer's weight-norm fold (hifigan.py:94-100, folded in double) and its fragment forms are one thread's work either way, so the
    engine built with FV_BUILD_THREADS=1 — the serial path — must give the same waveform bit for bit, for the three generator families."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, upsampler_config
    dev = _dev()
    cases = []
    cfg = dict(hop_length=64, upsample_rates=[4, 4, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4], resblock_kernel_sizes=[3, 7, 11],
               resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=80, upsample_initial_channel=256, use_template=False, pre_conv_kernel_size=7,
               post_conv_kernel_size=7)
    cases.append((_lib.FV_MODEL_HIFIGAN, cfg, syn.hifigan_state_dict(cfg, 21), syn.synthetic_mel(3, 80, 41, seed=22)))
    bcfg = dict(syn.BIGVGAN_24K)
    cases.append((_lib.FV_MODEL_BIGVGAN, bcfg, syn.bigvgan_state_dict(bcfg, 23), syn.synthetic_mel(2, 80, 19, seed=24)))
    outs = {}
    for threads in ("1", None):
        if threads is None:
            monkeypatch.delenv("FV_BUILD_THREADS", raising=False)
        else:
            monkeypatch.setenv("FV_BUILD_THREADS", threads)
        for i, (kind, c, sd, mel) in enumerate(cases):
            eng = Engine(kind, ups=upsampler_config(**c), state_dict=sd)
            outs[(threads, i)] = eng(torch.from_numpy(mel).to(dev)).clone()
    torch.cuda.synchronize()
    for i in range(len(cases)):
        assert torch.isfinite(outs[(None, i)]).all()
        assert torch.equal(outs[("1", i)], outs[(None, i)]), f"case {i}: the parallel build differs from the serial one"
