"""Seeded synthetic inputs and state dicts (numpy only, no torch RNG).

There is no network for datasets or checkpoints, so benchmarks and parity tests run on
random-but-fixed weights with the *reference's parameter names and shapes*
(``conv_pre.parametrizations.weight.original{0,1}``, ``ups.{i}…``,
``resblocks.{i}.blocks.{j}.convs{1,2}.{n}…`` — SURVEY.md §8b) and on synthetic log-mel
inputs with the reference's value convention (natural-log mel, floor ln(1e-5) = -11.5129;
/root/reference/fish_vocoder/data/transforms/spectrogram.py:93-94,
scripts/convert_diffsinger_mel.py:9-17).

numpy's ``default_rng(PCG64)`` stream is platform- and version-stable, so the golden
fixtures under ``tests/golden`` store only the seed + config + expected output; the weights
are regenerated identically wherever the tests run.  Unlike the reference's N(0, 0.01)
default init, the scales here keep activations O(1) through the whole stack so that an
absolute 1e-4 parity bar on the waveform is a meaningful test.
"""
from __future__ import annotations

from math import sqrt

import numpy as np

MEL_FLOOR = -11.512925  # ln(1e-5)


def synthetic_mel(batch: int, num_mels: int, frames: int, seed: int = 1234) -> np.ndarray:
    """clamp(N(-5, 2), ln 1e-5, 2) log-mel of shape (B, num_mels, T) fp32."""
    rng = np.random.default_rng(seed)
    x = rng.normal(-5.0, 2.0, size=(batch, num_mels, frames))
    return np.clip(x, MEL_FLOOR, 2.0).astype(np.float32)


def _wn_pair(rng, shape, fan_in, gain):
    """(g, v) such that the folded weight g*v/||v|| has std ~ gain/sqrt(fan_in); norm over dims != 0."""
    v = rng.normal(0.0, 1.0, size=shape)
    norm = np.sqrt((v.reshape(shape[0], -1) ** 2).sum(1))
    target = gain / sqrt(fan_in) * sqrt(np.prod(shape[1:]))   # norm of a row with the target std
    g = target * rng.uniform(0.8, 1.25, size=shape[0])
    del norm
    return (g.reshape((shape[0],) + (1,) * (len(shape) - 1)).astype(np.float32), v.astype(np.float32))


def _put_wn(sd, prefix, rng, shape, fan_in, gain, bias_len, bias_std=0.05):
    g, v = _wn_pair(rng, shape, fan_in, gain)
    sd[f"{prefix}.bias"] = rng.normal(0.0, bias_std, size=bias_len).astype(np.float32)
    sd[f"{prefix}.parametrizations.weight.original0"] = g
    sd[f"{prefix}.parametrizations.weight.original1"] = v


def _put_noise_convs(sd, rng, c0, rates):
    """noise_convs.{i}: Conv1d(1, C_i, 2 s, stride s, padding s // 2) with s = prod(rates[i + 1:]); Conv1d(1, C, 1) for the last
    stage (hifigan.py:192-204, bigvgan.py:311-324).  Plain convs: no weight norm."""
    for i in range(len(rates)):
        ch = c0 // 2 ** (i + 1)
        s_f0 = int(np.prod(rates[i + 1:])) if i + 1 < len(rates) else 1
        k = 2 * s_f0 if i + 1 < len(rates) else 1
        sd[f"noise_convs.{i}.weight"] = rng.normal(0.0, 0.5 / sqrt(k), size=(ch, 1, k)).astype(np.float32)
        sd[f"noise_convs.{i}.bias"] = rng.normal(0.0, 0.05, size=ch).astype(np.float32)


def hifigan_state_dict(cfg: dict, seed: int = 0) -> dict:
    """State dict for HiFiGANGenerator(**cfg) (use_template=False), reference key order
    (/root/reference/fish_vocoder/modules/generators/hifigan.py:158-224)."""
    rng = np.random.default_rng(seed)
    sd: dict = {}
    c0 = cfg["upsample_initial_channel"]
    nm = cfg["num_mels"]
    pk, qk = cfg.get("pre_conv_kernel_size", 7), cfg.get("post_conv_kernel_size", 7)
    _put_wn(sd, "conv_pre", rng, (c0, nm, pk), nm * pk, 0.35, c0)
    if cfg.get("use_template", False):   # reference key order: conv_pre, noise_convs, ups, resblocks, conv_post
        _put_noise_convs(sd, rng, c0, list(cfg["upsample_rates"]))
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        cin, cout = c0 // 2**i, c0 // 2 ** (i + 1)
        # ConvTranspose1d weight is (C_in, C_out, k): weight-norm dim 0 = C_in
        _put_wn(sd, f"ups.{i}", rng, (cin, cout, k), cin * max(k // u, 1), 1.3, cout)
    for i in range(len(cfg["upsample_rates"])):
        ch = c0 // 2 ** (i + 1)
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            for n in range(len(rd)):
                _put_wn(sd, f"resblocks.{i}.blocks.{j}.convs1.{n}", rng, (ch, ch, rk), ch * rk, 1.3, ch)
            for n in range(len(rd)):
                _put_wn(sd, f"resblocks.{i}.blocks.{j}.convs2.{n}", rng, (ch, ch, rk), ch * rk, 0.7, ch)
    ch = c0 // 2 ** len(cfg["upsample_rates"])
    _put_wn(sd, "conv_post", rng, (1, ch, qk), ch * qk, 0.25, 1)
    return sd


def bigvgan_state_dict(cfg: dict, seed: int = 0, with_filters: bool = True, post_beta: bool = True) -> dict:
    """State dict for BigVGANGenerator(**cfg): flat ``resblocks.{i*nk+j}`` AMPBlocks with
    ``activations.{m}.act.{alpha,beta}`` (log-scale) and ``activation_post``
    (/root/reference/fish_vocoder/modules/generators/bigvgan.py:277-349)."""
    rng = np.random.default_rng(seed)
    sd: dict = {}
    c0 = cfg["upsample_initial_channel"]
    nm = cfg["num_mels"]
    pk, qk = cfg.get("pre_conv_kernel_size", 7), cfg.get("post_conv_kernel_size", 7)
    nk = len(cfg["resblock_kernel_sizes"])
    _put_wn(sd, "conv_pre", rng, (c0, nm, pk), nm * pk, 0.35, c0)
    if cfg.get("use_template", False):   # reference key order: conv_pre, noise_convs, ups, ... (bigvgan.py:291-328)
        _put_noise_convs(sd, rng, c0, list(cfg["upsample_rates"]))
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        cin, cout = c0 // 2**i, c0 // 2 ** (i + 1)
        _put_wn(sd, f"ups.{i}", rng, (cin, cout, k), cin * max(k // u, 1), 1.0, cout)
    for i in range(len(cfg["upsample_rates"])):
        ch = c0 // 2 ** (i + 1)
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            p = f"resblocks.{i * nk + j}"
            for n in range(len(rd)):
                _put_wn(sd, f"{p}.convs1.{n}", rng, (ch, ch, rk), ch * rk, 0.9, ch)
            for n in range(len(rd)):
                _put_wn(sd, f"{p}.convs2.{n}", rng, (ch, ch, rk), ch * rk, 0.5, ch)
            for m in range(2 * len(rd)):
                sd[f"{p}.activations.{m}.act.alpha"] = rng.normal(0.0, 0.3, size=ch).astype(np.float32)
                sd[f"{p}.activations.{m}.act.beta"] = rng.normal(0.0, 0.3, size=ch).astype(np.float32)
    ch = c0 // 2 ** len(cfg["upsample_rates"])
    sd["activation_post.act.alpha"] = rng.normal(0.0, 0.3, size=ch).astype(np.float32)
    beta_post = rng.normal(0.0, 0.3, size=ch).astype(np.float32)   # drawn either way: the other tensors keep their values
    if post_beta:   # activation=Snake (bigvgan.py:266,335-337) has no beta on activation_post; the AMPBlocks always do
        sd["activation_post.act.beta"] = beta_post
    _put_wn(sd, "conv_post", rng, (1, ch, qk), ch * qk, 0.25, 1)
    del with_filters  # filter buffers are derived, not random: modules recreate them (kaiser-sinc design)
    return sd


def convnext_state_dict(cfg: dict, seed: int = 0, prefix: str = "") -> dict:
    """State dict for ConvNeXtEncoder(**cfg)
    (/root/reference/fish_vocoder/modules/encoders/convnext.py:146-204)."""
    rng = np.random.default_rng(seed)
    sd: dict = {}
    depths, dims = list(cfg["depths"]), list(cfg["dims"])
    ks = cfg.get("kernel_size", 7)
    cin = cfg["input_channels"]

    def nrm(shape, std):
        return rng.normal(0.0, std, size=shape).astype(np.float32)

    def ln(name, c):
        sd[f"{prefix}{name}.weight"] = (1.0 + 0.1 * rng.normal(size=c)).astype(np.float32)
        sd[f"{prefix}{name}.bias"] = nrm(c, 0.05)

    sd[f"{prefix}downsample_layers.0.0.weight"] = nrm((dims[0], cin, ks), 1.0 / sqrt(cin * ks))
    sd[f"{prefix}downsample_layers.0.0.bias"] = nrm(dims[0], 0.05)
    ln("downsample_layers.0.1", dims[0])
    for i in range(1, len(depths)):
        ln(f"downsample_layers.{i}.0", dims[i - 1])
        sd[f"{prefix}downsample_layers.{i}.1.weight"] = nrm((dims[i], dims[i - 1], 1), 1.0 / sqrt(dims[i - 1]))
        sd[f"{prefix}downsample_layers.{i}.1.bias"] = nrm(dims[i], 0.05)
    for i, (dep, c) in enumerate(zip(depths, dims)):
        for j in range(dep):
            p = f"{prefix}stages.{i}.{j}"
            sd[f"{p}.gamma"] = rng.uniform(0.05, 0.3, size=c).astype(np.float32)
            sd[f"{p}.dwconv.weight"] = nrm((c, 1, ks), 1.0 / sqrt(ks))
            sd[f"{p}.dwconv.bias"] = nrm(c, 0.05)
            sd[f"{p}.norm.weight"] = (1.0 + 0.1 * rng.normal(size=c)).astype(np.float32)
            sd[f"{p}.norm.bias"] = nrm(c, 0.05)
            sd[f"{p}.pwconv1.weight"] = nrm((4 * c, c), 1.0 / sqrt(c))
            sd[f"{p}.pwconv1.bias"] = nrm(4 * c, 0.05)
            sd[f"{p}.pwconv2.weight"] = nrm((c, 4 * c), 1.0 / sqrt(4 * c))
            sd[f"{p}.pwconv2.bias"] = nrm(c, 0.05)
    ln("norm", dims[-1])
    return sd


def istft_head_state_dict(cfg: dict, seed: int = 0, prefix: str = "") -> dict:
    """State dict for ISTFTHead(**cfg): ``out`` = Conv1d(dim, 2*n_fft, 1), ``istft.window`` = hann
    (/root/reference/fish_vocoder/modules/generators/vocos.py:33-41)."""
    rng = np.random.default_rng(seed)
    dim, n_fft = cfg["dim"], cfg["n_fft"]
    w = rng.normal(0.0, 0.6 / sqrt(dim), size=(2 * n_fft, dim, 1)).astype(np.float32)
    w[n_fft:] *= 3.0  # phases: spread over a few radians
    b = rng.normal(0.0, 0.2, size=2 * n_fft).astype(np.float32)
    b[:n_fft] -= 2.0  # log-magnitudes: keep |S| small so the waveform stays inside (-1, 1)
    n = np.arange(cfg["win_length"], dtype=np.float64)
    window = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / cfg["win_length"])).astype(np.float32)
    return {f"{prefix}out.weight": w, f"{prefix}out.bias": b, f"{prefix}istft.window": window}


def vocos_state_dict(cfg: dict, seed: int = 0) -> dict:
    """UnifyGenerator(backbone=ConvNeXtEncoder, head=ISTFTHead) — keys ``backbone.*`` / ``head.*``."""
    sd = convnext_state_dict(cfg["backbone"], seed, "backbone.")
    sd.update(istft_head_state_dict(cfg["head"], seed + 1, "head."))
    return sd


def firefly_state_dict(cfg: dict, seed: int = 0) -> dict:
    """UnifyGenerator(backbone=ConvNeXtEncoder, head=HiFiGANGenerator) (firefly-gan-base.yaml)."""
    sd = convnext_state_dict(cfg["backbone"], seed, "backbone.")
    sd.update({f"head.{k}": v for k, v in hifigan_state_dict(cfg["head"], seed + 1).items()})
    return sd


# Named configurations of BASELINE.json (shapes per config)
HIFIGAN_V1_44K = dict(
    hop_length=512, upsample_rates=[8, 8, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 2, 2],
    resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    num_mels=80, upsample_initial_channel=512, use_template=False,
    pre_conv_kernel_size=7, post_conv_kernel_size=7)

BIGVGAN_24K = dict(
    hop_length=256, upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4],
    resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    num_mels=80, upsample_initial_channel=512, use_template=False,
    pre_conv_kernel_size=7, post_conv_kernel_size=7)

VOCOS_24K = dict(
    backbone=dict(input_channels=80, depths=[3, 3, 27, 3], dims=[128, 256, 512, 1024],
                  drop_path_rate=0.4, kernel_size=7),
    head=dict(dim=1024, n_fft=1024, hop_length=256, win_length=1024, padding="same"))

# Firefly-GAN base (reference configs/model/generator/firefly-gan-base.yaml:1-20 with resolution/44100_512_2048.yaml: 128 mels,
# hop 512): ConvNeXt backbone (128 mel bins -> 512 features) feeding a HiFiGAN head with k = 13 pre / post convs
FIREFLY_BASE_44K = dict(
    backbone=dict(input_channels=128, depths=[3, 3, 9, 3], dims=[128, 256, 384, 512], drop_path_rate=0.2, kernel_size=7),
    head=dict(hop_length=512, upsample_rates=[8, 8, 2, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4, 4],
              resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
              num_mels=512, upsample_initial_channel=512, use_template=False,
              pre_conv_kernel_size=13, post_conv_kernel_size=13))


# RefineGANGenerator (/root/reference/fish_vocoder/modules/generators/refinegan.py:182-323) — no YAML in the reference; these
# are its ctor defaults
REFINEGAN_44K = dict(sampling_rate=44100, hop_length=256, downsample_rates=(2, 2, 8, 8), upsample_rates=(8, 8, 2, 2),
                     leaky_relu_slope=0.2, num_mels=128, start_channels=16)


def refinegan_stage_shapes(cfg: dict, frames: int):
    """[(C, T)] of the AdaIN noise tensors: for every up-sampling stage, C = that stage's output channels and T = its length;
    each stage consumes 3 branches x 2 AdaIN layers of them, in (stage, branch, layer) order."""
    ch = cfg["start_channels"] * 2 ** len(cfg["downsample_rates"]) * 2
    t = frames
    out = []
    for r in cfg["upsample_rates"]:
        ch //= 2
        t *= r
        out.append((ch, t))
    return out


def refinegan_noise(cfg: dict, batch: int, frames: int, seed: int = 0) -> list:
    """Standard-normal tensors standing in for AdaIN's torch.randn_like (refinegan.py:125), in order of use."""
    rng = np.random.default_rng(seed)
    return [rng.normal(size=(batch, c, t)).astype(np.float32)
            for (c, t) in refinegan_stage_shapes(cfg, frames) for _ in range(6)]


def refinegan_state_dict(cfg: dict, seed: int = 0) -> dict:
    """State dict of RefineGANGenerator(**cfg), reference key order (refinegan.py:203-286)."""
    rng = np.random.default_rng(seed)
    sd: dict = {}
    ch = cfg["start_channels"]
    _put_wn(sd, "template_conv", rng, (ch, 1, 7), 7, 1.0, ch)

    def resblock(prefix, cin, cout, k):
        for n in range(3):
            _put_wn(sd, f"{prefix}.convs1.{n}", rng, (cout, cin if n == 0 else cout, k), (cin if n == 0 else cout) * k, 1.3, cout)
        for n in range(3):
            _put_wn(sd, f"{prefix}.convs2.{n}", rng, (cout, cout, k), cout * k, 0.7, cout)

    for i, _ in enumerate(cfg["downsample_rates"]):
        resblock(f"downsample_blocks.{i}.1", ch, ch * 2, 7)
        ch *= 2
    _put_wn(sd, "mel_conv", rng, (ch, cfg["num_mels"], 7), cfg["num_mels"] * 7, 0.35, ch)
    ch *= 2
    for i, _ in enumerate(cfg["upsample_rates"]):
        cin, cout = ch + ch // 4, ch // 2
        p = f"upsample_conv_blocks.{i}"
        sd[f"{p}.input_conv.weight"] = rng.normal(0.0, 1.0 / sqrt(cin * 7), size=(cout, cin, 7)).astype(np.float32)
        sd[f"{p}.input_conv.bias"] = rng.normal(0.0, 0.05, size=cout).astype(np.float32)
        for j, k in enumerate((3, 7, 11)):
            sd[f"{p}.blocks.{j}.0.weight"] = rng.uniform(0.05, 0.3, size=cout).astype(np.float32)
            resblock(f"{p}.blocks.{j}.1", cout, cout, k)
            sd[f"{p}.blocks.{j}.2.weight"] = rng.uniform(0.05, 0.3, size=cout).astype(np.float32)
        ch = cout
    _put_wn(sd, "output_conv", rng, (1, ch, 7), ch * 7, 0.25, 1)
    return sd


def synthetic_template(batch: int, frames: int, hop: int, seed: int = 7) -> np.ndarray:
    """A pitch-template-like signal (B, 1, frames * hop): a few harmonics of a slowly varying f0 plus a little noise."""
    rng = np.random.default_rng(seed)
    n = frames * hop
    t = np.arange(n)[None, :]
    f0 = rng.uniform(0.004, 0.02, size=(batch, 1))
    x = sum(np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28, size=(batch, 1))) / h for h in (1, 2, 3))
    return (0.3 * x + 0.02 * rng.normal(size=(batch, n)))[:, None, :].astype(np.float32)
