"""CPU oracle for the fish_vocoder generator forward path (numpy + oracle/libfv_oracle.so).

TEST INFRASTRUCTURE ONLY — see the header of ``fv_oracle.c``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package ``vocoder_amd`` never does.

Each ``*_forward`` below restates one reference ``forward`` (paths relative to
``/root/reference``), walking a state dict that uses the *reference's* parameter
names, so a checkpoint that loads into the reference loads here too.

Pinning status:
  * HiFiGAN, ConvNeXt, Snake/SnakeBeta, ISTFTHead pre-ISTFT arithmetic: pinned against
    golden vectors produced by importing the reference in the build container
    (``tests/golden/gen_golden.py``).
  * alias_free_torch.Activation1d and vocos.spectral_ops.ISTFT are third-party packages
    absent from /root/reference: restated from their published algorithm and pinned only
    by known-answer tests — "parity unpinned" for those two pieces.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from math import prod

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfv_oracle.so")

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.c_int
_i64 = ctypes.c_int64


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc via oracle/Makefile)."""
    src = os.path.join(_HERE, "fv_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.fvo_num_threads.restype = _i
        L.fvo_set_num_threads.argtypes = [_i]
        L.fvo_weight_norm.argtypes = [_f, _f, _f, _i64, _i64]
        L.fvo_conv1d.argtypes = [_f, _f, _f, _f] + [_i] * 8
        L.fvo_conv_transpose1d.argtypes = [_f, _f, _f, _f] + [_i] * 7
        for n in ("fvo_silu", "fvo_tanh", "fvo_gelu"):
            getattr(L, n).argtypes = [_f, _f, _i64]
        L.fvo_leaky_relu.argtypes = [_f, _f, _i64, ctypes.c_float]
        L.fvo_snake.argtypes = [_f, _f, _f, _f, _i, _i, _i, _i]
        L.fvo_kaiser_sinc_filter.argtypes = [ctypes.c_double, ctypes.c_double, _i, _f]
        L.fvo_upsample_fir.argtypes = [_f, _f, _f, _i, _i, _i, _i, _i]
        L.fvo_downsample_fir.argtypes = [_f, _f, _f, _i, _i, _i, _i, _i]
        L.fvo_layernorm_cf.argtypes = [_f, _f, _f, _f, _i, _i, _i, ctypes.c_float]
        L.fvo_scale_residual.argtypes = [_f, _f, _f, _f, _i, _i, _i]
        L.fvo_istft_head_post.argtypes = [_f, _f, _f, _i, _i, _i]
        L.fvo_istft_same.argtypes = [_f, _f, _f] + [_i] * 6
        L.fvo_istft_crop.argtypes = [_f, _f, _f] + [_i] * 7
        _lib = L
        if "OMP_NUM_THREADS" not in os.environ:
            # default team size = the CPUs this process may actually use: the affinity mask capped by the container's cgroup CPU
            # quota (a 256-thread host that grants 16 CPUs of time would otherwise run every conv on 256 time-sliced threads)
            n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if q != "max":
                    n = max(1, min(n, int(float(q) / float(per) + 0.5)))
            except (OSError, ValueError):
                pass
            L.fvo_set_num_threads(min(n, 32))
    return _lib


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(_f)


def num_threads() -> int:
    return lib().fvo_num_threads()


def set_num_threads(n: int) -> None:
    lib().fvo_set_num_threads(int(n))


# ------------------------------------------------------------------------------------------------
# primitives
# ------------------------------------------------------------------------------------------------
def weight_norm(g, v) -> np.ndarray:
    """torch._weight_norm(v, g, dim=0)  (hifigan.py:31 etc.)."""
    g, v = _c(g).reshape(-1), _c(v)
    w = np.empty_like(v)
    lib().fvo_weight_norm(_p(g), _p(v), _p(w), v.shape[0], int(prod(v.shape[1:])))
    return w


def conv1d(x, w, b=None, dilation=1, padding=0, groups=1) -> np.ndarray:
    x, w = _c(x), _c(w)
    b = None if b is None else _c(b)
    B, Cin, T = x.shape
    Cout, cin_g, k = w.shape
    assert cin_g * groups == Cin
    Tout = T + 2 * padding - dilation * (k - 1)
    y = np.empty((B, Cout, Tout), np.float32)
    lib().fvo_conv1d(_p(x), _p(w), _p(b), _p(y), B, Cin, T, Cout, k, dilation, padding, groups)
    return y


def conv_transpose1d(x, w, b=None, stride=1, padding=0) -> np.ndarray:
    x, w = _c(x), _c(w)
    b = None if b is None else _c(b)
    B, Cin, Tin = x.shape
    cin2, Cout, k = w.shape
    assert cin2 == Cin
    Tout = (Tin - 1) * stride - 2 * padding + k
    y = np.empty((B, Cout, Tout), np.float32)
    lib().fvo_conv_transpose1d(_p(x), _p(w), _p(b), _p(y), B, Cin, Tin, Cout, k, stride, padding)
    return y


def _unary(name, x, *extra):
    x = _c(x)
    y = np.empty_like(x)
    getattr(lib(), name)(_p(x), _p(y), x.size, *extra)
    return y


def silu(x):
    return _unary("fvo_silu", x)


def tanh(x):
    return _unary("fvo_tanh", x)


def gelu(x):
    return _unary("fvo_gelu", x)


def leaky_relu(x, slope):
    return _unary("fvo_leaky_relu", x, ctypes.c_float(slope))


def snake(x, alpha, beta=None, logscale=False) -> np.ndarray:
    """Snake (beta=None) / SnakeBeta (bigvgan.py:60-71 / 121-135)."""
    x, alpha = _c(x), _c(alpha)
    beta = alpha if beta is None else _c(beta)
    B, C, T = x.shape
    y = np.empty_like(x)
    lib().fvo_snake(_p(x), _p(alpha), _p(beta), _p(y), B, C, T, int(bool(logscale)))
    return y


def kaiser_sinc_filter(cutoff: float, half_width: float, kernel_size: int) -> np.ndarray:
    taps = np.empty(kernel_size, np.float32)
    lib().fvo_kaiser_sinc_filter(cutoff, half_width, kernel_size, _p(taps))
    return taps


def upsample_fir(x, taps, ratio=2) -> np.ndarray:
    x, taps = _c(x), _c(taps)
    B, C, T = x.shape
    y = np.empty((B, C, T * ratio), np.float32)
    lib().fvo_upsample_fir(_p(x), _p(taps), _p(y), B, C, T, ratio, taps.size)
    return y


def downsample_fir(x, taps, ratio=2) -> np.ndarray:
    x, taps = _c(x), _c(taps)
    B, C, T = x.shape
    ks = taps.size
    pl, pr = ks // 2 - int(ks % 2 == 0), ks // 2
    Tout = (T + pl + pr - ks) // ratio + 1
    y = np.empty((B, C, Tout), np.float32)
    lib().fvo_downsample_fir(_p(x), _p(taps), _p(y), B, C, T, ratio, ks)
    return y


def activation1d(x, act, up_taps, down_taps, up_ratio=2, down_ratio=2) -> np.ndarray:
    """alias_free_torch.Activation1d: upsample -> act -> downsample (restated, parity unpinned)."""
    return downsample_fir(act(upsample_fir(x, up_taps, up_ratio)), down_taps, down_ratio)


def layernorm_cf(x, w, b, eps=1e-6) -> np.ndarray:
    x = _c(x)
    B, C, T = x.shape
    y = np.empty_like(x)
    lib().fvo_layernorm_cf(_p(x), _p(_c(w)), _p(_c(b)), _p(y), B, C, T, ctypes.c_float(eps))
    return y


def scale_residual(x, gamma, res) -> np.ndarray:
    x, res = _c(x), _c(res)
    B, C, T = x.shape
    y = np.empty_like(x)
    lib().fvo_scale_residual(_p(x), _p(None if gamma is None else _c(gamma)), _p(res), _p(y), B, C, T)
    return y


def istft_same(re, im, n_fft, hop, win) -> np.ndarray:
    """vocos.spectral_ops.ISTFT(padding='same') (restated, parity unpinned)."""
    re, im = _c(re), _c(im)
    B, NB, T = re.shape
    assert win == n_fft, "the reference head always uses win_length == n_fft"
    y = np.empty((B, T * hop), np.float32)
    lib().fvo_istft_same(_p(re), _p(im), _p(y), B, NB, T, n_fft, hop, win)
    return y


def istft_center(re, im, n_fft, hop, win) -> np.ndarray:
    """vocos.spectral_ops.ISTFT(padding='center') = torch.istft(spec, n_fft, hop, win, window, center=True) on the n_fft/2+1 live bins
    (a two-sided input of n_fft rows is cut to them before the c2r transform): the same windowed frames and envelope division as "same",
    n_fft / 2 samples trimmed from both ends -> (T - 1) * hop samples; raises like torch when the envelope has a zero there."""
    re, im = _c(re), _c(im)
    B, NB, T = re.shape
    assert win == n_fft, "the reference head always uses win_length == n_fft"
    if T < 2:
        raise RuntimeError("istft(center=True): a single frame leaves no samples")
    w2 = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win) / win)).astype(np.float32).astype(np.float64) ** 2
    env = np.zeros((T - 1) * hop + win)
    for t in range(T):
        env[t * hop:t * hop + win] += w2
    if env[n_fft // 2:n_fft // 2 + (T - 1) * hop].min() < 1e-11:
        raise RuntimeError("window overlap add min: 1")   # torch.istft's check
    y = np.empty((B, (T - 1) * hop), np.float32)
    lib().fvo_istft_crop(_p(re), _p(im), _p(y), B, NB, T, n_fft, hop, win, n_fft // 2)
    return y


# ------------------------------------------------------------------------------------------------
# state-dict helpers (reference key names)
# ------------------------------------------------------------------------------------------------
def _get_padding(k, d=1):  # hifigan.py:21-22
    return (k * d - d) // 2


def folded_weight(sd, prefix) -> np.ndarray:
    """Weight of a (possibly weight-normed) conv: parametrizations.weight.original{0,1} or plain .weight."""
    k0 = f"{prefix}.parametrizations.weight.original0"
    if k0 in sd:
        return weight_norm(sd[k0], sd[f"{prefix}.parametrizations.weight.original1"])
    if f"{prefix}.weight_g" in sd:  # legacy torch.nn.utils.weight_norm naming
        return weight_norm(sd[f"{prefix}.weight_g"], sd[f"{prefix}.weight_v"])
    return _c(sd[f"{prefix}.weight"])


def _bias(sd, prefix):
    return _c(sd[f"{prefix}.bias"]) if f"{prefix}.bias" in sd else None


def strip_prefix(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


# ------------------------------------------------------------------------------------------------
# HiFiGAN (fish_vocoder/modules/generators/hifigan.py)
# ------------------------------------------------------------------------------------------------
def resblock1_forward(sd, prefix, x, k, dilations):
    """ResBlock1.forward (hifigan.py:101-108)."""
    for n, d in enumerate(dilations):
        xt = silu(x)
        xt = conv1d(xt, folded_weight(sd, f"{prefix}.convs1.{n}"), _bias(sd, f"{prefix}.convs1.{n}"),
                    dilation=d, padding=_get_padding(k, d))
        xt = silu(xt)
        xt = conv1d(xt, folded_weight(sd, f"{prefix}.convs2.{n}"), _bias(sd, f"{prefix}.convs2.{n}"),
                    dilation=1, padding=_get_padding(k, 1))
        x = xt + x
    return x


def noise_conv(sd, prefix, template, stride) -> np.ndarray:
    """noise_convs[i](template): Conv1d(1, C, k=2*stride, stride=stride, padding=stride//2), or Conv1d(1, C, 1) for the
    last stage (hifigan.py:192-204).  template: (B, 1, T_audio)."""
    w = _c(sd[f"{prefix}.weight"])            # (C, 1, k)
    b = _c(sd[f"{prefix}.bias"])
    k = w.shape[2]
    pad = stride // 2 if k > 1 else 0
    tp = np.pad(_c(template)[:, 0].astype(np.float64), ((0, 0), (pad, pad)))
    win = np.lib.stride_tricks.sliding_window_view(tp, k, axis=1)[:, ::stride]      # (B, T_out, k)
    return (np.einsum("btk,ck->bct", win, w[:, 0].astype(np.float64)) + b[None, :, None]).astype(np.float32)


def hifigan_forward(sd, cfg, mel, collect=None, template=None) -> np.ndarray:
    """HiFiGANGenerator.forward with use_template=False (hifigan.py:226-249).

    cfg keys = the reference ctor kwargs (hifigan.py:137-151).  `collect`, if a dict, receives
    per-stage activations (for layer-level parity tests).
    """
    rates = list(cfg["upsample_rates"])
    uks = list(cfg["upsample_kernel_sizes"])
    rks = list(cfg["resblock_kernel_sizes"])
    rds = [list(d) for d in cfg["resblock_dilation_sizes"]]
    assert prod(rates) == cfg["hop_length"], f"hop_length must be {prod(rates)}"  # hifigan.py:154-156
    use_template = bool(cfg.get("use_template", False))
    if use_template and template is None:
        raise TypeError("use_template=True needs a template (B, 1, T_mel * hop_length)")
    pk, qk = cfg.get("pre_conv_kernel_size", 7), cfg.get("post_conv_kernel_size", 7)

    x = conv1d(mel, folded_weight(sd, "conv_pre"), _bias(sd, "conv_pre"), padding=_get_padding(pk))
    if collect is not None:
        collect["conv_pre"] = x
    for i, (u, k) in enumerate(zip(rates, uks)):
        x = silu(x)                                                           # hifigan.py:230
        x = conv_transpose1d(x, folded_weight(sd, f"ups.{i}"), _bias(sd, f"ups.{i}"),
                             stride=u, padding=(k - u) // 2)                  # hifigan.py:231
        if use_template:                                                      # hifigan.py:233-234
            x = x + noise_conv(sd, f"noise_convs.{i}", template, int(prod(rates[i + 1:])))
        if collect is not None:
            collect[f"ups.{i}"] = x
        # ParralelBlock: stack(...).mean(0)  (hifigan.py:132-133)
        outs = [resblock1_forward(sd, f"resblocks.{i}.blocks.{j}", x, rk, rd)
                for j, (rk, rd) in enumerate(zip(rks, rds))]
        x = np.mean(np.stack(outs, 0), axis=0, dtype=np.float32)
        if collect is not None:
            collect[f"resblocks.{i}"] = x
    x = _post_activation(cfg.get("post_activation", "silu"))(x)               # post_activation() (hifigan.py:150,213,245)
    x = conv1d(x, folded_weight(sd, "conv_post"), _bias(sd, "conv_post"), padding=_get_padding(qk))
    return tanh(x)


def _post_activation(spec):
    """cfg["post_activation"]: "silu" (the reference default, partial(nn.SiLU, inplace=True)), ("leaky_relu", slope), "relu", "gelu",
    "tanh" or "identity" — the element-wise nn.Modules the drop-in accepts for `post_activation` (hifigan.py:150)."""
    name, arg = (spec, None) if isinstance(spec, str) else (spec[0], spec[1])
    return {"silu": silu, "leaky_relu": lambda x: leaky_relu(x, arg), "relu": lambda x: leaky_relu(x, 0.0), "gelu": gelu, "tanh": tanh,
            "identity": lambda x: x}[name]


# ------------------------------------------------------------------------------------------------
# BigVGAN (fish_vocoder/modules/generators/bigvgan.py)
# ------------------------------------------------------------------------------------------------
def _aa_filters(sd, prefix):
    """Activation1d filter taps: use the checkpoint's buffers when present, else the default
    kaiser-sinc design (ratio 2, 12 taps: cutoff 0.25, half-width 0.3)."""
    ku, kd = f"{prefix}.upsample.filter", f"{prefix}.downsample.lowpass.filter"
    default = kaiser_sinc_filter(0.25, 0.3, 12)
    up = _c(sd[ku]).reshape(-1) if ku in sd else default
    dn = _c(sd[kd]).reshape(-1) if kd in sd else default
    return up, dn


def _aa_snakebeta(sd, prefix, x, logscale=True):
    """Activation1d(SnakeBeta(C, alpha_logscale=True)) (bigvgan.py:226-233,335-337)."""
    up, dn = _aa_filters(sd, prefix)
    a, b = sd[f"{prefix}.act.alpha"], sd.get(f"{prefix}.act.beta")
    return activation1d(x, lambda z: snake(z, a, b, logscale), up, dn)


def ampblock_forward(sd, prefix, x, k, dilations):
    """AMPBlock.forward (bigvgan.py:235-245): acts1 = activations[::2], acts2 = activations[1::2]."""
    for n, d in enumerate(dilations):
        xt = _aa_snakebeta(sd, f"{prefix}.activations.{2 * n}", x)
        xt = conv1d(xt, folded_weight(sd, f"{prefix}.convs1.{n}"), _bias(sd, f"{prefix}.convs1.{n}"),
                    dilation=d, padding=_get_padding(k, d))
        xt = _aa_snakebeta(sd, f"{prefix}.activations.{2 * n + 1}", xt)
        xt = conv1d(xt, folded_weight(sd, f"{prefix}.convs2.{n}"), _bias(sd, f"{prefix}.convs2.{n}"),
                    dilation=1, padding=_get_padding(k, 1))
        x = xt + x
    return x


def bigvgan_forward(sd, cfg, mel, collect=None, template=None) -> np.ndarray:
    """BigVGANGenerator.forward (bigvgan.py:352-371); use_template=True (the ctor default, bigvgan.py:267) adds
    noise_convs[i](template) after every upsampler (bigvgan.py:300-330,359-360)."""
    rates = list(cfg["upsample_rates"])
    uks = list(cfg["upsample_kernel_sizes"])
    rks = list(cfg["resblock_kernel_sizes"])
    rds = [list(d) for d in cfg["resblock_dilation_sizes"]]
    assert prod(rates) == cfg["hop_length"], f"hop_length must be {prod(rates)}"
    use_template = bool(cfg.get("use_template", False))
    if use_template and template is None:
        raise TypeError("use_template=True needs a template (B, 1, T_mel * hop_length)")
    pk, qk = cfg.get("pre_conv_kernel_size", 7), cfg.get("post_conv_kernel_size", 7)
    nk = len(rks)

    x = conv1d(mel, folded_weight(sd, "conv_pre"), _bias(sd, "conv_pre"), padding=_get_padding(pk))
    for i, (u, k) in enumerate(zip(rates, uks)):
        x = conv_transpose1d(x, folded_weight(sd, f"ups.{i}"), _bias(sd, f"ups.{i}"),
                             stride=u, padding=(k - u) // 2)                  # no pre-activation (bigvgan.py:355-356)
        if use_template:                                                      # bigvgan.py:359-360
            x = x + noise_conv(sd, f"noise_convs.{i}", template, int(prod(rates[i + 1:])))
        if collect is not None:
            collect[f"ups.{i}"] = x
        outs = [ampblock_forward(sd, f"resblocks.{i * nk + j}", x, rk, rd)
                for j, (rk, rd) in enumerate(zip(rks, rds))]
        x = np.mean(np.stack(outs, 0), axis=0, dtype=np.float32)            # bigvgan.py:361-365
        if collect is not None:
            collect[f"stage.{i}"] = x
    x = _aa_snakebeta(sd, "activation_post", x)                              # bigvgan.py:367
    x = conv1d(x, folded_weight(sd, "conv_post"), _bias(sd, "conv_post"), padding=_get_padding(qk))
    return tanh(x)


# ------------------------------------------------------------------------------------------------
# ConvNeXt encoder + Vocos ISTFT head + UnifyGenerator
# ------------------------------------------------------------------------------------------------
def convnext_block_forward(sd, prefix, x, kernel_size=7):
    """ConvNeXtBlock.forward (convnext.py:123-143); Linear layers applied as 1x1 convs on (B, C, T)."""
    C = x.shape[1]
    h = conv1d(x, sd[f"{prefix}.dwconv.weight"], sd[f"{prefix}.dwconv.bias"],
               padding=(kernel_size - 1) // 2, groups=C)
    h = layernorm_cf(h, sd[f"{prefix}.norm.weight"], sd[f"{prefix}.norm.bias"], 1e-6)
    h = conv1d(h, _c(sd[f"{prefix}.pwconv1.weight"])[:, :, None], sd[f"{prefix}.pwconv1.bias"])
    h = gelu(h)
    h = conv1d(h, _c(sd[f"{prefix}.pwconv2.weight"])[:, :, None], sd[f"{prefix}.pwconv2.bias"])
    return scale_residual(h, sd.get(f"{prefix}.gamma"), x)


def convnext_forward(sd, cfg, x) -> np.ndarray:
    """ConvNeXtEncoder.forward (convnext.py:206-214)."""
    depths, dims = list(cfg["depths"]), list(cfg["dims"])
    ks = cfg.get("kernel_size", 7)
    for i in range(len(depths)):
        if i == 0:   # stem: Conv1d(k) + LN_cf  (convnext.py:164-174)
            x = conv1d(x, sd["downsample_layers.0.0.weight"], sd["downsample_layers.0.0.bias"], padding=ks // 2)
            x = layernorm_cf(x, sd["downsample_layers.0.1.weight"], sd["downsample_layers.0.1.bias"], 1e-6)
        else:        # LN_cf + 1x1 conv (convnext.py:177-182)
            x = layernorm_cf(x, sd[f"downsample_layers.{i}.0.weight"], sd[f"downsample_layers.{i}.0.bias"], 1e-6)
            x = conv1d(x, sd[f"downsample_layers.{i}.1.weight"], sd[f"downsample_layers.{i}.1.bias"])
        for j in range(depths[i]):
            x = convnext_block_forward(sd, f"stages.{i}.{j}", x, ks)
    return layernorm_cf(x, sd["norm.weight"], sd["norm.bias"], 1e-6)


def istft_head_pre(sd, x):
    """ISTFTHead.forward up to (but excluding) self.istft (vocos.py:55-67): returns (re, im), each (B, n_fft, T)."""
    h = conv1d(x, sd["out.weight"], sd["out.bias"])
    B, C2, T = h.shape
    n_fft = C2 // 2
    re = np.empty((B, n_fft, T), np.float32)
    im = np.empty((B, n_fft, T), np.float32)
    lib().fvo_istft_head_post(_p(h), _p(re), _p(im), B, n_fft, T)
    return re, im


def istft_head_forward(sd, cfg, x) -> np.ndarray:
    """ISTFTHead.forward (vocos.py:43-69) -> (B, T*hop) (padding="same") / (B, (T-1)*hop) (padding="center")."""
    re, im = istft_head_pre(sd, x)
    istft = istft_center if cfg.get("padding", "same") == "center" else istft_same
    return istft(re, im, cfg["n_fft"], cfg["hop_length"], cfg["win_length"])


def vocos_forward(sd, cfg, mel) -> np.ndarray:
    """UnifyGenerator(ConvNeXtEncoder, ISTFTHead).forward, intended semantics (unify.py:18-33):
    head(backbone(x))[:, None, :]."""
    h = convnext_forward(strip_prefix(sd, "backbone."), cfg["backbone"], mel)
    y = istft_head_forward(strip_prefix(sd, "head."), cfg["head"], h)
    return y[:, None, :]


def firefly_forward(sd, cfg, mel) -> np.ndarray:
    """UnifyGenerator(ConvNeXtEncoder, HiFiGANGenerator) (configs/model/generator/firefly-gan-base.yaml)."""
    h = convnext_forward(strip_prefix(sd, "backbone."), cfg["backbone"], mel)
    return hifigan_forward(strip_prefix(sd, "head."), cfg["head"], h)


# ------------------------------------------------------------------------------------------------
# RefineGAN (fish_vocoder/modules/generators/refinegan.py)
# ------------------------------------------------------------------------------------------------
def linear_interp(x, scale_factor) -> np.ndarray:
    """nn.Upsample(scale_factor=s, mode="linear") = F.interpolate(..., align_corners=False) (refinegan.py:229,262):
    L_out = floor(L_in * s); src = max((dst + 0.5) / s - 0.5, 0); i0 = floor(src), i1 = min(i0 + 1, L_in - 1);
    y = x[i0] * (1 - frac) + x[i1] * frac  (aten upsample_linear1d with the given scale factor, fp32 coordinates)."""
    x = _c(x)
    lin = x.shape[-1]
    lout = int(np.floor(lin * float(scale_factor)))
    scale = np.float32(1.0 / float(scale_factor))
    src = (np.arange(lout, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
    src = np.maximum(src, np.float32(0.0))
    i0 = np.minimum(np.floor(src).astype(np.int64), lin - 1)
    i1 = np.minimum(i0 + 1, lin - 1)
    lam1 = (src - i0.astype(np.float32)).astype(np.float32)
    lam0 = np.float32(1.0) - lam1
    return (x[..., i0] * lam0 + x[..., i1] * lam1).astype(np.float32)


def refinegan_resblock_forward(sd, prefix, x, k, cin, cout, slope, dilations=(1, 3, 5)):
    """refinegan.ResBlock.forward (refinegan.py:87-100): BOTH convs of a pair are dilated; no residual for the first pair
    when the channel count changes."""
    for n, d in enumerate(dilations):
        xt = leaky_relu(x, slope)
        xt = conv1d(xt, folded_weight(sd, f"{prefix}.convs1.{n}"), _bias(sd, f"{prefix}.convs1.{n}"), dilation=d,
                    padding=_get_padding(k, d))
        xt = leaky_relu(xt, slope)
        xt = conv1d(xt, folded_weight(sd, f"{prefix}.convs2.{n}"), _bias(sd, f"{prefix}.convs2.{n}"), dilation=d,
                    padding=_get_padding(k, d))
        x = xt + x if (n != 0 or cin == cout) else xt
    return x


def refinegan_forward(sd, cfg, mel, template, noise) -> np.ndarray:
    """RefineGANGenerator.forward(mel, template) (refinegan.py:287-323) with AdaIN's torch.randn_like (refinegan.py:125)
    replaced by the given standard-normal tensors `noise` (list, in order of use: stage, branch, layer)."""
    slope = float(cfg.get("leaky_relu_slope", 0.2))
    downs_r, ups_r = list(cfg["downsample_rates"]), list(cfg["upsample_rates"])
    assert prod(downs_r) == prod(ups_r) == cfg["hop_length"]               # refinegan.py:202
    ch = cfg["start_channels"]
    x = conv1d(template, folded_weight(sd, "template_conv"), _bias(sd, "template_conv"), padding=3)
    downs = []
    for i, r in enumerate(downs_r):
        x = leaky_relu(x, slope)                                               # in place: the skip is the activated tensor
        downs.append(x)
        x = linear_interp(x, 1.0 / r)
        x = refinegan_resblock_forward(sd, f"downsample_blocks.{i}.1", x, 7, ch, ch * 2, slope)
        ch *= 2
    m = conv1d(mel, folded_weight(sd, "mel_conv"), _bias(sd, "mel_conv"), padding=3)
    x = np.concatenate([x, m], axis=1)
    ch *= 2
    it = iter(noise)

    def adain(x, w):                                                           # refinegan.py:124-127
        # AdaIN(channels=...) is built without the generator's slope (refinegan.py:157,165): always LeakyReLU(0.2)
        return leaky_relu(x + next(it) * w[None, :, None], 0.2)

    for i, (r, down) in enumerate(zip(ups_r, reversed(downs))):
        x = leaky_relu(x, slope)
        x = linear_interp(x, r)
        x = np.concatenate([x, down], axis=1)
        p = f"upsample_conv_blocks.{i}"
        cout = ch // 2
        x = conv1d(x, _c(sd[f"{p}.input_conv.weight"]), _c(sd[f"{p}.input_conv.bias"]), padding=3)
        outs = []
        for j, k in enumerate((3, 7, 11)):
            y = adain(x, _c(sd[f"{p}.blocks.{j}.0.weight"]))
            y = refinegan_resblock_forward(sd, f"{p}.blocks.{j}.1", y, k, cout, cout, slope)
            outs.append(adain(y, _c(sd[f"{p}.blocks.{j}.2.weight"])))
        x = np.mean(np.stack(outs, 0), axis=0, dtype=np.float32)
        ch = cout
    x = leaky_relu(x, slope)
    x = conv1d(x, folded_weight(sd, "output_conv"), _bias(sd, "output_conv"), padding=3)
    return tanh(x)


# ------------------------------------------------------------------------------------------------
# Log-mel front-end (next row f1): fish_vocoder/data/transforms/spectrogram.py
# ------------------------------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    """torchaudio.functional._hz_to_mel(mel_scale="slaney") — third-party (torchaudio, absent here), restated from the
    published Slaney / librosa formula: linear below 1 kHz (200/3 Hz per mel), log above."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def melscale_fbanks_slaney(n_freqs, f_min, f_max, n_mels, sample_rate) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(norm="slaney", mel_scale="slaney") -> (n_freqs, n_mels) fp32.
    PARITY UNPINNED (torchaudio is not installed and not vendored in /root/reference; call site spectrogram.py:83-91)."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2)
    f_pts = _mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
    return (fb * enorm[None, :]).astype(np.float32)


def linear_spectrogram(wave, n_fft, win_length, hop_length) -> np.ndarray:
    """LinearSpectrogram.forward, mode pow2_sqrt, center=False (spectrogram.py:25-56): reflect-pad
    ((win-hop)//2, (win-hop+1)//2) -> STFT(hann periodic) -> sqrt(re^2 + im^2 + 1e-6).  (B, L) -> (B, n_fft/2+1, frames)."""
    y = np.asarray(wave, dtype=np.float32)
    if y.ndim == 3:
        y = y[:, 0]
    pl, pr = (win_length - hop_length) // 2, (win_length - hop_length + 1) // 2
    yp = np.pad(y.astype(np.float64), ((0, 0), (pl, pr)), mode="reflect")
    n = np.arange(win_length, dtype=np.float64)
    window = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)).astype(np.float32).astype(np.float64)
    if win_length < n_fft:  # torch.stft centres a short window inside n_fft
        lpad = (n_fft - win_length) // 2
        window = np.pad(window, (lpad, n_fft - win_length - lpad))
    frames = 1 + (yp.shape[1] - n_fft) // hop_length
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(frames)[:, None]
    seg = yp[:, idx] * window[None, None, :]                 # (B, frames, n_fft)
    spec = np.fft.rfft(seg, axis=-1)                          # (B, frames, n_fft/2+1)
    mag = np.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-6)
    return np.ascontiguousarray(mag.transpose(0, 2, 1)).astype(np.float32)


def logmel_forward(wave, cfg) -> np.ndarray:
    """LogMelSpectrogram.forward (spectrogram.py:99-104): mel_scale(spectrogram(x)) -> log(clamp(., 1e-5))."""
    sr = cfg["sample_rate"]
    f_max = cfg.get("f_max") or sr // 2
    mag = linear_spectrogram(wave, cfg["n_fft"], cfg["win_length"], cfg["hop_length"])
    fb = melscale_fbanks_slaney(cfg["n_fft"] // 2 + 1, cfg.get("f_min", 0.0), f_max, cfg["n_mels"], sr)
    mel = np.einsum("bft,fm->bmt", mag.astype(np.float64), fb.astype(np.float64))
    return np.log(np.maximum(mel, 1e-5)).astype(np.float32)
