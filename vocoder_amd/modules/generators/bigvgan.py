"""Drop-in for ``fish_vocoder.modules.generators.bigvgan.BigVGANGenerator`` (reference bigvgan.py:255-379).

State-dict layout mirrored: flat ``resblocks.{i*num_kernels+j}`` AMPBlocks with ``convs{1,2}.{n}`` and six
``activations.{m}`` = ``Activation1d(SnakeBeta)`` each holding ``act.{alpha,beta}`` plus the alias-free filter buffers
``upsample.filter`` / ``downsample.lowpass.filter`` (names as registered by alias_free_torch==0.0.6 — third-party,
absent here; DESIGN.md lists this as unverified), and ``activation_post``.  The anti-aliased snake
(2x kaiser-sinc up -> x + sin^2(a x)/b -> 2x low-pass down) runs as one fused HIP kernel.
"""
from __future__ import annotations

import math
from math import prod
from typing import Callable

import torch
from torch import nn
from torch.nn.utils.parametrizations import weight_norm

from .. import _base
from ... import _lib
from ...engine import Engine, upsampler_config
from .hifigan import ResBlockParams, _normal_init, get_padding, noise_conv_params


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> torch.Tensor:
    """Kaiser-windowed sinc low-pass taps, normalised to unit DC gain (alias_free_torch's design rule)."""
    half = kernel_size // 2
    a = 2.285 * (half - 1) * math.pi * (4 * half_width) + 7.95
    beta = 0.1102 * (a - 8.7) if a > 50.0 else (0.5842 * (a - 21) ** 0.4 + 0.07886 * (a - 21.0) if a >= 21.0 else 0.0)
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False, dtype=torch.float64)
    if kernel_size % 2 == 0:
        t = torch.arange(-half, half, dtype=torch.float64) + 0.5
    else:
        t = torch.arange(kernel_size, dtype=torch.float64) - half
    if cutoff == 0:
        return torch.zeros(1, 1, kernel_size)
    f = 2 * cutoff * window * torch.sinc(2 * cutoff * t)
    return (f / f.sum()).to(torch.float32).view(1, 1, kernel_size)


class Snake(nn.Module):
    """Parameter holder for x + sin^2(alpha x)/alpha (reference bigvgan.py:18-71)."""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__()
        self.in_features, self.alpha_logscale = in_features, alpha_logscale
        init = torch.zeros(in_features) if alpha_logscale else torch.ones(in_features)
        self.alpha = nn.Parameter(init * alpha, requires_grad=alpha_trainable)


class SnakeBeta(Snake):
    """Parameter holder for x + sin^2(alpha x)/beta (reference bigvgan.py:74-135)."""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__(in_features, alpha, alpha_trainable, alpha_logscale)
        init = torch.zeros(in_features) if alpha_logscale else torch.ones(in_features)
        self.beta = nn.Parameter(init * alpha, requires_grad=alpha_trainable)


class _Filter(nn.Module):
    def __init__(self, taps):
        super().__init__()
        self.register_buffer("filter", taps)


class _Down(nn.Module):
    def __init__(self, taps):
        super().__init__()
        self.lowpass = _Filter(taps)


class Activation1dParams(nn.Module):
    """``act`` + ``upsample.filter`` + ``downsample.lowpass.filter`` (ratio 2, 12 taps, cutoff 0.25, half-width 0.3)."""

    def __init__(self, activation: nn.Module):
        super().__init__()
        self.act = activation
        taps = kaiser_sinc_filter1d(0.25, 0.3, 12)
        self.upsample = _Filter(taps.clone())
        self.downsample = _Down(taps.clone())


class AMPBlockParams(ResBlockParams):
    def __init__(self, channels, kernel_size, dilation, activation=SnakeBeta, snake_logscale=True):
        super().__init__(channels, kernel_size, dilation)
        self.activations = nn.ModuleList(
            Activation1dParams(activation(channels, alpha_logscale=snake_logscale)) for _ in range(2 * len(dilation)))


class BigVGANGenerator(_base.EngineModule):
    def __init__(
        self,
        *,
        hop_length: int = 512,
        upsample_rates=(8, 8, 2, 2, 2),
        upsample_kernel_sizes=(16, 16, 8, 2, 2),
        resblock_kernel_sizes=(3, 7, 11),
        resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
        num_mels: int = 128,
        upsample_initial_channel: int = 512,
        activation: Callable = SnakeBeta,
        use_template: bool = True,
        pre_conv_kernel_size: int = 7,
        post_conv_kernel_size: int = 7,
    ):
        super().__init__()
        assert prod(upsample_rates) == hop_length, f"hop_length must be {prod(upsample_rates)}"
        if activation not in (SnakeBeta, Snake):
            raise NotImplementedError("activation must be SnakeBeta (the reference default, bigvgan.py:266) or Snake")
        self.use_template = bool(use_template)
        self.num_upsamples, self.num_kernels = len(upsample_rates), len(resblock_kernel_sizes)
        self._cfg = dict(
            hop_length=hop_length, upsample_rates=list(upsample_rates),
            upsample_kernel_sizes=list(upsample_kernel_sizes), resblock_kernel_sizes=list(resblock_kernel_sizes),
            resblock_dilation_sizes=[list(d) for d in resblock_dilation_sizes], num_mels=num_mels,
            upsample_initial_channel=upsample_initial_channel, use_template=self.use_template,
            pre_conv_kernel_size=pre_conv_kernel_size, post_conv_kernel_size=post_conv_kernel_size)
        c0 = upsample_initial_channel
        self.conv_pre = weight_norm(nn.Conv1d(num_mels, c0, pre_conv_kernel_size, padding=get_padding(pre_conv_kernel_size)))
        self.noise_convs = noise_conv_params(c0, upsample_rates) if self.use_template else nn.ModuleList()
        self.ups = nn.ModuleList(
            weight_norm(nn.ConvTranspose1d(c0 >> i, c0 >> (i + 1), k, u, padding=(k - u) // 2))
            for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)))
        # the reference builds AMPBlock(ch, k, d) with no activation argument (bigvgan.py:330): the blocks are always
        # SnakeBeta(alpha_logscale=True); the generator's `activation` only selects activation_post (bigvgan.py:335-337)
        self.resblocks = nn.ModuleList(
            AMPBlockParams(c0 >> (i + 1), k, d)
            for i in range(self.num_upsamples) for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes))
        ch = c0 >> self.num_upsamples
        self.activation_post = Activation1dParams(activation(ch, alpha_logscale=True))
        self.conv_post = weight_norm(nn.Conv1d(ch, 1, post_conv_kernel_size, padding=get_padding(post_conv_kernel_size)))
        _normal_init(self.ups)
        _normal_init(self.conv_post)

    def _make_engine(self, state_dict):
        return Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**self._cfg), state_dict=state_dict, precision=self.precision)

    def forward(self, x, template=None):
        if self.use_template and template is None:
            raise TypeError("use_template=True: forward needs template (B, 1, T_mel * hop_length)")
        return self._run(x, template if self.use_template else None)
