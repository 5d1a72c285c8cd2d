"""Drop-in for ``fish_vocoder.modules.generators.hifigan.HiFiGANGenerator`` (reference hifigan.py:136-257).

Same keyword-only constructor, same state-dict keys
(``conv_pre|ups.{i}|resblocks.{i}.blocks.{j}.convs{1,2}.{n}|conv_post`` +
``.parametrizations.weight.original{0,1}`` / ``.bias``), same ``forward(x, template=None)``:
``(B, num_mels, T_mel) fp32 -> (B, 1, T_mel * hop_length) fp32`` in (-1, 1).  The forward itself is
one ``fv_forward`` call into the HIP engine (fused SiLU -> conv -> residual MFMA kernels).
"""
from __future__ import annotations

from functools import partial
from math import prod
from typing import Callable

from torch import nn
from torch.nn.utils.parametrizations import weight_norm

from .. import _base
from ... import _lib
from ...engine import Engine, post_activation_to_act, upsampler_config


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    """'same' padding of an odd, dilated kernel (reference hifigan.py:21-22)."""
    return dilation * (kernel_size - 1) // 2


def _normal_init(module: nn.Module, std: float = 0.01) -> None:
    for m in module.modules():
        if isinstance(m, (nn.Conv1d, nn.ConvTranspose1d)):
            # through the weight-norm parametrization, like `m.weight.data.normal_` upstream (hifigan.py:15-18)
            m.weight = m.weight.detach().normal_(0.0, std)


def noise_conv_params(c0: int, upsample_rates) -> nn.ModuleList:
    """``noise_convs.{i}``: strided Conv1d(1 -> C_i) applied to the pitch template after up-sampling stage i
    (reference hifigan.py:192-204); plain convs, no weight-norm."""
    convs = nn.ModuleList()
    n = len(upsample_rates)
    for i in range(n):
        ch = c0 >> (i + 1)
        if i + 1 < n:
            s = int(prod(upsample_rates[i + 1:]))
            convs.append(nn.Conv1d(1, ch, kernel_size=2 * s, stride=s, padding=s // 2))
        else:
            convs.append(nn.Conv1d(1, ch, kernel_size=1))
    return convs


class ResBlockParams(nn.Module):
    """Parameter container named like ResBlock1 / AMPBlock: ``convs1.{n}`` (dilated) and ``convs2.{n}``."""

    def __init__(self, channels: int, kernel_size: int, dilation):
        super().__init__()
        if len(dilation) != 3:
            raise IndexError("ResBlock1 uses exactly three dilations (reference hifigan.py:38,47,56)")
        self.convs1 = nn.ModuleList(
            weight_norm(nn.Conv1d(channels, channels, kernel_size, dilation=d, padding=get_padding(kernel_size, d)))
            for d in dilation)
        self.convs2 = nn.ModuleList(
            weight_norm(nn.Conv1d(channels, channels, kernel_size, padding=get_padding(kernel_size)))
            for _ in dilation)
        _normal_init(self)


class ParallelBlockParams(nn.Module):
    """``blocks.{j}`` — one ResBlockParams per resblock kernel size (reference ParralelBlock, hifigan.py:117-130)."""

    def __init__(self, channels, kernel_sizes, dilation_sizes):
        super().__init__()
        assert len(kernel_sizes) == len(dilation_sizes)
        self.blocks = nn.ModuleList(ResBlockParams(channels, k, d) for k, d in zip(kernel_sizes, dilation_sizes))


class HiFiGANGenerator(_base.EngineModule):
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
        use_template: bool = True,
        pre_conv_kernel_size: int = 7,
        post_conv_kernel_size: int = 7,
        post_activation: Callable = partial(nn.SiLU, inplace=True),
    ):
        super().__init__()
        assert prod(upsample_rates) == hop_length, f"hop_length must be {prod(upsample_rates)}"
        act = post_activation()
        fv_act, slope = post_activation_to_act(act)   # raises for modules without a kernel form
        self.activation_post = act  # no parameters; kept for repr/state parity
        self.use_template = bool(use_template)
        self.num_upsamples = len(upsample_rates)
        self.num_kernels = len(resblock_kernel_sizes)
        self._cfg = dict(
            hop_length=hop_length, upsample_rates=list(upsample_rates),
            upsample_kernel_sizes=list(upsample_kernel_sizes), resblock_kernel_sizes=list(resblock_kernel_sizes),
            resblock_dilation_sizes=[list(d) for d in resblock_dilation_sizes], num_mels=num_mels,
            upsample_initial_channel=upsample_initial_channel, use_template=self.use_template,
            pre_conv_kernel_size=pre_conv_kernel_size, post_conv_kernel_size=post_conv_kernel_size,
            post_activation=fv_act, post_activation_slope=slope)

        c0 = upsample_initial_channel
        self.conv_pre = weight_norm(nn.Conv1d(num_mels, c0, pre_conv_kernel_size, padding=get_padding(pre_conv_kernel_size)))
        self.noise_convs = noise_conv_params(c0, upsample_rates) if self.use_template else nn.ModuleList()
        self.ups = nn.ModuleList(
            weight_norm(nn.ConvTranspose1d(c0 >> i, c0 >> (i + 1), k, u, padding=(k - u) // 2))
            for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)))
        self.resblocks = nn.ModuleList(
            ParallelBlockParams(c0 >> (i + 1), resblock_kernel_sizes, resblock_dilation_sizes)
            for i in range(self.num_upsamples))
        self.conv_post = weight_norm(nn.Conv1d(c0 >> self.num_upsamples, 1, post_conv_kernel_size,
                                               padding=get_padding(post_conv_kernel_size)))
        _normal_init(self.ups)
        _normal_init(self.conv_post)

    def _make_engine(self, state_dict):
        return Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**self._cfg), state_dict=state_dict, precision=self.precision)

    def forward(self, x, template=None):
        if self.use_template and template is None:
            raise TypeError("use_template=True: forward needs template (B, 1, T_mel * hop_length)")
        return self._run(x, template if self.use_template else None)   # like upstream, an unused template is ignored
