"""Drop-in for ``fish_vocoder.modules.generators.refinegan.RefineGANGenerator`` (reference refinegan.py:182-323).

Same keyword-only constructor and state-dict keys (``template_conv``, ``downsample_blocks.{i}.1.convs{1,2}.{n}``, ``mel_conv``,
``upsample_conv_blocks.{i}.input_conv`` / ``.blocks.{j}.{0,2}.weight`` (AdaIN) / ``.blocks.{j}.1.convs{1,2}.{n}``,
``output_conv``), same ``forward(mel, template) -> (B, 1, T * hop_length)``.

The reference's AdaIN layers add ``torch.randn_like(x) * weight`` on every call (refinegan.py:125), so its output is a random
variable.  Here the normal samples are an explicit input: ``forward(mel, template, noise=None)`` draws them with
``torch.randn`` on the input's device when ``noise`` is None (same distribution, different stream than the reference's),
or uses the given flat tensor of ``noise_elems(B, T)`` samples — which is what makes parity with the reference testable
(tests/golden/refinegan_*.npz were captured with torch.randn_like replaced by the same seeded samples).
"""
from __future__ import annotations

from math import prod

import torch
from torch import nn
from torch.nn.utils.parametrizations import weight_norm

from .. import _base
from ... import _lib
from ...engine import Engine, refinegan_config


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    return int((kernel_size * dilation - dilation) / 2)


class ResBlockParams(nn.Module):
    """Parameter container named like refinegan.ResBlock (refinegan.py:38-110): both ``convs1`` and ``convs2`` are dilated."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 7, dilation=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList(
            weight_norm(nn.Conv1d(in_channels if i == 0 else out_channels, out_channels, kernel_size, dilation=d,
                                  padding=get_padding(kernel_size, d))) for i, d in enumerate(dilation))
        self.convs2 = nn.ModuleList(
            weight_norm(nn.Conv1d(out_channels, out_channels, kernel_size, dilation=d, padding=get_padding(kernel_size, d)))
            for d in dilation)
        for m in self.modules():
            if isinstance(m, nn.Conv1d):   # refinegan.py:107-110
                m.weight = m.weight.detach().normal_(0.0, 0.01)
                m.bias.data.fill_(0.0)


class AdaINParams(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))


class ParallelResBlockParams(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_sizes=(3, 7, 11), dilation=(1, 3, 5)):
        super().__init__()
        self.input_conv = nn.Conv1d(in_channels, out_channels, kernel_size=7, stride=1, padding=3)
        self.blocks = nn.ModuleList(
            nn.Sequential(AdaINParams(out_channels), ResBlockParams(out_channels, out_channels, k, dilation),
                          AdaINParams(out_channels)) for k in kernel_sizes)


class RefineGANGenerator(_base.EngineModule):
    def __init__(
        self,
        *,
        sampling_rate: int = 44100,
        hop_length: int = 256,
        downsample_rates=(2, 2, 8, 8),
        upsample_rates=(8, 8, 2, 2),
        leaky_relu_slope: float = 0.2,
        num_mels: int = 128,
        start_channels: int = 16,
    ):
        super().__init__()
        assert prod(downsample_rates) == prod(upsample_rates) == hop_length   # refinegan.py:202
        self.sampling_rate = sampling_rate
        self.hop_length = hop_length
        self.downsample_rates = tuple(downsample_rates)
        self.upsample_rates = tuple(upsample_rates)
        self.leaky_relu_slope = leaky_relu_slope
        self._cfg = dict(sampling_rate=sampling_rate, hop_length=hop_length, downsample_rates=list(downsample_rates),
                         upsample_rates=list(upsample_rates), leaky_relu_slope=leaky_relu_slope, num_mels=num_mels,
                         start_channels=start_channels)
        ch = start_channels
        self.template_conv = weight_norm(nn.Conv1d(1, ch, kernel_size=7, stride=1, padding=3))
        self.downsample_blocks = nn.ModuleList()
        for rate in downsample_rates:
            self.downsample_blocks.append(nn.Sequential(nn.Upsample(scale_factor=1 / rate, mode="linear"),
                                                        ResBlockParams(ch, ch * 2, 7, (1, 3, 5))))
            ch *= 2
        self.mel_conv = weight_norm(nn.Conv1d(num_mels, ch, kernel_size=7, stride=1, padding=3))
        ch *= 2
        self.upsample_blocks = nn.ModuleList()
        self.upsample_conv_blocks = nn.ModuleList()
        for rate in upsample_rates:
            self.upsample_blocks.append(nn.Upsample(scale_factor=rate, mode="linear"))
            self.upsample_conv_blocks.append(ParallelResBlockParams(ch + ch // 4, ch // 2))
            ch //= 2
        self.output_conv = weight_norm(nn.Conv1d(ch, 1, kernel_size=7, stride=1, padding=3))

    def _make_engine(self, state_dict):
        return Engine(_lib.FV_MODEL_REFINEGAN, refine=refinegan_config(**self._cfg), state_dict=state_dict,
                      precision=self.precision)

    def noise_elems(self, batch: int, frames: int) -> int:
        t, ch, n = frames, self._cfg["start_channels"] * 2 ** len(self.upsample_rates) * 2, 0
        for r in self.upsample_rates:
            ch //= 2
            t *= r
            n += 6 * batch * ch * t
        return n

    def forward(self, mel: torch.Tensor, template: torch.Tensor, noise: torch.Tensor | None = None) -> torch.Tensor:
        if template is None:
            raise TypeError("RefineGANGenerator.forward(mel, template): the pitch template is required (refinegan.py:287)")
        if noise is None and isinstance(mel, torch.Tensor) and mel.is_cuda:
            noise = torch.randn(self.noise_elems(mel.shape[0], mel.shape[2]), dtype=torch.float32, device=mel.device)
        return self._run(mel, template, noise)
