"""Drop-in for ``fish_vocoder.data.transforms.spectrogram.LogMelSpectrogram`` (reference spectrogram.py:59-104) —
SURVEY §8 "next" row f1: the step immediately before the generator in test.py:71, so that wav -> mel -> wav stays on
the GPU.  Same ctor kwargs and buffer names (``spectrogram.window``, ``mel_scale.fb``); ``forward(wave (B, L) or
(B, 1, L)) -> (B, n_mels, frames)`` natural-log mel with floor ln(1e-5).

Engine side (FV_MODEL_LOGMEL): reflect padding + polyphase re-layout (hop channels x n_fft/hop taps) turns the STFT into a
stride-1 conv on the fp32 MFMA kernel against a window-folded real-DFT basis; magnitude; slaney filterbank as a pointwise
conv with log(clamp) fused into its epilogue.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from ...modules import _base
from ... import _lib
from ...engine import Engine, logmel_config


def melscale_fbanks_slaney(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """Triangular slaney-scale, slaney-normalised filterbank (n_freqs, n_mels) — what torchaudio's
    ``MelScale(norm="slaney", mel_scale="slaney")`` registers as its ``fb`` buffer (third-party; restated)."""
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz_to_mel(f: float) -> float:
        return min_log_mel + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp

    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m_pts = torch.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2, dtype=torch.float64)
    f_pts = torch.where(m_pts >= min_log_mel, min_log_hz * torch.exp(logstep * (m_pts - min_log_mel)), f_sp * m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    fb = torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)
    return (fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)).to(torch.float32)


class LinearSpectrogram(_base.EngineModule):
    """Reference spectrogram.py:8-56: reflect-padded ``torch.stft(center=False)`` magnitude ``sqrt(re^2 + im^2 + 1e-6)``,
    ``forward(y (B, L) or (B, 1, L)) -> (B, n_fft/2+1, frames)``.  Same buffer name (``window``); runs on the log-mel
    engine with the filterbank stage switched off (``n_mels = 0``).  Inside LogMelSpectrogram it only holds the window."""

    def __init__(self, n_fft=2048, win_length=2048, hop_length=512, center=False, mode="pow2_sqrt"):
        super().__init__()
        if mode != "pow2_sqrt":
            raise NotImplementedError("only mode='pow2_sqrt' (the reference default) is built")
        self.n_fft, self.win_length, self.hop_length, self.center, self.mode = n_fft, win_length, hop_length, center, mode
        self.register_buffer("window", torch.hann_window(win_length))
        self._cfg = dict(sample_rate=2, n_fft=n_fft, win_length=win_length, hop_length=hop_length, n_mels=0, center=center)
        logmel_config(**self._cfg)   # validates center=False

    def _make_engine(self, state_dict):
        sd = {"spectrogram.window": v for k, v in state_dict.items() if k == "window"}
        return Engine(_lib.FV_MODEL_LOGMEL, mel=logmel_config(**self._cfg), state_dict=sd, precision=self.precision)

    def forward(self, y: torch.Tensor) -> torch.Tensor:
        if y.ndim == 2:
            y = y[:, None, :]
        return self._run(y)


class _MelScaleBuffers(nn.Module):
    def __init__(self, fb):
        super().__init__()
        self.register_buffer("fb", fb)


class LogMelSpectrogram(_base.EngineModule):
    def __init__(self, sample_rate=44100, n_fft=2048, win_length=2048, hop_length=512, n_mels=128, center=False,
                 f_min=0.0, f_max=None):
        super().__init__()
        self.sample_rate, self.n_fft, self.win_length, self.hop_length = sample_rate, n_fft, win_length, hop_length
        self.center, self.n_mels, self.f_min, self.f_max = center, n_mels, f_min, f_max or sample_rate // 2
        self._cfg = dict(sample_rate=sample_rate, n_fft=n_fft, win_length=win_length, hop_length=hop_length,
                         n_mels=n_mels, center=center, f_min=f_min, f_max=self.f_max)
        logmel_config(**self._cfg)   # validates center=False
        self.spectrogram = LinearSpectrogram(n_fft, win_length, hop_length, center)
        self.mel_scale = _MelScaleBuffers(melscale_fbanks_slaney(n_fft // 2 + 1, f_min, self.f_max, n_mels, sample_rate))

    def _make_engine(self, state_dict):
        return Engine(_lib.FV_MODEL_LOGMEL, mel=logmel_config(**self._cfg), state_dict=state_dict, precision=self.precision)

    def compress(self, x: torch.Tensor) -> torch.Tensor:
        return torch.log(torch.clamp(x, min=1e-5))

    def decompress(self, x: torch.Tensor) -> torch.Tensor:
        return torch.exp(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 2:
            x = x[:, None, :]
        return self._run(x)
