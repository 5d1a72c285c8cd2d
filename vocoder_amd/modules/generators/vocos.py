"""Drop-in for ``fish_vocoder.modules.generators.vocos.ISTFTHead`` (reference vocos.py:6-69).

Keys: ``out.{weight,bias}`` (Conv1d(dim, 2*n_fft, 1)) and the ``istft.window`` buffer.  ``forward`` returns the 2-D
``(B, T * hop_length)`` waveform like the reference (``padding="center"``: ``(B, (T - 1) * hop_length)``, torch.istft's length); ``template=`` is accepted and ignored so that
``UnifyGenerator`` can call ``head(x, template=template)`` (the reference raises TypeError there).
Engine side: only the n_fft/2+1 live rows of the 1x1 conv are computed, the inverse real DFT runs as an fp32-MFMA GEMM
against a window-folded basis, followed by a fused overlap-add / envelope kernel.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _base
from ... import _lib
from ...engine import Engine, istft_head_config


class _IstftBuffers(nn.Module):
    def __init__(self, win_length: int):
        super().__init__()
        self.register_buffer("window", torch.hann_window(win_length))


class ISTFTHead(_base.EngineModule):
    def __init__(self, dim: int, n_fft: int, hop_length: int, win_length: int, padding: str = "same"):
        super().__init__()
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self._cfg = dict(dim=dim, n_fft=n_fft, hop_length=hop_length, win_length=win_length, padding=padding)
        istft_head_config(**self._cfg)  # validates padding ("same" | "center")
        self.istft = _IstftBuffers(win_length)
        self.out = nn.Conv1d(dim, n_fft * 2, 1)

    def _make_engine(self, state_dict):
        return Engine(_lib.FV_MODEL_ISTFT_HEAD, head=istft_head_config(**self._cfg), state_dict=state_dict, precision=self.precision)

    def forward(self, x: torch.Tensor, template=None) -> torch.Tensor:
        return self._run(x)[:, 0, :]
