"""Drop-in for ``fish_vocoder.modules.encoders.convnext.ConvNeXtEncoder`` (reference convnext.py:146-214).

Keys: ``downsample_layers.{i}.{0,1}``, ``stages.{i}.{j}.{dwconv,norm,pwconv1,pwconv2}.{weight,bias}``,
``stages.{i}.{j}.gamma``, ``norm.{weight,bias}``.  Eval-mode semantics only (DropPath is identity).  Engine side:
depthwise conv + LayerNorm fused in one kernel, the two pointwise Linear layers run as 1x1 convs on the fp32 MFMA
kernel with GELU / layer-scale + residual fused into their epilogues.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _base
from ... import _lib
from ...engine import Engine, convnext_config


class LayerNorm(nn.Module):
    """weight/bias holder for both data formats of the reference LayerNorm (convnext.py:47-74)."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps, self.data_format = eps, data_format


class ConvNeXtBlockParams(nn.Module):
    def __init__(self, dim, layer_scale_init_value=1e-6, mlp_ratio=4.0, kernel_size=7):
        super().__init__()
        self.dwconv = nn.Conv1d(dim, dim, kernel_size, padding=(kernel_size - 1) // 2, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, int(mlp_ratio * dim))
        self.pwconv2 = nn.Linear(int(mlp_ratio * dim), dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim)) if layer_scale_init_value > 0 else None


class ConvNeXtEncoder(_base.EngineModule):
    def __init__(self, input_channels: int = 3, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768),
                 drop_path_rate: float = 0.0, layer_scale_init_value: float = 1e-6, kernel_size: int = 7):
        super().__init__()
        assert len(depths) == len(dims)
        self._cfg = dict(input_channels=input_channels, depths=list(depths), dims=list(dims), kernel_size=kernel_size)
        self.drop_path_rate = drop_path_rate  # identity in eval (convnext.py:20-21); recorded only
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.Sequential(
            nn.Conv1d(input_channels, dims[0], kernel_size, padding=kernel_size // 2),
            LayerNorm(dims[0], eps=1e-6, data_format="channels_first")))
        for i in range(len(depths) - 1):
            self.downsample_layers.append(nn.Sequential(
                LayerNorm(dims[i], eps=1e-6, data_format="channels_first"), nn.Conv1d(dims[i], dims[i + 1], 1)))
        self.stages = nn.ModuleList(
            nn.Sequential(*[ConvNeXtBlockParams(dims[i], layer_scale_init_value, kernel_size=kernel_size)
                            for _ in range(depths[i])]) for i in range(len(depths)))
        self.norm = LayerNorm(dims[-1], eps=1e-6, data_format="channels_first")
        for m in self.modules():  # trunc_normal(0.02) / zero bias, as upstream (convnext.py:201-204)
            if isinstance(m, (nn.Conv1d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.constant_(m.bias, 0)

    def _make_engine(self, state_dict):
        return Engine(_lib.FV_MODEL_CONVNEXT, backbone=convnext_config(**self._cfg), state_dict=state_dict, precision=self.precision)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x)
