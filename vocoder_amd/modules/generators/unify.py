"""Drop-in for ``fish_vocoder.modules.generators.unify.UnifyGenerator`` (reference unify.py:5-60):
``head(backbone(x), template=template)``, channel dim added for 2-D heads.  ``vq`` is out of scope (no shipped
config wires one) and must be None."""
from __future__ import annotations

import torch
from torch import nn


class UnifyGenerator(nn.Module):
    def __init__(self, backbone: nn.Module, head: nn.Module, vq: nn.Module | None = None):
        super().__init__()
        if vq is not None:
            raise NotImplementedError("UnifyGenerator(vq=...) is out of scope: no shipped generator config sets vq")
        self.backbone, self.head, self.vq = backbone, head, None

    def forward(self, x: torch.Tensor, template=None) -> torch.Tensor:
        x = self.head(self.backbone(x), template=template)
        return x[:, None, :] if x.ndim == 2 else x

    def encode(self, x):
        raise ValueError("VQ module is not present in the model.")  # unify.py:36-37

    def decode(self, codes, template=None):
        raise ValueError("VQ module is not present in the model.")  # unify.py:44-45

    def remove_parametrizations(self):
        for m in (self.backbone, self.head):
            if hasattr(m, "remove_parametrizations"):
                m.remove_parametrizations()
