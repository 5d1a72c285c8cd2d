"""Shared plumbing of the drop-in generator modules.

A drop-in module is an ``nn.Module`` whose *parameters* are ordinary torch parameters registered under the reference's
names (so ``load_state_dict(strict=True)``, ``.to(device)``, ``.eval()``, ``state_dict()`` behave exactly like the
reference's — test.py:32-38), but whose ``forward`` hands the tensors to the HIP engine.  The parameter-holding
submodules are never called.  The engine is (re)built lazily from ``state_dict()`` whenever the weights may have
changed (load_state_dict, ``_apply``: ``.to()/.cuda()/.float()``).
"""
from __future__ import annotations

import torch
from torch import nn


class EngineModule(nn.Module):
    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_engine", None)
        object.__setattr__(self, "_precision", "f32")
        object.__setattr__(self, "_conv_algorithm", "auto")
        object.__setattr__(self, "_batch_invariant", False)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_engine())

    # -- subclasses implement -------------------------------------------------------------
    def _make_engine(self, state_dict):  # pragma: no cover - abstract
        raise NotImplementedError

    # -- arithmetic of the MFMA-bound convs: "f32" (exact fp32, the reference's; default) or "f16x3" (opt-in) ----------
    @property
    def precision(self) -> str:
        return self.__dict__.get("_precision", "f32")

    @precision.setter
    def precision(self, value: str) -> None:
        from .. import _lib
        if value not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {value!r}")
        if value != self.precision:
            self.invalidate_engine()
        object.__setattr__(self, "_precision", value)

    # -- which fp32 sums the conv layers form (include/fishvoc.h fv_conv_algo): "auto" (per launch, fastest; default), "direct", "winograd";
    #    batch_invariant = True: kernel choices from the layer shape alone, a clip's output no longer depends on the batch it is part of ----
    @property
    def conv_algorithm(self) -> str:
        return self.__dict__.get("_conv_algorithm", "auto")

    @conv_algorithm.setter
    def conv_algorithm(self, value: str) -> None:
        from .. import _lib
        if value not in _lib.CONV_ALGOS:
            raise ValueError(f"conv_algorithm must be one of {sorted(_lib.CONV_ALGOS)}, got {value!r}")
        object.__setattr__(self, "_conv_algorithm", value)
        eng = self.__dict__.get("_engine")
        if eng is not None:
            eng.set_conv_algorithm(value)

    @property
    def batch_invariant(self) -> bool:
        return bool(self.__dict__.get("_batch_invariant", False))

    @batch_invariant.setter
    def batch_invariant(self, value: bool) -> None:
        object.__setattr__(self, "_batch_invariant", bool(value))
        eng = self.__dict__.get("_engine")
        if eng is not None:
            eng.set_batch_invariant(bool(value))

    # -- engine cache ----------------------------------------------------------------------
    def invalidate_engine(self) -> None:
        eng = self.__dict__.get("_engine")
        if eng is not None:
            eng.close()
        object.__setattr__(self, "_engine", None)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_engine()
        return super()._apply(fn, *args, **kwargs)

    def engine(self, device):
        eng = self.__dict__.get("_engine")
        if eng is None or eng.device != device:
            self.invalidate_engine()
            with torch.cuda.device(device):
                eng = self._make_engine({k: v for k, v in self.state_dict().items()})
            if self.conv_algorithm != "auto":
                eng.set_conv_algorithm(self.conv_algorithm)
            if self.batch_invariant:
                eng.set_batch_invariant(True)
            object.__setattr__(self, "_engine", eng)
        return eng

    def _run(self, x: torch.Tensor, template: torch.Tensor | None = None, noise: torch.Tensor | None = None) -> torch.Tensor:
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise RuntimeError(
                f"{type(self).__name__}.forward needs a CUDA/HIP tensor (got "
                f"{getattr(x, 'device', type(x))}); vocoder_amd has no CPU path — move the model and input to the GPU")
        if self.training:
            raise RuntimeError(f"{type(self).__name__} is inference-only: call .eval() first")
        with torch.no_grad():
            if template is not None:
                template = template.to(device=x.device, dtype=torch.float32)
            if noise is not None:
                noise = noise.to(device=x.device, dtype=torch.float32).reshape(-1)
            return self.engine(x.device)(x.to(torch.float32), None, template, noise)

    def remove_parametrizations(self):
        """Kept for API compatibility (hifigan.py:251).  Weight-norm is folded inside the engine at load time, so this
        is a no-op; unlike the reference's version (which raises TypeError) it is harmless."""
        return None

    def extra_repr(self) -> str:
        return "backend=libfishvoc_hip (gfx950)"
