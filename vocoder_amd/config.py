"""A small Hydra-compatible config loader for the generator path.

The reference builds its generator with ``hydra.utils.instantiate(cfg.model)`` from
``fish_vocoder/configs/model/{gan.yaml, generator/*.yaml, resolution/*.yaml}`` (test.py:25-31).  Hydra/OmegaConf are
not installed here, and only three of their features are on the path: the ``defaults`` composition of
``model/gan.yaml:1-6`` (``resolution@_here_`` + ``generator``), ``${a.b}`` / ``${eval:...}`` interpolation
(test.py:19) and recursive ``_target_`` instantiation with keyword arguments.  This module implements exactly those,
with the same key names, so ``configs/model/generator/*.yaml`` files written for the reference load unchanged:
a ``_target_`` under ``fish_vocoder.modules.{generators,encoders}`` resolves to the drop-in class of the same name
under ``vocoder_amd.modules``.
"""
from __future__ import annotations

import importlib
import os
import re
from typing import Any

import yaml

CONFIG_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")

# reference module path -> drop-in module path (class names are identical)
TARGET_ALIASES = {
    "fish_vocoder.modules.generators.hifigan": "vocoder_amd.modules.generators.hifigan",
    "fish_vocoder.modules.generators.bigvgan": "vocoder_amd.modules.generators.bigvgan",
    "fish_vocoder.modules.generators.refinegan": "vocoder_amd.modules.generators.refinegan",
    "fish_vocoder.modules.generators.vocos": "vocoder_amd.modules.generators.vocos",
    "fish_vocoder.modules.generators.unify": "vocoder_amd.modules.generators.unify",
    "fish_vocoder.modules.encoders.convnext": "vocoder_amd.modules.encoders.convnext",
    "fish_vocoder.data.transforms.spectrogram": "vocoder_amd.data.transforms.spectrogram",
}

_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _lookup(root: dict, dotted: str) -> Any:
    cur: Any = root
    for part in dotted.strip().split("."):
        if isinstance(cur, dict) and part in cur:
            cur = cur[part]
        elif isinstance(cur, list) and part.isdigit():
            cur = cur[int(part)]
        else:
            raise KeyError(f"interpolation key '{dotted}' not found")
    return cur


def _resolve_str(s: str, root: dict, depth: int = 0) -> Any:
    if depth > 16:
        raise RecursionError(f"interpolation too deep in '{s}'")
    m = _INTERP.fullmatch(s.strip())
    if m and not m.group(1).startswith("eval:"):
        # a lone ${a.b}: keep the referenced value's type (int, list, ...)
        return _resolve(_lookup(root, m.group(1)), root, depth + 1)

    def sub(mo):
        body = mo.group(1)
        if body.startswith("eval:"):
            return mo.group(0)  # inner plain keys first
        return str(_resolve(_lookup(root, body), root, depth + 1))

    prev = None
    while prev != s:
        prev = s
        s = _INTERP.sub(sub, s)
    m = _INTERP.fullmatch(s.strip())
    if m and m.group(1).startswith("eval:"):
        expr = m.group(1)[5:].strip()
        if len(expr) >= 2 and expr[0] == expr[-1] and expr[0] in "'\"":
            expr = expr[1:-1]
        # the reference registers Python's eval as the resolver (test.py:19); arithmetic only, no builtins
        return eval(expr, {"__builtins__": {}}, {})
    return s


def _resolve(node: Any, root: dict, depth: int = 0) -> Any:
    if isinstance(node, dict):
        return {k: _resolve(v, root, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    if isinstance(node, str) and "${" in node:
        return _resolve_str(node, root, depth)
    return node


def load_yaml(path: str) -> dict:
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _deep_merge(base: dict, over: dict) -> dict:
    out = dict(base)
    for k, v in over.items():
        out[k] = _deep_merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def _load_generator(root: str, name: str, _depth: int = 0) -> dict:
    """``model/generator/<name>.yaml`` with Hydra's in-group ``defaults`` list (vocos-huge.yaml: ``[vocos, _self_]``): the
    named siblings are merged in order, ``_self_`` marks where this file's own keys go (last when absent)."""
    if _depth > 8:
        raise RecursionError(f"generator config '{name}': defaults chain too deep")
    node = dict(load_yaml(os.path.join(root, "model", "generator", f"{name}.yaml")))
    defaults = node.pop("defaults", None)
    if not defaults:
        return node
    merged: dict = {}
    seen_self = False
    for d in defaults:
        if d == "_self_":
            merged = _deep_merge(merged, node)
            seen_self = True
        elif isinstance(d, str):
            merged = _deep_merge(merged, _load_generator(root, d, _depth + 1))
        else:
            raise ValueError(f"generator config '{name}': unsupported defaults entry {d!r}")
    return merged if seen_self else _deep_merge(merged, node)


def compose_model(generator: str = "hifigan", resolution: str = "44100_512_2048", overrides: dict | None = None,
                  config_root: str | None = None) -> dict:
    """Equivalent of ``model/gan.yaml``'s defaults for inference: resolution keys merged at ``model`` level
    (``resolution@_here_``), ``model.generator`` from ``model/generator/<generator>.yaml``; ``overrides`` are
    dotted keys relative to ``model`` (e.g. ``{"num_mels": 80}`` == CLI ``model.num_mels=80``).
    Returns the fully interpolated ``{"model": {...}}`` dict."""
    root = config_root or CONFIG_ROOT
    model = dict(load_yaml(os.path.join(root, "model", "resolution", f"{resolution}.yaml")))
    model["generator"] = _load_generator(root, generator)
    for key, val in (overrides or {}).items():
        cur = model
        parts = key.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = val
    cfg = {"model": model}
    return _resolve(cfg, cfg)


def locate(target: str):
    mod, _, name = target.rpartition(".")
    mod = TARGET_ALIASES.get(mod, mod)
    return getattr(importlib.import_module(mod), name)


def instantiate(node: Any, **kwargs) -> Any:
    """Recursive ``_target_`` instantiation (keyword arguments only, like the reference's configs)."""
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return {k: instantiate(v) for k, v in node.items()}
    args = {k: instantiate(v) for k, v in node.items() if not k.startswith("_")}
    args.update(kwargs)
    cls = locate(node["_target_"])
    if node.get("_partial_"):
        from functools import partial
        return partial(cls, **args)
    return cls(**args)


def build_generator(generator: str = "hifigan", resolution: str = "44100_512_2048", overrides: dict | None = None,
                    config_root: str | None = None):
    cfg = compose_model(generator, resolution, overrides, config_root)
    return instantiate(cfg["model"]["generator"]), cfg
