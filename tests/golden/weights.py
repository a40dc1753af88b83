"""Deterministic, seed-addressed parameter/input generators shared by
``make_golden.py`` (which pushes them INTO the reference modules) and by the
tests (which rebuild the very same tensors without the reference).

Only numpy's PCG64 stream is used, so the tensors are bit-identical wherever
numpy >= 1.17 runs; nothing here is derived from reference code.
Scales are deliberately larger than the reference's init (std 0.02) so that
attention logits, GELU and sin() leave their linear regime in the fixtures.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Tuple

import numpy as np


def _draw(rng: np.random.Generator, shape, scale: float, shift: float = 0.0) -> np.ndarray:
    return (rng.standard_normal(size=shape) * scale + shift).astype(np.float32)


def make_state(shapes: Mapping[str, Tuple[int, ...]], seed: int, family: str) -> Dict[str, np.ndarray]:
    """One array per named parameter, drawn in the mapping's iteration order.

    family "vit": v2 VisionTransformer keys;  "gen": v1 Generator keys.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, np.ndarray] = {}
    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        if family == "vit":
            is_ln = (".norm" in name) and ("attention" not in name)
            if is_ln:
                out[name] = _draw(rng, shape, 0.1, 1.0 if leaf == "weight" else 0.0)
            elif leaf == "bias":
                out[name] = _draw(rng, shape, 0.05)
            elif leaf in ("pos_embedding", "cls_token"):
                out[name] = _draw(rng, shape, 0.5)
            else:  # conv / linear weight: fan_in scaled so activations stay O(1)
                fan_in = int(np.prod(shape[1:]))
                out[name] = _draw(rng, shape, 1.0 / math.sqrt(fan_in))
        elif family == "gen":
            if name == "embedding":
                out[name] = _draw(rng, shape, 1.0)
            elif leaf in ("beta", "gamma"):
                out[name] = _draw(rng, shape, 0.5, 0.8)
            elif "layer_norm.weight" in name:
                out[name] = _draw(rng, shape, 0.1, 1.0)
            elif "layer_norm.bias" in name:
                out[name] = _draw(rng, shape, 0.1)
            elif name == "output_network.0.linear.weight":
                out[name] = _draw(rng, shape, 0.5 / shape[1])
            elif name == "output_network.1.linear.weight":
                out[name] = _draw(rng, shape, math.sqrt(2.0 / shape[1]) / 30.0)
            elif leaf == "weight":
                out[name] = _draw(rng, shape, 1.0 / math.sqrt(shape[1]))
            else:
                out[name] = _draw(rng, shape, 0.02)
        elif family == "v1att":  # bias-free q/k/v [hd, E], output_linear [E_out, H*hd] (+ bias)
            if leaf == "bias":
                out[name] = _draw(rng, shape, 0.05)
            else:
                out[name] = _draw(rng, shape, 1.0 / math.sqrt(shape[1]))
        else:
            raise ValueError(family)
    return out


def make_input(shape, seed: int, kind: str = "normal") -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    if kind == "uniform":  # images in [-1, 1]
        return (rng.random(size=shape) * 2.0 - 1.0).astype(np.float32)
    return rng.standard_normal(size=shape).astype(np.float32)


def summarize(a: np.ndarray, n: int = 192) -> Dict[str, np.ndarray]:
    """Compact fingerprint of a tensor: shape, l2 norm, sum and a strided sample."""
    flat = np.asarray(a, dtype=np.float32).reshape(-1)
    stride = max(1, flat.size // n)
    return {
        "shape": np.asarray(a.shape, dtype=np.int64),
        "norm": np.asarray(np.sqrt(np.sum(flat.astype(np.float64) ** 2)), dtype=np.float64),
        "sum": np.asarray(np.sum(flat.astype(np.float64)), dtype=np.float64),
        "sample": flat[::stride][:n].copy(),
    }
