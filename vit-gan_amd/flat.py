"""Flat parameter layouts: reference state_dict name -> (element offset, shape).

The offsets come from the C library (vg_vit_layout / vg_gen_layout) so that the
Python views and the kernels share one source of truth.  Key names and shapes are
the reference's (src/v2/modules.py module tree; src/v1/generator.py).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Tuple

from . import _lib

Slot = Tuple[int, Tuple[int, ...]]


def vit_dims_struct(channels, image, patch, embed, heads, layers, mlp_ratio, classes) -> _lib.VgVitDims:
    return _lib.VgVitDims(channels, image, patch, embed, heads, layers, mlp_ratio, classes)


def vit_layout(d: _lib.VgVitDims) -> _lib.VgVitLayout:
    lay = _lib.VgVitLayout()
    _lib.check(_lib.lib().vg_vit_layout(C.byref(d), C.byref(lay)), "vg_vit_layout (unsupported ViT shape)")
    return lay


def vit_slots(d: _lib.VgVitDims, prefix: str = "vit.") -> "OrderedDict[str, Slot]":
    """Reference registration order (src/v2/modules.py:67-80,110-121,168-176,190-192,216-228)."""
    lay = vit_layout(d)
    E, P, Cc, r = d.E, d.P, d.C, d.R
    NP = (d.IH // d.P) ** 2
    s: "OrderedDict[str, Slot]" = OrderedDict()
    s[prefix + "embedding.pos_embedding"] = (lay.pos, (1, NP, E))
    s[prefix + "embedding.cls_token"] = (lay.cls, (1, 1, E))
    s[prefix + "embedding.conv1.weight"] = (lay.conv_w, (E, Cc, P, P))
    s[prefix + "embedding.conv1.bias"] = (lay.conv_b, (E,))
    for i in range(d.L):
        lo = lay.layer0 + i * lay.layer_stride
        b = f"{prefix}encoder.{i}."
        s[b + "norm1.weight"] = (lo + lay.ln1_w, (E,))
        s[b + "norm1.bias"] = (lo + lay.ln1_b, (E,))
        for j, nm in enumerate(("queries", "keys", "values")):
            s[b + f"attention.{nm}.weight"] = (lo + lay.wqkv + j * E * E, (E, E))
            s[b + f"attention.{nm}.bias"] = (lo + lay.bqkv + j * E, (E,))
        s[b + "attention.out_projection.weight"] = (lo + lay.wo, (E, E))
        s[b + "attention.out_projection.bias"] = (lo + lay.bo, (E,))
        s[b + "norm2.weight"] = (lo + lay.ln2_w, (E,))
        s[b + "norm2.bias"] = (lo + lay.ln2_b, (E,))
        s[b + "fc1.weight"] = (lo + lay.w1, (r * E, E))
        s[b + "fc1.bias"] = (lo + lay.b1, (r * E,))
        s[b + "fc2.weight"] = (lo + lay.w2, (E, r * E))
        s[b + "fc2.bias"] = (lo + lay.b2, (E,))
    s[prefix + "norm.weight"] = (lay.lnf_w, (E,))
    s[prefix + "norm.bias"] = (lay.lnf_b, (E,))
    s[prefix + "classifier.fc1.weight"] = (lay.hw1, (E, E))
    s[prefix + "classifier.fc1.bias"] = (lay.hb1, (E,))
    s[prefix + "classifier.fc2.weight"] = (lay.hw2, (d.Kc, E))
    s[prefix + "classifier.fc2.bias"] = (lay.hb2, (d.Kc,))
    return s


def gen_layout(d: _lib.VgGenDims) -> _lib.VgGenLayout:
    lay = _lib.VgGenLayout()
    _lib.check(_lib.lib().vg_gen_layout(C.byref(d), C.byref(lay)), "vg_gen_layout (unsupported generator shape)")
    return lay


def gen_slots(d: _lib.VgGenDims) -> "OrderedDict[str, Slot]":
    """``src.v1.generator.Generator().state_dict()`` order."""
    lay = gen_layout(d)
    E, T, hd = d.E, d.T, d.E // d.H
    s: "OrderedDict[str, Slot]" = OrderedDict()
    s["embedding"] = (lay.emb, (T, E))
    s["mapping_mlp.model.0.0.weight"] = (lay.map_w, (T * E, d.Z))
    s["mapping_mlp.model.0.0.bias"] = (lay.map_b, (T * E,))
    for i in range(d.L):
        lo = lay.layer0 + i * lay.layer_stride
        b = f"transformer_layers.{i}."
        for ln, (w_, b_, s_) in (("layer_norm_1", (lay.sln1_w, lay.sln1_b, lay.sln1_s)),
                                 ("layer_norm_2", (lay.sln2_w, lay.sln2_b, lay.sln2_s))):
            s[b + ln + ".beta"] = (lo + s_ + 1, (1, 1, 1))
            s[b + ln + ".gamma"] = (lo + s_, (1, 1, 1))
            s[b + ln + ".layer_norm.weight"] = (lo + w_, (E,))
            s[b + ln + ".layer_norm.bias"] = (lo + b_, (E,))
        for h in range(d.H):
            for j, nm in enumerate(("q", "k", "v")):  # fused [3E,E]: all q heads | all k heads | all v heads
                s[b + f"msha.attention_heads.{h}.{nm}.weight"] = (lo + lay.wqkv + (j * E + h * hd) * E, (hd, E))
        s[b + "msha.output_linear.weight"] = (lo + lay.wo, (E, E))
        s[b + "msha.output_linear.bias"] = (lo + lay.bo, (E,))
        s[b + "mlp.model.0.0.weight"] = (lo + lay.wm, (E, E))
        s[b + "mlp.model.0.0.bias"] = (lo + lay.bm, (E,))
    s["sln.beta"] = (lay.slnf_s + 1, (1, 1, 1))
    s["sln.gamma"] = (lay.slnf_s, (1, 1, 1))
    s["sln.layer_norm.weight"] = (lay.slnf_w, (E,))
    s["sln.layer_norm.bias"] = (lay.slnf_b, (E,))
    s["output_network.0.linear.weight"] = (lay.s1_w, (d.O, E))
    s["output_network.0.linear.bias"] = (lay.s1_b, (d.O,))
    s["output_network.1.linear.weight"] = (lay.s2_w, (d.CW, d.O))
    s["output_network.1.linear.bias"] = (lay.s2_b, (d.CW,))
    return s


def numel(shape) -> int:
    n = 1
    for x in shape:
        n *= int(x)
    return n


def pack(slots: "Dict[str, Slot]", total: int, state, device=None, dtype=None):
    """Flat fp32 tensor holding ``state`` (name -> tensor/ndarray) at the slot offsets."""
    import torch

    flat = torch.zeros(total, dtype=torch.float32)
    for name, (off, shape) in slots.items():
        t = torch.as_tensor(state[name], dtype=torch.float32).reshape(-1)
        assert t.numel() == numel(shape), name
        flat[off:off + t.numel()] = t
    if device is not None:
        flat = flat.to(device)
    return flat


def unpack(slots: "Dict[str, Slot]", flat):
    return {name: flat[off:off + numel(shape)].view(shape) for name, (off, shape) in slots.items()}
