"""fp32 CPU restatement of the reference's v2 ViT (discriminator / generator trunk).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Pinned by tests/golden/vit_*.npz.

Functional style: every function takes a ``state`` mapping that uses the
reference's ``state_dict`` key names (``vit.embedding.conv1.weight`` ...), so a
reference checkpoint can be fed in unchanged.  Each function cites the reference
lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Mapping, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass(frozen=True)
class VitDims:
    """Shape parameters of one VisionTransformer (src/v2/modules.py:202-214)."""

    channels: int = 3
    image: int = 32
    patch: int = 4
    embed: int = 384
    heads: int = 4
    layers: int = 6
    mlp_ratio: int = 2
    classes: int = 1

    @property
    def n_patches(self) -> int:  # src/v2/modules.py:74
        return (self.image // self.patch) ** 2

    @property
    def seq(self) -> int:  # CLS prepended, src/v2/modules.py:96-98
        return self.n_patches + 1

    @property
    def head_dim(self) -> int:  # src/v2/modules.py:108
        return self.embed // self.heads


def vit_param_shapes(d: VitDims, prefix: str = "vit.") -> Dict[str, tuple]:
    """Name -> shape of every parameter, in reference registration order.

    Mirrors the module tree of src/v2/modules.py:67-80 (EmbedLayer), :110-121
    (SelfAttention), :168-176 (Encoder), :190-192 (Classifier), :216-228
    (VisionTransformer).  106 entries for 6 layers.
    """
    E, P, C = d.embed, d.patch, d.channels
    out: Dict[str, tuple] = {}
    out[prefix + "embedding.pos_embedding"] = (1, d.n_patches, E)
    out[prefix + "embedding.cls_token"] = (1, 1, E)
    out[prefix + "embedding.conv1.weight"] = (E, C, P, P)
    out[prefix + "embedding.conv1.bias"] = (E,)
    for i in range(d.layers):
        b = f"{prefix}encoder.{i}."
        out[b + "norm1.weight"] = (E,)
        out[b + "norm1.bias"] = (E,)
        for nm in ("queries", "keys", "values", "out_projection"):
            out[b + f"attention.{nm}.weight"] = (E, E)
            out[b + f"attention.{nm}.bias"] = (E,)
        out[b + "norm2.weight"] = (E,)
        out[b + "norm2.bias"] = (E,)
        out[b + "fc1.weight"] = (E * d.mlp_ratio, E)
        out[b + "fc1.bias"] = (E * d.mlp_ratio,)
        out[b + "fc2.weight"] = (E, E * d.mlp_ratio)
        out[b + "fc2.bias"] = (E,)
    out[prefix + "norm.weight"] = (E,)
    out[prefix + "norm.bias"] = (E,)
    out[prefix + "classifier.fc1.weight"] = (E, E)
    out[prefix + "classifier.fc1.bias"] = (E,)
    out[prefix + "classifier.fc2.weight"] = (d.classes, E)
    out[prefix + "classifier.fc2.bias"] = (d.classes,)
    return out


def patch_embed(state: Mapping[str, Tensor], x: Tensor, d: VitDims, prefix: str = "vit.",
                masks: Optional[Mapping] = None) -> Tensor:
    """EmbedLayer.forward, src/v2/modules.py:82-100.  ``masks`` (test hook): explicit dropout multipliers
    (mask / keep) keyed "embed", ("attn", l), ("mlp", l) standing in for nn.Dropout's RNG (:99,:179,:180).

    A stride-P, kernel-P convolution is a per-patch matrix product: cut the image
    into PxP tiles, flatten each tile in (c, py, px) order - the memory order of
    ``conv1.weight[e]`` - and multiply by W[E, C*P*P]^T.
    """
    B, C, IH, IW = x.shape
    P, E = d.patch, d.embed
    gh, gw = IH // P, IW // P
    w = state[prefix + "embedding.conv1.weight"].reshape(E, C * P * P)
    bias = state[prefix + "embedding.conv1.bias"]
    tiles = x.reshape(B, C, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * P * P)
    tok = tiles @ w.t() + bias  # [B, N, E]   (:84-92)
    tok = tok + state[prefix + "embedding.pos_embedding"]  # (:93-95)
    cls = state[prefix + "embedding.cls_token"].expand(B, 1, E)  # (:96-98)
    out = torch.cat([cls, tok], dim=1)
    if masks is not None and "embed" in masks:
        out = out * masks["embed"]  # self.dropout(x), :99
    return out


def self_attention(state: Mapping[str, Tensor], x: Tensor, heads: int, base: str,
                   taps: Optional[dict] = None) -> Tensor:
    """SelfAttention.forward, src/v2/modules.py:123-162."""
    B, S, E = x.shape
    hd = E // heads

    def proj(name: str) -> Tensor:  # (:128-139) Linear then split heads -> [B,H,S,hd]
        y = F.linear(x, state[base + name + ".weight"], state[base + name + ".bias"])
        return y.reshape(B, S, heads, hd).transpose(1, 2)

    q, k, v = proj("queries"), proj("keys"), proj("values")
    scores = (q @ k.transpose(-1, -2)) / (float(hd) ** 0.5)  # (:142-149)
    prob = torch.softmax(scores, dim=-1)  # (:151)
    if taps is not None:
        taps["attn_prob"] = prob
    ctx = (prob @ v).transpose(1, 2).reshape(B, S, E)  # (:153-159)
    return F.linear(ctx, state[base + "out_projection.weight"], state[base + "out_projection.bias"])  # (:161)


def encoder_block(state: Mapping[str, Tensor], x: Tensor, heads: int, base: str,
                  taps: Optional[dict] = None, m_attn: Optional[Tensor] = None, m_mlp: Optional[Tensor] = None) -> Tensor:
    """Encoder.forward (pre-LN block), src/v2/modules.py:178-183; dropout1/dropout2 = the given multipliers."""
    E = x.shape[-1]
    h = F.layer_norm(x, (E,), state[base + "norm1.weight"], state[base + "norm1.bias"], 1e-5)
    a = self_attention(state, h, heads, base + "attention.", taps)
    x = x + (a if m_attn is None else a * m_attn)
    h = F.layer_norm(x, (E,), state[base + "norm2.weight"], state[base + "norm2.bias"], 1e-5)
    h = F.linear(h, state[base + "fc1.weight"], state[base + "fc1.bias"])
    h = F.gelu(h)  # nn.GELU() default = exact erf form (:174)
    h = F.linear(h, state[base + "fc2.weight"], state[base + "fc2.bias"])
    return x + (h if m_mlp is None else h * m_mlp)


def classifier_head(state: Mapping[str, Tensor], x: Tensor, prefix: str = "vit.") -> Tensor:
    """Classifier.forward, src/v2/modules.py:194-199 (CLS row -> Linear/Tanh/Linear)."""
    c = x[:, 0, :]
    c = torch.tanh(F.linear(c, state[prefix + "classifier.fc1.weight"], state[prefix + "classifier.fc1.bias"]))
    return F.linear(c, state[prefix + "classifier.fc2.weight"], state[prefix + "classifier.fc2.bias"])


def vit_forward(state: Mapping[str, Tensor], x: Tensor, d: VitDims, prefix: str = "vit.",
                taps: Optional[dict] = None, masks: Optional[Mapping] = None) -> Tensor:
    """VisionTransformer.forward, src/v2/modules.py:232-238 ( == ViTDiscriminator.forward :393-395)."""
    h = patch_embed(state, x, d, prefix, masks)
    if taps is not None:
        taps["embed"] = h
        taps["blocks"] = []
    for i in range(d.layers):
        blk_taps = {} if (taps is not None and i == 0) else None
        ma = masks.get(("attn", i)) if masks is not None else None
        mm = masks.get(("mlp", i)) if masks is not None else None
        h = encoder_block(state, h, d.heads, f"{prefix}encoder.{i}.", blk_taps, ma, mm)
        if taps is not None:
            taps["blocks"].append(h)
            if blk_taps:
                taps.update(blk_taps)
    E = d.embed
    h = F.layer_norm(h, (E,), state[prefix + "norm.weight"], state[prefix + "norm.bias"], 1e-5)
    return classifier_head(state, h, prefix)


def vit_generator_v2_forward(state: Mapping[str, Tensor], x: Tensor, d: VitDims) -> Tensor:
    """ViTGenerator.forward, src/v2/modules.py:368-372, including its flat ``view``.

    The final view raises exactly like the reference unless
    ``B * batch_size`` is a multiple of ``C*IH*IW``.
    """
    y = vit_forward(state, x, d, "vit.")
    y = F.linear(y, state["linear.weight"], state["linear.bias"])
    return y.view(-1, d.channels, d.image, d.image)


def init_vit_state(d: VitDims, seed: int, prefix: str = "vit.") -> Dict[str, Tensor]:
    """Random state following vit_init_weights, src/v2/modules.py:241-253.

    Distribution only (trunc-normal std .02 in +-2 for weights/cls/pos, zeros
    for biases, LN = (1, 0)); the RNG stream differs from the reference's.
    """
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, Tensor] = {}
    for name, shape in vit_param_shapes(d, prefix).items():
        leaf = name.rsplit(".", 1)[-1]
        is_ln = ".norm" in name and "attention" not in name
        if is_ln:
            st[name] = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
        elif leaf == "bias":
            st[name] = torch.zeros(shape)
        else:
            t = torch.empty(shape)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=0.02, generator=g)
            st[name] = t
    return st


def matmul_flops_per_image(d: VitDims) -> float:
    """Algorithmic matmul FLOPs of one forward per image (SURVEY 8d / BASELINE.md 3)."""
    N, S, E, r, L = d.n_patches, d.seq, d.embed, d.mlp_ratio, d.layers
    ckk = d.channels * d.patch * d.patch
    per_layer = 2 * S * E * 3 * E + 2 * S * S * E + 2 * S * S * E + 2 * S * E * E + 2 * (2 * S * E * r * E)
    return 2 * N * ckk * E + L * per_layer + 2 * E * E + 2 * E * d.classes
