"""fp32 CPU restatement of the reference's v1 generator (SLN blocks + SIREN output).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Pinned by tests/golden/gen_*.npz.

State keys are those of ``src.v1.generator.Generator().state_dict()``.
Dropout layers (attention 0.2 / mlp 0.2, src/v1/config.py:36,39) are identity
here: parity is defined in eval mode / p = 0 (RNG streams cannot be matched).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Mapping, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass(frozen=True)
class GenDims:
    """src/v1/config.py:45-49,60-66 defaults."""

    latent: int = 1024
    tokens: int = 32  # = image_size: one token per image row (src/v1/generator.py:19,25)
    embed: int = 384
    heads: int = 4
    layers: int = 4
    siren_hidden: int = 768
    channels: int = 3
    image: int = 32
    omega0: float = 30.0
    patch: int = 0  # 0: reference layout (token = image row, flat view); > 0: patch-grid variant (unpinned)

    @property
    def head_dim(self) -> int:  # src/v1/transformer.py:54-57
        return self.embed // self.heads

    @property
    def out_features(self) -> int:  # src/v1/generator.py:51
        return self.channels * (self.patch * self.patch if self.patch else self.image)


def gen_param_shapes(d: GenDims) -> Dict[str, tuple]:
    """Name -> shape in ``Generator().state_dict()`` order (src/v1/generator.py:12-55)."""
    E, T, hd = d.embed, d.tokens, d.head_dim
    out: Dict[str, tuple] = {}
    out["embedding"] = (T, E)
    out["mapping_mlp.model.0.0.weight"] = (T * E, d.latent)
    out["mapping_mlp.model.0.0.bias"] = (T * E,)
    for i in range(d.layers):
        b = f"transformer_layers.{i}."
        for ln in ("layer_norm_1", "layer_norm_2"):
            out[b + ln + ".beta"] = (1, 1, 1)
            out[b + ln + ".gamma"] = (1, 1, 1)
            out[b + ln + ".layer_norm.weight"] = (E,)
            out[b + ln + ".layer_norm.bias"] = (E,)
        for h in range(d.heads):
            for nm in ("q", "k", "v"):
                out[b + f"msha.attention_heads.{h}.{nm}.weight"] = (hd, E)
        out[b + "msha.output_linear.weight"] = (E, E)
        out[b + "msha.output_linear.bias"] = (E,)
        out[b + "mlp.model.0.0.weight"] = (E, E)
        out[b + "mlp.model.0.0.bias"] = (E,)
    out["sln.beta"] = (1, 1, 1)
    out["sln.gamma"] = (1, 1, 1)
    out["sln.layer_norm.weight"] = (E,)
    out["sln.layer_norm.bias"] = (E,)
    out["output_network.0.linear.weight"] = (d.siren_hidden, E)
    out["output_network.0.linear.bias"] = (d.siren_hidden,)
    out["output_network.1.linear.weight"] = (d.out_features, d.siren_hidden)
    out["output_network.1.linear.bias"] = (d.out_features,)
    return out


def sln(state: Mapping[str, Tensor], base: str, h: Tensor, w: Tensor) -> Tensor:
    """SLN.forward, src/v1/spectral_layer_norm.py:19-20: gamma*w*LN(h) + beta*w."""
    E = h.shape[-1]
    ln = F.layer_norm(h, (E,), state[base + "layer_norm.weight"], state[base + "layer_norm.bias"], 1e-5)
    return state[base + "gamma"] * w * ln + state[base + "beta"] * w


def mhsa(state: Mapping[str, Tensor], base: str, x: Tensor, d: GenDims,
         taps: Optional[dict] = None) -> Tensor:
    """MultiHeadSelfAttention.forward, src/v1/attention.py:97-103 with lp=1 heads (:43-52,:69-70).

    Per head: bias-free q/k/v, dot-product scores, softmax(scores / sqrt(H*hd)),
    i.e. the divisor is sqrt(E) not sqrt(hd) (scale=self.output_dimension, :90).
    """
    outs = []
    div = math.sqrt(float(d.heads * d.head_dim))
    for h in range(d.heads):
        hb = f"{base}attention_heads.{h}."
        q = F.linear(x, state[hb + "q.weight"])
        k = F.linear(x, state[hb + "k.weight"])
        v = F.linear(x, state[hb + "v.weight"])
        p = torch.softmax((q @ k.transpose(-1, -2)) / div, dim=-1)
        if taps is not None and h == 0:
            taps["attn_prob_h0"] = p
        outs.append(p @ v)
    cat = torch.cat(outs, dim=-1)
    return F.linear(cat, state[base + "output_linear.weight"], state[base + "output_linear.bias"])


def sln_block(state: Mapping[str, Tensor], base: str, h: Tensor, w: Tensor, d: GenDims,
              taps: Optional[dict] = None, m_attn: Optional[Tensor] = None, m_mlp: Optional[Tensor] = None) -> Tensor:
    """TransformerSLN.forward, src/v1/transformer.py:85-88 (``w`` is returned unchanged there).
    m_attn / m_mlp (test hook): explicit multipliers standing in for attention_dropout / the MLP's Dropout."""
    a = mhsa(state, base + "msha.", sln(state, base + "layer_norm_1.", h, w), d, taps)
    htmp = (a if m_attn is None else a * m_attn) + h
    # MLP with layers=[] is a single Linear, no activation (src/v1/muilti_layer_perceptron.py:37-42)
    m = F.linear(sln(state, base + "layer_norm_2.", htmp, w),
                 state[base + "mlp.model.0.0.weight"], state[base + "mlp.model.0.0.bias"])
    return (m if m_mlp is None else m * m_mlp) + htmp


def siren(state: Mapping[str, Tensor], base: str, x: Tensor, omega0: float) -> Tensor:
    """SIREN.forward, src/v1/siren.py:44-45."""
    return torch.sin(omega0 * F.linear(x, state[base + "linear.weight"], state[base + "linear.bias"]))


def fourier_position_table(d: GenDims) -> Tensor:
    """Optional Fourier positional input of the SIREN (named by north_star, ABSENT from the reference: parity unpinned;
    this function is the definition).  [T, E] table: token t sits at (x, y) in [0,1)^2 - its patch-grid cell centre
    (patch > 0) or (0.5, row centre) for the v1 row tokens - and column e = 4*j + c holds
    (sin, cos)(2 pi f_j x), (sin, cos)(2 pi f_j y) for c = 0..3 with E/4 log-spaced frequencies f_j from 1 to side/2."""
    T, E = d.tokens, d.embed
    side = d.image // d.patch if d.patch else d.tokens
    t = torch.arange(T, dtype=torch.float64)
    if d.patch:
        x, y = ((t % side) + 0.5) / side, (torch.div(t, side, rounding_mode="floor") + 0.5) / side
    else:
        x, y = torch.full((T,), 0.5, dtype=torch.float64), (t + 0.5) / side
    nb = E // 4
    f = torch.pow(torch.tensor(max(side / 2.0, 1.0), dtype=torch.float64), torch.arange(nb, dtype=torch.float64) / max(nb - 1, 1))
    ax, ay = 2 * math.pi * x[:, None] * f[None, :], 2 * math.pi * y[:, None] * f[None, :]
    return torch.stack([torch.sin(ax), torch.cos(ax), torch.sin(ay), torch.cos(ay)], dim=-1).reshape(T, E).float()


def gen_forward(state: Mapping[str, Tensor], z: Tensor, d: GenDims, taps: Optional[dict] = None,
                masks: Optional[Mapping] = None, pos_table: Optional[Tensor] = None) -> Tensor:
    """Generator.forward, src/v1/generator.py:58-69.  pos_table: the optional Fourier positional input (not in the
    reference), added to the final SLN output."""
    B = z.shape[0]
    w = F.linear(z, state["mapping_mlp.model.0.0.weight"], state["mapping_mlp.model.0.0.bias"])
    w = w.view(B, d.tokens, d.embed)  # (:59-61)
    h = state["embedding"]  # [T,E], broadcasts over the batch in the first block (:62)
    if taps is not None:
        taps["w"] = w
        taps["blocks"] = []
    for i in range(d.layers):
        ma = masks.get(("attn", i)) if masks is not None else None
        mm = masks.get(("mlp", i)) if masks is not None else None
        h = sln_block(state, f"transformer_layers.{i}.", h, w, d, taps if i == 0 else None, ma, mm)
        if taps is not None:
            taps["blocks"].append(h)
    y = sln(state, "sln.", h, w)  # (:65)
    if pos_table is not None:
        y = y + pos_table
    y = siren(state, "output_network.0.", y, d.omega0)
    y = siren(state, "output_network.1.", y, d.omega0)  # [B, T, C*IW]
    if d.patch:  # patch-grid variant: token t = (gy, gx) carries one C x P x P patch in (c, py, px) order
        gh, P = d.image // d.patch, d.patch
        return y.view(B, gh, gh, d.channels, P, P).permute(0, 3, 1, 4, 2, 5).reshape(B, d.channels, d.image, d.image)
    return y.view(B, d.channels, d.image, d.image)  # flat reinterpretation (:66-68)


def init_gen_state(d: GenDims, seed: int) -> Dict[str, Tensor]:
    """Random state with the reference's init distributions.

    embedding ~ N(0,1) (src/v1/generator.py:24-26); SLN beta/gamma ~ N(0,1)
    (spectral_layer_norm.py:16-17); nn.Linear default (kaiming-uniform a=sqrt5 =
    U(+-1/sqrt(in)) for weight and bias); SIREN weights U(+-1/in) first,
    U(+-sqrt(6/in)/omega0) otherwise (siren.py:29-42), bias nn.Linear default.
    """
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, Tensor] = {}

    def uni(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    for name, shape in gen_param_shapes(d).items():
        if name == "embedding" or name.endswith(("beta", "gamma")):
            st[name] = torch.randn(shape, generator=g)
        elif "layer_norm.weight" in name:
            st[name] = torch.ones(shape)
        elif "layer_norm.bias" in name:
            st[name] = torch.zeros(shape)
        elif name == "output_network.0.linear.weight":
            st[name] = uni(shape, 1.0 / shape[1])
        elif name == "output_network.1.linear.weight":
            st[name] = uni(shape, math.sqrt(6.0 / shape[1]) / d.omega0)
        elif name.endswith("weight"):
            st[name] = uni(shape, 1.0 / math.sqrt(shape[1]))
        else:  # Linear bias: fan_in of its weight
            fan_in = gen_param_shapes(d)[name[:-4] + "weight"][1]
            st[name] = uni(shape, 1.0 / math.sqrt(fan_in))
    return st


def matmul_flops_per_image(d: GenDims) -> float:
    """F_G1 of SURVEY 8d: 243.79 MFLOP at the defaults."""
    Z, T, E, O = d.latent, d.tokens, d.embed, d.siren_hidden
    per_layer = 2 * T * E * 3 * E + 4 * T * T * E + 2 * T * E * E + 2 * T * E * E
    return 2 * Z * T * E + d.layers * per_layer + 2 * T * E * O + 2 * T * O * d.out_features
