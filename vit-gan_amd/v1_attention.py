"""The reference's v1 attention modules on the HIP kernels: ``Attention`` and
``MultiHeadSelfAttention`` of src/v1/attention.py with both score functions - ``lp = 1`` dot product (generator) and
``lp = 2`` Euclidean distance (discriminator, attention.py:66-67) - and the per-forward spectral rescale (:54-64).

Same constructor arguments, attribute names and ``state_dict`` keys as the reference
(``attention_heads.<h>.{q,k,v}.weight``, ``output_linear.{weight,bias}``).  The H per-head projections run as ONE fused
[3*H*hd, E] GEMM (the weights are concatenated per call - they are separate Parameters in the state_dict) and all heads
go through one fused attention launch.  No CPU fallback.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
from torch import nn

from . import ops


def TransformerParameters(number_of_heads=4, attention_dropout_rate=0.2, mlp_layers=(), mlp_activation="relu", mlp_dropout=0.2,
                          input_features=None, spectral_scaling=None, lp=None):
    """Field-for-field stand-in for src/v1/config.py:36-44 (a plain namespace; the reference's is a pydantic model)."""
    return SimpleNamespace(number_of_heads=number_of_heads, attention_dropout_rate=attention_dropout_rate, mlp_layers=list(mlp_layers),
                           mlp_activation=mlp_activation, mlp_dropout=mlp_dropout, input_features=input_features,
                           spectral_scaling=spectral_scaling, lp=lp)


class Attention(nn.Module):
    """src/v1/attention.py:7-70.  A single head; inside ``MultiHeadSelfAttention`` the heads are evaluated together."""

    def __init__(self, transformer_parameters, output_features, scale=None):
        super().__init__()
        self.output_features = output_features
        self.scale = output_features if scale is None else scale
        self.spectral_scaling = transformer_parameters.spectral_scaling
        assert transformer_parameters.lp in [1, 2], \
            f"Unsupported norm for attention: lp={transformer_parameters.lp} but should be 1 or 2"
        self.lp = transformer_parameters.lp
        self.q = nn.Linear(transformer_parameters.input_features, output_features, bias=False)
        self.k = nn.Linear(transformer_parameters.input_features, output_features, bias=False)
        self.v = nn.Linear(transformer_parameters.input_features, output_features, bias=False)
        if self.spectral_scaling:
            self.init_spectrum = [float(s) for s in self._max_spectrum()]

    def _max_spectrum(self):
        return [torch.linalg.svdvals(w.detach().float()).max() for w in (self.q.weight, self.k.weight, self.v.weight)]

    def _weight_spectral_rescale(self):
        """attention.py:60-64: every forward replaces each weight by init_sigma / sigma_max(W) * W (a new Parameter)."""
        for lin, s0, s in zip((self.q, self.k, self.v), self.init_spectrum, self._max_spectrum()):
            lin.weight = nn.Parameter(s0 / s * lin.weight.detach())

    def forward(self, x):
        if self.spectral_scaling:
            self._weight_spectral_rescale()
        w = torch.cat([self.q.weight, self.k.weight, self.v.weight], dim=0)  # [3*hd, E]
        qkv = ops.linear(x, w)
        return ops.attention(qkv, 1, 1.0 / math.sqrt(float(self.scale)), self.lp)


class MultiHeadSelfAttention(nn.Module):
    """src/v1/attention.py:73-103."""

    def __init__(self, transformer_parameters, output_size: int, head_dimension: int):
        super().__init__()
        self.output_dimension = transformer_parameters.number_of_heads * head_dimension
        self.output_features = output_size
        self.attention_heads = nn.ModuleList([
            Attention(transformer_parameters=transformer_parameters, output_features=head_dimension, scale=self.output_dimension)
            for _ in range(transformer_parameters.number_of_heads)])
        self.output_linear = nn.Linear(self.output_dimension, self.output_features)

    def forward(self, x):
        heads = list(self.attention_heads)
        for h in heads:
            if h.spectral_scaling:
                h._weight_spectral_rescale()
        # all q heads | all k heads | all v heads: the layout the fused kernel indexes by column offset
        w = torch.cat([getattr(h, nm).weight for nm in ("q", "k", "v") for h in heads], dim=0)
        qkv = ops.linear(x, w)
        att = ops.attention(qkv, len(heads), 1.0 / math.sqrt(float(self.output_dimension)), heads[0].lp)
        return ops.linear(att, self.output_linear.weight, self.output_linear.bias)
