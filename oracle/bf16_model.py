"""Rounding-faithful CPU model of the HIP engine's arithmetic (second parity tier).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

``vit_oracle`` / ``gen_oracle`` restate the reference in fp32 and are pinned to the reference's own outputs.  The HIP
engine computes the same networks with bf16 tensors in HBM and fp32 accumulation, so comparing it with the fp32 oracle
needs bounds wide enough for ~50 bf16 roundings in sequence (and for sin(30 x) behind them) - wide enough to hide a
mis-scaled scalar gradient.  This module is the same computation with a bf16 rounding at EXACTLY the places where the
kernels store bf16 (and nowhere else: statistics, log-sum-exp, SIREN pre-activations, accumulators, gradients of
parameters, the fp32 gradient of the modulation vector stay fp32), forward and backward:

  * GEMM operands: activations as stored (bf16), weights = the bf16 shadow of the fp32 master (gradient passes straight
    through to the master);
  * every activation the forward stores: X[l], LN/SLN outputs, qkv, attention output, x_mid, gelu(.) and gelu'(.) (the derivative as one byte, grid 1/200),
    tanh(.), sin(.) ; the unnormalised softmax numerator is rounded before P.V (the kernels feed the exponentiated
    accumulators to the MFMA as bf16), the denominator is the fp32 row sum;
  * every gradient tensor the backward stores: dL/dX[l], the LN-input gradients, d qkv, d(attention out), d z1 (after the
    stored gelu' is applied to the fp32 accumulator), masked copies (rounded again after the dropout factor), d image;
    attention backward recomputes P from the fp32 lse, rounds P and dS before their MFMAs, delta uses the STORED output.

It is validated two ways: against the fp32 oracle on CPU (tests/test_bf16_model_cpu.py: it must stay within the loose
bf16 bounds of the first tier, and collapse to the fp32 oracle bit-for-bit when rounding is switched off) and against the
kernels on the GPU stage by stage, each stage fed the tensors the engine produced for it, at two bf16 ulps of the
largest element (tests/test_blocks_gpu.py; whole networks are chaotic at the ulp level beyond ~1 block, tests/parity_tiers.py).  Citations: the same reference lines as the
functions of vit_oracle.py / gen_oracle.py they mirror; kernel sites are named in the comments.
"""
from __future__ import annotations

import math
from typing import Mapping, Optional

import torch
import torch.nn.functional as F

from .gen_oracle import GenDims
from .vit_oracle import VitDims

Tensor = torch.Tensor

_ROUND = True  # tests switch this off to prove the model collapses to the fp32 oracle


def bf(x: Tensor) -> Tensor:
    """Value-level bf16 rounding (round-to-nearest-even, what v_cvt_pk_bf16_f32 does)."""
    return x.to(torch.bfloat16).float() if _ROUND else x


class _Stored(torch.autograd.Function):
    """A tensor the engine keeps in HBM as bf16, whose gradient it also keeps as bf16."""

    @staticmethod
    def forward(ctx, x):
        return bf(x)

    @staticmethod
    def backward(ctx, g):
        return bf(g)


class _Shadow(torch.autograd.Function):
    """bf16 shadow of an fp32 master weight (FlatParams.shadow): rounded on use, fp32 gradient to the master."""

    @staticmethod
    def forward(ctx, w):
        return bf(w)

    @staticmethod
    def backward(ctx, g):
        return g


class _Drop(torch.autograd.Function):
    """Fused dropout.  Forward: the epilogue multiplies the fp32 value (no extra rounding).  Backward: the masked copy
    of the gradient is a bf16 tensor of its own (norm.hip vg_ln_bwd_kernel `dxm`, vg_dropout_apply)."""

    @staticmethod
    def forward(ctx, x, m):
        ctx.save_for_backward(m)
        return x * m

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        return bf(g * m), None


class _GeluStore(torch.autograd.Function):
    """fc1 epilogue (gemm.hip / gemm_wr.hip ACT_GELU with c2_gelu_grad = 2): stores bf16 gelu(pre) AND gelu'(pre) as one byte per
    element on the grid k/200 (code = round(200 g) + 27, vg_common.h vg_g8_pack4); the fc2 dgrad epilogue multiplies its fp32
    accumulator by the decoded derivative and stores bf16 (VG_ACT_MUL_Z8)."""

    @staticmethod
    def forward(ctx, pre):
        phi = 0.5 * (1.0 + torch.erf(pre * 0.7071067811865476))
        gd = phi + pre * 0.3989422804014327 * torch.exp(-0.5 * pre * pre)
        if _ROUND:
            code = torch.clamp(torch.round(gd * 200.0 + 27.0), 0.0, 255.0)
            gd = (code - 27.0) * 0.005
        ctx.save_for_backward(gd)
        return bf(pre * phi)

    @staticmethod
    def backward(ctx, g):
        (gd,) = ctx.saved_tensors
        return bf(g * gd)


class _TanhStore(torch.autograd.Function):
    """classifier fc1 (ACT_TANH) stores bf16 tanh; vg_head_bwd_dz_kernel uses the stored value: dz = bf16(g (1 - t^2))."""

    @staticmethod
    def forward(ctx, pre):
        t = bf(torch.tanh(pre))
        ctx.save_for_backward(t)
        return t

    @staticmethod
    def backward(ctx, g):
        (t,) = ctx.saved_tensors
        return bf(g * (1.0 - t * t))


class _SirenStore(torch.autograd.Function):
    """SIREN layer (ACT_SIN with pre_f32): y = bf16(sin(w0 pre)), pre kept in fp32; backward (vg_sin_grad_kernel /
    VG_ACT_MUL_COS on the fp32 accumulator): d pre = bf16(g w0 cos(w0 pre))."""

    @staticmethod
    def forward(ctx, pre, w0):
        ctx.save_for_backward(pre)
        ctx.w0 = w0
        return bf(torch.sin(w0 * pre))

    @staticmethod
    def backward(ctx, g):
        (pre,) = ctx.saved_tensors
        return bf(g * ctx.w0 * torch.cos(ctx.w0 * pre)), None


def e4m3(x: Tensor) -> Tensor:
    """Value-level OCP e4m3 rounding (what v_cvt_pk_fp8_f32 does to an MFMA operand of the fp8 attention mode)."""
    return x.to(torch.float8_e4m3fn).float() if _ROUND else x


class _Attention(torch.autograd.Function):
    """attention.hip vg_attn_fwd_kernel / vg_attn_bwd_kernel on [B,H,S,hd] operands (already bf16-valued).

    fp8 = True is the C5 mode (VgVitNet.attn_fp8, BASELINE.json configs[4]; src/v2/modules.py:142-155 with e4m3 MFMA operands):
    the ACTIVATION-side products take OCP e4m3 operands - Q and K for the scores, in the forward and in the backward's
    recompute (so P matches the forward's lse), and 256 p and V for P.V (the numerators scaled into e4m3's normal range) -
    while every product with a gradient operand (dP, dV, dQ, dK) stays bf16 on the bf16 Q, K, V: gradients need the range."""

    @staticmethod
    def forward(ctx, q, k, v, scale, fp8=False):
        if fp8:
            s = (e4m3(q) @ e4m3(k).transpose(-1, -2)) * scale
        else:
            s = (q @ k.transpose(-1, -2)) * scale
        m = s.max(dim=-1, keepdim=True).values
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)          # fp32 row sum of the UNrounded numerators
        if fp8:
            o = (e4m3(p * 256.0) @ e4m3(v)) / 256.0 / l
        else:
            o = (bf(p) @ v) / l                  # numerators rounded for the MFMA (pack_pair)
        ctx.save_for_backward(q, k, v, m + torch.log(l), bf(o))
        ctx.scale = scale
        ctx.fp8 = fp8
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, lse, o_stored = ctx.saved_tensors
        if ctx.fp8:
            s = (e4m3(q) @ e4m3(k).transpose(-1, -2)) * ctx.scale
        else:
            s = (q @ k.transpose(-1, -2)) * ctx.scale
        p = torch.exp(s - lse)                                       # recomputed from the fp32 lse
        dp = do @ v.transpose(-1, -2)
        delta = (do * o_stored).sum(dim=-1, keepdim=True)            # from the stored bf16 output
        ds = bf(p * (dp - delta) * ctx.scale)
        return ds @ k, ds.transpose(-1, -2) @ q, bf(p).transpose(-1, -2) @ do, None, None


class _Mapping(torch.autograd.Function):
    """Generator mapping Linear (vg_gen_forward / vg_gen_backward tail): w = bf16(z W^T + b).  Its gradient is summed in
    fp32 over the 2L+1 SLN uses (dw_acc); the weight gradient uses the bf16 cast of that sum, the bias gradient the fp32
    sum itself (vg_colsum_f32 on dw_acc)."""

    @staticmethod
    def forward(ctx, z, w, b):
        ctx.save_for_backward(z)
        return bf(z @ w.t() + b)

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        return None, bf(g).t() @ z, g.sum(dim=0)


class _EmbeddingUses(torch.autograd.Function):
    """Block 0 of the generator reads the learned embedding twice: as the SLN input from the bf16 shadow and as the
    residual from the fp32 master (engine.hip vg_gen_forward, l == 0); the backward writes ONE bf16 gradient tensor per
    sample for both paths (the SLN1 backward's output) which vg_batch_sum then sums over the batch."""

    @staticmethod
    def forward(ctx, e):
        return bf(e), e.clone()

    @staticmethod
    def backward(ctx, g_sln, g_res):
        return bf(g_sln + g_res)


def stored(x: Tensor) -> Tensor:
    return _Stored.apply(x)


def shadow(w: Tensor) -> Tensor:
    return _Shadow.apply(w)


def drop(x: Tensor, m: Optional[Tensor]) -> Tensor:
    return x if m is None else _Drop.apply(x, m)


def _ln(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


# ---------------------------------------------------------------------------------------------------------------------
# v2 VisionTransformer (mirrors vit_oracle.vit_forward; engine.hip vg_vit_forward / vg_vit_backward_stages).
# Split at the tensors the engine keeps between stages (X[l]), so a test can run ONE stage on the engine's own inputs.
# ---------------------------------------------------------------------------------------------------------------------
def vit_embed(state: Mapping[str, Tensor], x: Tensor, d: VitDims, prefix: str = "vit.", mask: Optional[Tensor] = None) -> Tensor:
    """image -> X[0]  (vg_patchify + embedding GEMM with bias / position / dropout epilogue + vg_fill_cls)."""
    B, C, IH, IW = x.shape
    P, E = d.patch, d.embed
    gh, gw = IH // P, IW // P
    xr = stored(x)  # vg_patchify casts the image to bf16; d_img leaves as bf16
    tiles = xr.reshape(B, C, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * P * P)
    wc = shadow(state[prefix + "embedding.conv1.weight"]).reshape(E, C * P * P)
    tok = tiles @ wc.t() + state[prefix + "embedding.conv1.bias"] + state[prefix + "embedding.pos_embedding"]
    cls = state[prefix + "embedding.cls_token"].expand(B, 1, E)
    return stored(drop(torch.cat([cls, tok], dim=1), mask))  # dropout after every addend (drop_post = 1)


def vit_block(state: Mapping[str, Tensor], h: Tensor, d: VitDims, b: str, m_attn: Optional[Tensor] = None,
              m_mlp: Optional[Tensor] = None, fp8: bool = False) -> Tensor:
    """X[l] -> X[l+1]; ``h`` must be a stored tensor (or, when teacher-forcing, ``stored(leaf)``); fp8: e4m3 attention operands."""
    B, S, E = h.shape
    H, hd = d.heads, d.head_dim
    xn1 = stored(_ln(h, state[b + "norm1.weight"], state[b + "norm1.bias"]))
    wqkv = torch.cat([shadow(state[b + f"attention.{n}.weight"]) for n in ("queries", "keys", "values")], dim=0)
    bqkv = torch.cat([state[b + f"attention.{n}.bias"] for n in ("queries", "keys", "values")], dim=0)
    qkv = stored(xn1 @ wqkv.t() + bqkv)
    q, k, v = (t.reshape(B, S, H, hd).transpose(1, 2) for t in qkv.split(E, dim=-1))
    ao = stored(_Attention.apply(q, k, v, 1.0 / math.sqrt(float(hd)), fp8).transpose(1, 2).reshape(B, S, E))
    lin = ao @ shadow(state[b + "attention.out_projection.weight"]).t() + state[b + "attention.out_projection.bias"]
    xmid = stored(drop(lin, m_attn) + h)
    xn2 = stored(_ln(xmid, state[b + "norm2.weight"], state[b + "norm2.bias"]))
    a1 = _GeluStore.apply(xn2 @ shadow(state[b + "fc1.weight"]).t() + state[b + "fc1.bias"])
    lin = a1 @ shadow(state[b + "fc2.weight"]).t() + state[b + "fc2.bias"]
    return stored(drop(lin, m_mlp) + xmid)


def vit_head(state: Mapping[str, Tensor], h: Tensor, prefix: str = "vit.") -> Tensor:
    """X[L] -> logits: final LayerNorm on the CLS rows only (they alone reach the classifier), Linear-Tanh-Linear; the
    last Linear reads fp32 weights (vg_head_fc2_kernel)."""
    hc = stored(_ln(h[:, 0, :], state[prefix + "norm.weight"], state[prefix + "norm.bias"]))
    t = _TanhStore.apply(hc @ shadow(state[prefix + "classifier.fc1.weight"]).t() + state[prefix + "classifier.fc1.bias"])
    return t @ state[prefix + "classifier.fc2.weight"].t() + state[prefix + "classifier.fc2.bias"]


def vit_forward(state: Mapping[str, Tensor], x: Tensor, d: VitDims, prefix: str = "vit.",
                masks: Optional[Mapping] = None, fp8: bool = False) -> Tensor:
    masks = masks or {}
    h = vit_embed(state, x, d, prefix, masks.get("embed"))
    for i in range(d.layers):
        h = vit_block(state, h, d, f"{prefix}encoder.{i}.", masks.get(("attn", i)), masks.get(("mlp", i)), fp8)
    return vit_head(state, h, prefix)


# ---------------------------------------------------------------------------------------------------------------------
# v1 SLN / SIREN generator (mirrors gen_oracle.gen_forward; engine.hip vg_gen_forward / vg_gen_backward_stages)
# ---------------------------------------------------------------------------------------------------------------------
def _sln(state: Mapping[str, Tensor], base: str, h: Tensor, w: Tensor, taps: Optional[dict] = None) -> Tensor:
    ln = _ln(h, state[base + "layer_norm.weight"], state[base + "layer_norm.bias"])
    y = w * (state[base + "gamma"] * ln + state[base + "beta"])  # norm.hip: wm * (g_s * r + b_s)
    if taps is not None:  # tests: the terms whose sums are the scalar gradients, d gamma = sum(dy w ln), d beta = sum(dy w)
        y.retain_grad()
        taps[base] = (y, w.detach(), ln.detach())
    return y


def gen_mapping(state: Mapping[str, Tensor], z: Tensor, d: GenDims) -> Tensor:
    """z -> modulation vectors w [B, T, E] (bf16 in HBM; gradient summed in fp32 over its 2L+1 uses)."""
    w = _Mapping.apply(bf(z), shadow(state["mapping_mlp.model.0.0.weight"]), state["mapping_mlp.model.0.0.bias"])
    return w.view(z.shape[0], d.tokens, d.embed)


def gen_embedding(state: Mapping[str, Tensor], B: int, d: GenDims):
    """The two reads of the learned embedding by block 0: (SLN input from the bf16 shadow, residual from the fp32 master)."""
    return _EmbeddingUses.apply(state["embedding"].expand(B, d.tokens, d.embed))


def gen_block(state: Mapping[str, Tensor], b: str, h_sln: Tensor, h_res: Tensor, w: Tensor, d: GenDims,
              m_attn: Optional[Tensor] = None, m_mlp: Optional[Tensor] = None, taps: Optional[dict] = None) -> Tensor:
    """h -> h_out of one TransformerSLN block; for blocks > 0 pass the same stored tensor as h_sln and h_res."""
    B, T, E = w.shape
    H, hd = d.heads, d.head_dim
    s1 = stored(_sln(state, b + "layer_norm_1.", h_sln, w, taps))
    wqkv = torch.cat([shadow(state[f"{b}msha.attention_heads.{hh}.{n}.weight"]) for n in ("q", "k", "v") for hh in range(H)], dim=0)
    qkv = stored(s1 @ wqkv.t())
    q, k, v = (t.reshape(B, T, H, hd).transpose(1, 2) for t in qkv.split(E, dim=-1))
    scale = 1.0 / math.sqrt(float(E))  # softmax(q.k / sqrt(H*hd)), src/v1/attention.py:51,90
    cat = stored(_Attention.apply(q, k, v, scale).transpose(1, 2).reshape(B, T, E))
    lin = cat @ shadow(state[b + "msha.output_linear.weight"]).t() + state[b + "msha.output_linear.bias"]
    htmp = stored(drop(lin, m_attn) + h_res)
    s2 = stored(_sln(state, b + "layer_norm_2.", htmp, w, taps))
    lin = s2 @ shadow(state[b + "mlp.model.0.0.weight"]).t() + state[b + "mlp.model.0.0.bias"]
    return stored(drop(lin, m_mlp) + htmp)


def gen_head(state: Mapping[str, Tensor], h: Tensor, w: Tensor, d: GenDims, pos_table: Optional[Tensor] = None,
             taps: Optional[dict] = None) -> Tensor:
    """h_L -> token rows [B, T, CW]: final SLN (+ optional position table) and the two SIREN layers."""
    sf = stored(_sln(state, "sln.", h, w, taps))
    if pos_table is not None:
        sf = stored(sf + pos_table)  # vg_add_table rewrites the bf16 tensor in place
    y = _SirenStore.apply(sf @ shadow(state["output_network.0.linear.weight"]).t() + state["output_network.0.linear.bias"], d.omega0)
    return _SirenStore.apply(y @ shadow(state["output_network.1.linear.weight"]).t() + state["output_network.1.linear.bias"], d.omega0)


def rows_to_image(y: Tensor, d: GenDims) -> Tensor:
    B = y.shape[0]
    if d.patch:
        g, P = d.image // d.patch, d.patch
        return y.view(B, g, g, d.channels, P, P).permute(0, 3, 1, 4, 2, 5).reshape(B, d.channels, d.image, d.image)
    return y.view(B, d.channels, d.image, d.image)


def gen_forward(state: Mapping[str, Tensor], z: Tensor, d: GenDims, masks: Optional[Mapping] = None,
                pos_table: Optional[Tensor] = None) -> Tensor:
    masks = masks or {}
    w = gen_mapping(state, z, d)
    h_sln, h_res = gen_embedding(state, z.shape[0], d)
    for i in range(d.layers):
        h_sln = h_res = gen_block(state, f"transformer_layers.{i}.", h_sln, h_res, w, d, masks.get(("attn", i)), masks.get(("mlp", i)))
    return rows_to_image(gen_head(state, h_sln, w, d, pos_table), d)
