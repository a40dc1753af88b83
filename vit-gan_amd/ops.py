"""torch.autograd.Functions over the single-operator C ABI (bf16 compute, fp32 accumulate).

These give every reference block (EmbedLayer, SelfAttention, Encoder, Classifier) a standalone
HIP path with autograd.  The whole-network passes in ``modules.py`` bypass them (one C call per
forward/backward); they are used when a block is called on its own, or when dropout is active.
There is no CPU path: a non-cuda tensor raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib

BF = torch.bfloat16


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: the HIP path needs cuda tensors (got {t.device}); there is no CPU fallback")


def _bf(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(BF).contiguous()


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _pad_rows(w: torch.Tensor, mult: int = 8) -> torch.Tensor:
    n = w.shape[0]
    if n % mult == 0:
        return w
    out = torch.zeros((n + mult - n % mult,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    out[:n] = w
    return out


def _wgrad(dy_b, x_b, M, N, K):
    """dW[N,K] = dy^T x via the split-K MFMA kernel (deterministic)."""
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    splits = max(1, min(8, 512 // max(tiles, 1), max(1, (M // 64) // 4)))
    dW = torch.empty(N, K, dtype=torch.float32, device=dy_b.device)
    slab = torch.empty(splits * N * K, dtype=torch.float32, device=dy_b.device)
    _lib.check(_lib.lib().vg_linear_wgrad(_p(dy_b), _p(x_b), _p(dW), _p(slab), slab.numel(), M, N, K, splits, 0, _st()), "vg_linear_wgrad")
    return dW


def _bias_grad(dy_b, M, N):
    parts = _lib.lib().vg_colsum_bf16_parts(M)
    ws = torch.empty(parts * N, dtype=torch.float32, device=dy_b.device)
    db = torch.empty(N, dtype=torch.float32, device=dy_b.device)
    _lib.check(_lib.lib().vg_colsum_bf16(_p(dy_b), N, M, N, _p(ws), _p(db), 0, _st()), "vg_colsum_bf16")
    return db


class LinearFn(torch.autograd.Function):
    """y = x W^T + b (+ res).  F.linear of src/v2/modules.py:128-139,161."""

    @staticmethod
    def forward(ctx, x, weight, bias, res):
        _need_cuda(x, "linear")
        K0 = x.shape[-1]
        N0 = weight.shape[0]
        xb = _bf(x).reshape(-1, K0)
        M = xb.shape[0]
        wb = _pad_rows(_bf(weight))
        if K0 % 8:  # zero-pad the reduction dim (e.g. Linear(classes_count=10, ...))
            xb = _pad_rows(xb.t().contiguous()).t().contiguous()
            wb = _pad_rows(wb.t().contiguous()).t().contiguous()
        K = xb.shape[1]
        N = wb.shape[0]
        bb = None if bias is None else _pad_rows(_f32(bias))
        rb = None if res is None else _bf(res).reshape(M, N0)
        if rb is not None and N != N0:
            raise RuntimeError("residual with a padded output is not supported")
        y = torch.empty(M, N, dtype=BF, device=x.device)
        _lib.check(_lib.lib().vg_linear_fwd(_p(xb), _p(wb), _p(bb), _p(rb), _p(y), None, None, M, N, K, 0, 0.0, _st()), "vg_linear_fwd")
        ctx.save_for_backward(xb, wb)
        ctx.dims = (M, N, K, N0, K0, bias is not None, res is not None, x.shape, x.dtype)
        return y[:, :N0].reshape(x.shape[:-1] + (N0,)).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        M, N, K, N0, K0, has_b, has_r, xshape, xdtype = ctx.dims
        dyb = _bf(dy).reshape(M, N0)
        if N != N0:
            t = torch.zeros(M, N, dtype=BF, device=dy.device)
            t[:, :N0] = dyb
            dyb = t
        dx = torch.empty(M, K, dtype=BF, device=dy.device)
        _lib.check(_lib.lib().vg_linear_dgrad(_p(dyb), _p(wb), _p(dx), M, N, K, 0, None, None, 0.0, _st()), "vg_linear_dgrad")
        dW = _wgrad(dyb, xb, M, N, K)[:N0, :K0]
        db = _bias_grad(dyb, M, N)[:N0] if has_b else None
        dres = dy if has_r else None
        return dx[:, :K0].reshape(xshape).to(xdtype), dW, db, dres


class MlpFn(torch.autograd.Function):
    """y = act(x W1^T + b1) W2^T + b2 with act in {gelu, tanh}: fc1/GELU/fc2 of the encoder block
    (src/v2/modules.py:173-182) and Linear/Tanh/Linear of the classifier (:196-198)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act):
        _need_cuda(x, "mlp")
        K = x.shape[-1]
        Hd, N0 = w1.shape[0], w2.shape[0]
        xb = _bf(x).reshape(-1, K)
        M = xb.shape[0]
        w1b, w2b = _bf(w1), _pad_rows(_bf(w2))
        N = w2b.shape[0]
        b1f, b2f = _f32(b1), _pad_rows(_f32(b2))
        a = torch.empty(M, Hd, dtype=BF, device=x.device)
        z = torch.empty(M, Hd, dtype=BF, device=x.device) if act == 1 else None
        y = torch.empty(M, N, dtype=BF, device=x.device)
        L = _lib.lib()
        _lib.check(L.vg_linear_fwd(_p(xb), _p(w1b), _p(b1f), None, _p(a), _p(z), None, M, Hd, K, act, 0.0, _st()), "vg_linear_fwd")
        _lib.check(L.vg_linear_fwd(_p(a), _p(w2b), _p(b2f), None, _p(y), None, None, M, N, Hd, 0, 0.0, _st()), "vg_linear_fwd")
        ctx.save_for_backward(xb, w1b, w2b, a, z if z is not None else a)
        ctx.dims = (M, K, Hd, N, N0, act, x.shape, x.dtype)
        return y[:, :N0].reshape(x.shape[:-1] + (N0,)).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xb, w1b, w2b, a, z = ctx.saved_tensors
        M, K, Hd, N, N0, act, xshape, xdtype = ctx.dims
        L = _lib.lib()
        dyb = _bf(dy).reshape(M, N0)
        if N != N0:
            t = torch.zeros(M, N, dtype=BF, device=dy.device)
            t[:, :N0] = dyb
            dyb = t
        dz = torch.empty(M, Hd, dtype=BF, device=dy.device)
        mode = 4 if act == 1 else 6  # gelu'(z) / 1 - tanh^2
        _lib.check(L.vg_linear_dgrad(_p(dyb), _p(w2b), _p(dz), M, N, Hd, mode, _p(z), None, 0.0, _st()), "vg_linear_dgrad")
        dx = torch.empty(M, K, dtype=BF, device=dy.device)
        _lib.check(L.vg_linear_dgrad(_p(dz), _p(w1b), _p(dx), M, Hd, K, 0, None, None, 0.0, _st()), "vg_linear_dgrad")
        dW2 = _wgrad(dyb, a, M, N, Hd)[:N0]
        db2 = _bias_grad(dyb, M, N)[:N0]
        dW1 = _wgrad(dz, xb, M, Hd, K)
        db1 = _bias_grad(dz, M, Hd)
        return dx.reshape(xshape).to(xdtype), dW1, db1, dW2, db2, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm(E), eps 1e-5 (src/v2/modules.py:168,172,225)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _need_cuda(x, "layernorm")
        E = x.shape[-1]
        xb = _bf(x).reshape(-1, E)
        R = xb.shape[0]
        g, b = _f32(gamma), _f32(beta)
        y = torch.empty(R, E, dtype=BF, device=x.device)
        mean = torch.empty(R, dtype=torch.float32, device=x.device)
        rstd = torch.empty(R, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().vg_layernorm_fwd(_p(xb), E, _p(g), _p(b), _p(y), E, _p(mean), _p(rstd), R, E, eps, _st()), "vg_layernorm_fwd")
        ctx.save_for_backward(xb, g, mean, rstd)
        ctx.dims = (R, E, x.shape, x.dtype)
        return y.reshape(x.shape).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xb, g, mean, rstd = ctx.saved_tensors
        R, E, xshape, xdtype = ctx.dims
        L = _lib.lib()
        dyb = _bf(dy).reshape(R, E)
        parts = L.vg_layernorm_bwd_parts(R)
        part = torch.empty(parts, 3 * E, dtype=torch.float32, device=dy.device)
        dx = torch.empty(R, E, dtype=BF, device=dy.device)
        _lib.check(L.vg_layernorm_bwd(_p(dyb), _p(xb), _p(mean), _p(rstd), _p(g), None, _p(dx), _p(part), R, E, _st()), "vg_layernorm_bwd")
        dg = torch.empty(E, dtype=torch.float32, device=dy.device)
        db = torch.empty(E, dtype=torch.float32, device=dy.device)
        _lib.check(L.vg_colsum_f32(_p(part), parts, 3 * E, _p(dg), E, _p(db), E, None, E, None, 0, 0, _st()), "vg_colsum_f32")
        return dx.reshape(xshape).to(xdtype), dg, db, None


class AttentionFn(torch.autograd.Function):
    """softmax(scale * score(q, k)) v over heads; qkv [B,S,3E] (Q|K|V thirds, head-major) -> [B,S,E].
    lp = 1: score = q k^T (src/v2/modules.py:142-159, v1 attention.py:69-70); lp = 2: score = cdist(q, k), the v1
    discriminator's L2 attention (attention.py:66-67)."""

    @staticmethod
    def forward(ctx, qkv, heads, scale, lp=1):
        _need_cuda(qkv, "attention")
        if lp not in (1, 2):
            raise ValueError(f"Unsupported norm for attention: lp={lp} but should be 1 or 2")
        B, S, E3 = qkv.shape
        E = E3 // 3
        HE = E // heads
        qb = _bf(qkv).reshape(B * S, E3)
        out = torch.empty(B * S, E, dtype=BF, device=qkv.device)
        lse = torch.empty(B, heads, S, dtype=torch.float32, device=qkv.device)
        L = _lib.lib()
        fn = L.vg_attention_fwd if lp == 1 else L.vg_attention_l2_fwd
        _lib.check(fn(_p(qb), _p(out), _p(lse), B, heads, S, HE, scale, _st()), "vg_attention_fwd")
        ctx.save_for_backward(qb, out, lse)
        ctx.dims = (B, heads, S, HE, scale, qkv.dtype, lp)
        return out.reshape(B, S, E).to(qkv.dtype)

    @staticmethod
    def backward(ctx, dout):
        qb, out, lse = ctx.saved_tensors
        B, H, S, HE, scale, dt, lp = ctx.dims
        dob = _bf(dout).reshape(B * S, H * HE)
        dqkv = torch.empty_like(qb)
        L = _lib.lib()
        fn = L.vg_attention_bwd if lp == 1 else L.vg_attention_l2_bwd
        _lib.check(fn(_p(qb), _p(out), _p(dob), _p(lse), _p(dqkv), B, H, S, HE, scale, _st()), "vg_attention_bwd")
        return dqkv.reshape(B, S, 3 * H * HE).to(dt), None, None, None


class UnfoldTokensFn(torch.autograd.Function):
    """v1 overlapping-window tokeniser (src/v1/patch_encoder.py:54-73): [B,C,IH,IH] -> [B, n*n, C*W*W], the reference's
    flat view of the double unfold; backward gathers every covering window per pixel."""

    @staticmethod
    def forward(ctx, images, patch, overlap):
        _need_cuda(images, "unfold_tokens")
        B, C, IH, IW = images.shape
        assert IH == IW, "The provided images are not square shaped"
        W = patch + 2 * overlap
        stride = (IH - patch - 2 * overlap) // patch + 1
        n = (IH - (W - 1) - 1) // stride + 1
        is_bf = images.dtype == BF
        src = images.contiguous() if is_bf else images.float().contiguous()
        out = torch.empty(B, n * n, C * W * W, dtype=BF, device=images.device)
        _lib.check(_lib.lib().vg_unfold_tokens_fwd(_p(src), int(is_bf), _p(out), B, C, IH, patch, overlap, _st()), "vg_unfold_tokens_fwd")
        ctx.geo = (B, C, IH, patch, overlap, images.dtype)
        return out.to(images.dtype)

    @staticmethod
    def backward(ctx, dtok):
        B, C, IH, patch, overlap, dt = ctx.geo
        d = _bf(dtok).contiguous()
        dimg = torch.empty(B, C, IH, IH, dtype=BF, device=dtok.device)
        _lib.check(_lib.lib().vg_unfold_tokens_bwd(_p(d), _p(dimg), B, C, IH, patch, overlap, _st()), "vg_unfold_tokens_bwd")
        return dimg.to(dt), None, None


def unfold_tokens(images, patch: int, overlap: int):
    return UnfoldTokensFn.apply(images, patch, overlap)


def linear(x, weight, bias=None, res=None):
    return LinearFn.apply(x, weight, bias, res)


def mlp(x, w1, b1, w2, b2, act: str):
    return MlpFn.apply(x, w1, b1, w2, b2, {"gelu": 1, "tanh": 3}[act])


def layer_norm(x, gamma, beta, eps: float = 1e-5):
    return LayerNormFn.apply(x, gamma, beta, eps)


def attention(qkv, heads: int, scale: Optional[float] = None, lp: int = 1):
    if scale is None:
        scale = 1.0 / math.sqrt(qkv.shape[-1] // 3 // heads)
    return AttentionFn.apply(qkv, heads, scale, lp)
