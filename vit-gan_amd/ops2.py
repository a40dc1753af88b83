"""Twice-differentiable operator set over the C ABI, for the gradient penalty (src/v2/utils.py:124-144).

``ops.py`` gives every operator a backward; the penalty differentiates the discriminator's INPUT GRADIENT with respect to
its parameters, so the backward operators on the input-gradient path need a backward of their own.  Each operator here is a
pair of ``torch.autograd.Function``s:

  level 1   forward = the forward kernel;  backward = a level-2 Function (plus the parameter gradients, which the penalty's
            graph never differentiates, straight from the kernels);
  level 2   forward = the backward kernel (vg_linear_dgrad, vg_layernorm_bwd, vg_attention_bwd, vg_act_bwd);
            backward = its derivative: for Linear the GEMM family again (d(dY) = ddX W^T, dW = dY^T ddX), for LayerNorm,
            attention and the activations the second-order kernels of csrc/second_order.hip.

Tensors between operators are bf16-valued (what the kernels store), carried in the caller's dtype.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib
from .ops import BF, _bf, _bias_grad, _f32, _need_cuda, _p, _st, _wgrad

ACTS = {"gelu": 1, "tanh": 3}

# Set (by ``input_grad_only()``) while a backward pass is run for the INPUT gradient alone - the first backward of the gradient penalty,
# ``torch.autograd.grad(out, inputs=interpolated, create_graph=True)``: autograd calls every operator's backward on the path and throws
# the parameter gradients away (a third of the penalty's weight-gradient GEMMs and all of its first-level bias / LayerNorm folds).
_INPUT_GRAD_ONLY = False


class input_grad_only:
    """``with ops2.input_grad_only(): torch.autograd.grad(out, x, create_graph=True)`` - the operators' first-level backwards skip the
    parameter gradients nobody asked for (they return None for them, which autograd reads as zero)."""

    def __enter__(self):
        global _INPUT_GRAD_ONLY
        self._prev, _INPUT_GRAD_ONLY = _INPUT_GRAD_ONLY, True
        return self

    def __exit__(self, *exc):
        global _INPUT_GRAD_ONLY
        _INPUT_GRAD_ONLY = self._prev
        return False


# ---------------------------------------------------------------------------------------------------------------------
# Deferred, grouped weight gradients.  In the penalty's second backward every Linear of a block gets TWO weight-gradient
# contributions (dW = dY^T ddX from the backward of its input gradient, and the ordinary dW = dY^T X of its forward under the
# second-order upstream gradient), at different times: through autograd that is 8 single-problem split-K launches per block,
# each with its own slab fold and its own AccumulateGrad add (2.9 of the penalty's 9 ms at B = 256).  Inside
# ``deferred_weight_grads(flat_grad)`` a Linear that was given its ``slot`` in the flat gradient buffer hands autograd None for
# dW and queues (dY, X) instead; on exit the queue goes out as ONE grouped launch + ONE fold per block and contribution
# (vg_linear_wgrad_group), accumulating straight into the flat buffer the parameters' .grad are views of.
# ---------------------------------------------------------------------------------------------------------------------
_QUEUE = None


class WeightSlot:
    """Where a Linear's weight gradient lives: ``offset`` (elements) into the flat gradient buffer, inside the fold region
    [``region``, ``region + region_floats``) that the weights of its block tile exactly (flat.vit_slots: wqkv | wo | w1 | w2)."""
    __slots__ = ("offset", "region", "region_floats")

    def __init__(self, offset: int, region: int, region_floats: int):
        self.offset, self.region, self.region_floats = int(offset), int(region), int(region_floats)


class deferred_weight_grads:
    """``with ops2.deferred_weight_grads(flat.grad): loss.backward()`` - see above.  The parameters' .grad must be views of
    ``flat_grad`` (FlatParams.attach_grads): the queued gradients are accumulated into it, not returned to autograd."""

    def __init__(self, flat_grad: torch.Tensor):
        self.grad, self.items = flat_grad, []

    def __enter__(self):
        global _QUEUE
        if _QUEUE is not None:
            raise RuntimeError("deferred_weight_grads does not nest")
        _QUEUE = self
        return self

    def __exit__(self, exc_type, *exc):
        global _QUEUE
        _QUEUE = None
        if exc_type is None:
            self.flush()
        self.items = []
        return False

    def push(self, slot: WeightSlot, dyb: torch.Tensor, xb: torch.Tensor, M: int, N: int, K: int) -> None:
        self.items.append((slot, dyb, xb, M, N, K))

    def flush(self) -> None:
        L, st, dev = _lib.lib(), _st(), self.grad.device
        # contribution index of an item = how many earlier items went to the same weight
        seen, sets = {}, {}
        for it in self.items:
            c = seen.get(it[0].offset, 0)
            seen[it[0].offset] = c + 1
            sets.setdefault((it[0].region, it[0].region_floats, c, it[3]), []).append(it)
        slab = None
        for (region, rf, _c, M), its in sets.items():
            tiles = sum(((N + 127) // 128) * max(1, K // 384) for _, _, _, _, N, K in its)
            covered = sum(N * K for _, _, _, _, N, K in its) == rf and len(its) <= 8
            if covered:
                splits = max(1, min(12, 256 // max(tiles, 1), M // 256))
                need = splits * rf
                if slab is None or slab.numel() < need:
                    slab = torch.empty(need, dtype=torch.float32, device=dev)
                n = len(its)
                dys = (C.c_void_p * n)(*[t[1].data_ptr() for t in its])
                xs = (C.c_void_p * n)(*[t[2].data_ptr() for t in its])
                Ns, Ks = (C.c_int * n)(*[t[4] for t in its]), (C.c_int * n)(*[t[5] for t in its])
                offs = (C.c_longlong * n)(*[t[0].offset - region for t in its])
                dst = C.c_void_p(self.grad.data_ptr() + 4 * region)
                rc = L.vg_linear_wgrad_group(n, dys, xs, Ns, Ks, offs, M, splits, _p(slab), slab.numel(), dst, rf, 1, st)
                if rc == 0:
                    continue
                if rc != -2:  # -2: the items do not tile the region (a frozen weight, a weight used twice): one by one below
                    _lib.check(rc, "vg_linear_wgrad_group")
            for slot, dyb, xb, Mi, N, K in its:
                splits = max(1, min(8, 512 // max(((N + 127) // 128) * ((K + 127) // 128), 1), max(1, (Mi // 64) // 4)))
                need = splits * N * K
                if slab is None or slab.numel() < need:
                    slab = torch.empty(need, dtype=torch.float32, device=dev)
                _lib.check(L.vg_linear_wgrad(_p(dyb), _p(xb), C.c_void_p(self.grad.data_ptr() + 4 * slot.offset), _p(slab), slab.numel(),
                                             Mi, N, K, splits, 1, st), "vg_linear_wgrad")


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def _pad_cols(t: torch.Tensor, n: int) -> torch.Tensor:
    if t.shape[1] == n:
        return t
    out = torch.zeros(t.shape[0], n, dtype=t.dtype, device=t.device)
    out[:, :t.shape[1]] = t
    return out


def _pad_rows_to(t: torch.Tensor, n: int) -> torch.Tensor:
    if t.shape[0] == n:
        return t
    out = torch.zeros((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    out[:t.shape[0]] = t
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Linear
# ---------------------------------------------------------------------------------------------------------------------
class _LinearDgrad(torch.autograd.Function):
    """dX[M,K] = dY[M,N] W[N,K]  (vg_linear_dgrad); its backward is two more GEMMs."""

    @staticmethod
    def forward(ctx, dy, weight, slot=None):
        M, N0 = dy.shape
        K = weight.shape[1]
        N = _pad8(N0)
        dyb = _pad_cols(_bf(dy), N)
        wb = _pad_rows_to(_bf(weight), N)
        dx = torch.empty(M, K, dtype=BF, device=dy.device)
        _lib.check(_lib.lib().vg_linear_dgrad(_p(dyb), _p(wb), _p(dx), M, N, K, 0, None, None, 0.0, _st()), "vg_linear_dgrad")
        ctx.save_for_backward(dyb, wb)
        ctx.dims = (M, N, N0, K, dy.dtype)
        ctx.slot = slot
        return dx.to(dy.dtype)

    @staticmethod
    def backward(ctx, ddx):
        dyb, wb = ctx.saved_tensors
        M, N, N0, K, dt = ctx.dims
        ub = _bf(ddx)
        d_dy = torch.empty(M, N, dtype=BF, device=ddx.device)   # d(dY) = ddX W^T: the forward Linear kernel
        _lib.check(_lib.lib().vg_linear_fwd(_p(ub), _p(wb), None, None, _p(d_dy), None, None, M, N, K, 0, 0.0, _st()), "vg_linear_fwd")
        if _QUEUE is not None and ctx.slot is not None and N == N0:
            _QUEUE.push(ctx.slot, dyb, ub, M, N, K)              # dW = dY^T ddX, grouped with the block's other weights on exit
            return d_dy.to(dt), None, None
        dW = _wgrad(dyb, ub, M, N, K)[:N0]                       # dW = dY^T ddX: the weight-gradient kernel
        return d_dy[:, :N0].to(dt), dW, None


class Linear2(torch.autograd.Function):
    """y = x W^T + b, twice differentiable along x and W."""

    @staticmethod
    def forward(ctx, x, weight, bias, slot=None):
        _need_cuda(x, "linear")
        K = x.shape[-1]
        N0 = weight.shape[0]
        if K % 8:
            raise RuntimeError("ops2.linear: the reduction dimension must be a multiple of 8")
        xb = _bf(x).reshape(-1, K)
        M, N = xb.shape[0], _pad8(N0)
        wb = _pad_rows_to(_bf(weight), N)
        bb = None if bias is None else _pad_rows_to(_f32(bias), N)
        y = torch.empty(M, N, dtype=BF, device=x.device)
        _lib.check(_lib.lib().vg_linear_fwd(_p(xb), _p(wb), _p(bb), None, _p(y), None, None, M, N, K, 0, 0.0, _st()), "vg_linear_fwd")
        ctx.save_for_backward(xb, weight)
        ctx.dims = (M, N, N0, K, bias is not None, x.shape, x.dtype)
        ctx.slot = slot
        return y[:, :N0].reshape(x.shape[:-1] + (N0,)).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xb, weight = ctx.saved_tensors
        M, N, N0, K, has_b, xshape, xdtype = ctx.dims
        dy2 = dy.reshape(M, N0)
        dx = _LinearDgrad.apply(dy2, weight, ctx.slot)          # differentiable
        if _INPUT_GRAD_ONLY:
            return dx.reshape(xshape).to(xdtype), None, None, None
        dyb = _pad_cols(_bf(dy2), N)
        if _QUEUE is not None and ctx.slot is not None and N == N0:
            _QUEUE.push(ctx.slot, dyb, xb, M, N, K)
            dW = None
        else:
            dW = _wgrad(dyb, xb, M, N, K)[:N0]
        db = _bias_grad(dyb, M, N)[:N0] if has_b else None
        return dx.reshape(xshape).to(xdtype), dW, db, None


# ---------------------------------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------------------------------
class _LayerNormBwd(torch.autograd.Function):
    """(dx, d gamma, d beta) = LN'(dy; x, gamma)  (vg_layernorm_bwd); backward = vg_layernorm_bwd_bwd."""

    @staticmethod
    def forward(ctx, dy, x, gamma, mean, rstd):
        R, E = x.shape
        L = _lib.lib()
        dyb, xb, g = _bf(dy), _bf(x), _f32(gamma)
        parts = L.vg_layernorm_bwd_parts(R)
        part = torch.empty(parts, 3 * E, dtype=torch.float32, device=dy.device)
        dx = torch.empty(R, E, dtype=BF, device=dy.device)
        _lib.check(L.vg_layernorm_bwd(_p(dyb), _p(xb), _p(mean), _p(rstd), _p(g), None, _p(dx), _p(part), R, E, _st()), "vg_layernorm_bwd")
        dg = torch.empty(E, dtype=torch.float32, device=dy.device)
        db = torch.empty(E, dtype=torch.float32, device=dy.device)
        if not _INPUT_GRAD_ONLY:  # (else: never read - LayerNorm2.backward drops them)
            _lib.check(L.vg_colsum_f32(_p(part), parts, 3 * E, _p(dg), E, _p(db), E, None, E, None, 0, 0, _st()), "vg_colsum_f32")
        ctx.save_for_backward(dyb, xb, g, mean, rstd)
        ctx.dims = (R, E, dy.dtype, x.dtype)
        ctx.mark_non_differentiable(dg, db)
        return dx.to(dy.dtype), dg, db

    @staticmethod
    def backward(ctx, u, _ug, _ub):
        dyb, xb, g, mean, rstd = ctx.saved_tensors
        R, E, dyt, xt = ctx.dims
        L = _lib.lib()
        ub = _bf(u)
        d_dy = torch.empty(R, E, dtype=BF, device=u.device)
        d_x = torch.empty(R, E, dtype=BF, device=u.device)
        parts = L.vg_layernorm_bwd_bwd_parts(R)
        part = torch.empty(parts, E, dtype=torch.float32, device=u.device)
        _lib.check(L.vg_layernorm_bwd_bwd(_p(ub), _p(dyb), _p(xb), _p(mean), _p(rstd), _p(g), _p(d_dy), _p(d_x), _p(part), R, E, _st()),
                   "vg_layernorm_bwd_bwd")
        d_g = torch.empty(E, dtype=torch.float32, device=u.device)
        _lib.check(L.vg_colsum_f32(_p(part), parts, E, _p(d_g), E, None, 0, None, 0, None, 0, 0, _st()), "vg_colsum_f32")
        return d_dy.to(dyt), d_x.to(xt), d_g, None, None


class LayerNorm2(torch.autograd.Function):
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
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.dims = (R, E)
        return y.reshape(x.shape).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        R, E = ctx.dims
        dx, dg, db = _LayerNormBwd.apply(dy.reshape(R, E), x.reshape(R, E), gamma, mean, rstd)
        if _INPUT_GRAD_ONLY:
            return dx.reshape(x.shape), None, None, None
        return dx.reshape(x.shape), dg, db, None


# ---------------------------------------------------------------------------------------------------------------------
# activations (GELU / tanh) on a stored pre-activation
# ---------------------------------------------------------------------------------------------------------------------
class _ActBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, h, act):
        dyb, hb = _bf(dy).reshape(-1), _bf(h).reshape(-1)
        dh = torch.empty_like(hb)
        _lib.check(_lib.lib().vg_act_bwd(_p(dyb), _p(hb), _p(dh), hb.numel(), act, _st()), "vg_act_bwd")
        ctx.save_for_backward(dyb, hb)
        ctx.meta = (act, dy.shape, dy.dtype, h.dtype)
        return dh.reshape(dy.shape).to(dy.dtype)

    @staticmethod
    def backward(ctx, u):
        dyb, hb = ctx.saved_tensors
        act, shape, dyt, ht = ctx.meta
        ub = _bf(u).reshape(-1)
        d_dy, d_h = torch.empty_like(hb), torch.empty_like(hb)
        _lib.check(_lib.lib().vg_act_bwd_bwd(_p(ub), _p(dyb), _p(hb), _p(d_dy), _p(d_h), hb.numel(), act, _st()), "vg_act_bwd_bwd")
        return d_dy.reshape(shape).to(dyt), d_h.reshape(shape).to(ht), None


class Act2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, act):
        _need_cuda(h, "activation")
        if h.numel() % 4:
            raise RuntimeError("ops2.act: element count must be a multiple of 4")
        hb = _bf(h).reshape(-1)
        y = torch.empty_like(hb)
        _lib.check(_lib.lib().vg_act_fwd(_p(hb), _p(y), hb.numel(), act, _st()), "vg_act_fwd")
        ctx.save_for_backward(h)
        ctx.act = act
        return y.reshape(h.shape).to(h.dtype)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        return _ActBwd.apply(dy, h, ctx.act), None


# ---------------------------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------------------------
class _AttentionBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dout, qkv, out, lse, heads, scale):
        B, S, E3 = qkv.shape
        E = E3 // 3
        qb, dob = _bf(qkv).reshape(B * S, E3), _bf(dout).reshape(B * S, E)
        dqkv = torch.empty_like(qb)
        _lib.check(_lib.lib().vg_attention_bwd(_p(qb), _p(out), _p(dob), _p(lse), _p(dqkv), B, heads, S, E // heads, scale, _st()),
                   "vg_attention_bwd")
        ctx.save_for_backward(qb, dob, lse)
        ctx.meta = (B, heads, S, E // heads, scale, dout.dtype, qkv.dtype)
        return dqkv.reshape(B, S, E3).to(qkv.dtype)

    @staticmethod
    def backward(ctx, u):
        qb, dob, lse = ctx.saved_tensors
        B, H, S, HE, scale, dot, qt = ctx.meta
        ub = _bf(u).reshape(B * S, 3 * H * HE)
        d_do = torch.empty_like(dob)
        d_q = torch.empty_like(qb)
        _lib.check(_lib.lib().vg_attention_bwd_bwd(_p(qb), _p(dob), _p(lse), _p(ub), _p(d_do), _p(d_q), B, H, S, HE, scale, _st()),
                   "vg_attention_bwd_bwd")
        return d_do.reshape(B, S, H * HE).to(dot), d_q.reshape(B, S, 3 * H * HE).to(qt), None, None, None, None


class Attention2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, heads, scale):
        _need_cuda(qkv, "attention")
        B, S, E3 = qkv.shape
        E = E3 // 3
        qb = _bf(qkv).reshape(B * S, E3)
        out = torch.empty(B * S, E, dtype=BF, device=qkv.device)
        lse = torch.empty(B, heads, S, dtype=torch.float32, device=qkv.device)
        _lib.check(_lib.lib().vg_attention_fwd(_p(qb), _p(out), _p(lse), B, heads, S, E // heads, scale, _st()), "vg_attention_fwd")
        ctx.save_for_backward(qkv, out, lse)
        ctx.meta = (heads, scale)
        return out.reshape(B, S, E).to(qkv.dtype)

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        heads, scale = ctx.meta
        return _AttentionBwd.apply(dout, qkv, out, lse, heads, scale), None, None


def linear(x, weight, bias=None, slot: Optional[WeightSlot] = None):
    """``slot``: where this weight's gradient lives in the flat gradient buffer - lets ``deferred_weight_grads`` group it."""
    return Linear2.apply(x, weight, bias, slot)


def layer_norm(x, gamma, beta, eps: float = 1e-5):
    return LayerNorm2.apply(x, gamma, beta, eps)


def act(h, kind: str):
    return Act2.apply(h, ACTS[kind])


def attention(qkv, heads: int, scale: Optional[float] = None):
    if scale is None:
        scale = 1.0 / math.sqrt(qkv.shape[-1] // 3 // heads)
    return Attention2.apply(qkv, heads, scale)
