"""The reference's v1 generator (self-modulated LayerNorm blocks + SIREN output) on the HIP engine.

``SirenGenerator`` has the parameter tree - hence the ``state_dict`` keys and shapes - of
``src.v1.generator.Generator`` (src/v1/generator.py:12-55): mapping Linear(Z -> T*E), learned
embedding [T,E], L x TransformerSLN (two SLN, H bias-free per-head q/k/v, output Linear, single-Linear
MLP), final SLN, two SIREN layers; output ``[B, C, IH, IW]`` is the flat view of ``[B, T, C*IW]``.
Forward/backward are ONE C call each into the fused engine.

Dropout: the reference block applies Dropout(0.2) to the attention output and inside the MLP
(src/v1/config.py:36,39) in train mode; here both are fused into the GEMM epilogues (counter-based
mask, p quantised to 1/256) and are active in ``train()`` mode, identity in ``eval()`` - parity with the
reference is defined in eval mode, RNG streams cannot match.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib, flat
from .flatparams import FlatParams


class _Holder(nn.Module):
    """Parameter container: exists only to give parameters their reference names."""


def _linear_holder(out_f, in_f, bias=True):
    h = _Holder()
    h.weight = nn.Parameter(torch.empty(out_f, in_f))
    if bias:
        h.bias = nn.Parameter(torch.empty(out_f))
    return h


def _mlp_holder(out_f, in_f):  # MLP(...).model = ModuleList([Sequential(Linear, Dropout)])
    m = _Holder()
    m.model = nn.ModuleList([nn.Sequential(_linear_holder(out_f, in_f), nn.Identity())])
    return m


def _sln_holder(E):
    s = _Holder()
    s.layer_norm = nn.LayerNorm(E)
    s.beta = nn.Parameter(torch.randn(1, 1, 1))
    s.gamma = nn.Parameter(torch.randn(1, 1, 1))
    return s


class _GenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: "SirenGenerator", z, anchor, drop_p):
        fp = mod._flat
        fp.refresh_shadow()
        ctx.drop = (float(drop_p), int(torch.randint(0, 2 ** 62, (1,)).item()) if drop_p > 0 else 0)
        B = z.shape[0]
        zin = z.detach().float().contiguous()
        ws = torch.empty(mod._ws_bytes(B), dtype=torch.uint8, device=z.device)
        img = torch.empty(B, mod.channels, mod.image_size, mod.image_size, dtype=torch.bfloat16, device=z.device)
        net = mod._net(ctx.drop)
        _lib.check(_lib.lib().vg_gen_forward(C.byref(net), B, zin.data_ptr(), ws.data_ptr(), img.data_ptr(),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vg_gen_forward")
        ctx.mod, ctx.ws, ctx.B = mod, ws, B
        return img.to(mod.out_dtype)

    @staticmethod
    def backward(ctx, dimg):
        mod = ctx.mod
        mod._flat.attach_grads()
        d = dimg.detach().to(torch.bfloat16).contiguous()
        net = mod._net(ctx.drop)
        _lib.check(_lib.lib().vg_gen_backward(C.byref(net), ctx.B, ctx.ws.data_ptr(), d.data_ptr(),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vg_gen_backward")
        ctx.ws = None
        return None, None, None, None


def fourier_position_table(T: int, E: int, image_size: int, patch_size: int) -> torch.Tensor:
    """[T, E] fp32: token t at (x, y) in [0,1)^2 (patch-grid cell centre, or (0.5, row centre) for the v1 row tokens);
    column 4*j + c = (sin, cos)(2 pi f_j x), (sin, cos)(2 pi f_j y), E/4 log-spaced frequencies from 1 to side/2."""
    side = image_size // patch_size if patch_size else T
    t = torch.arange(T, dtype=torch.float64)
    if patch_size:
        x, y = ((t % side) + 0.5) / side, (torch.div(t, side, rounding_mode="floor") + 0.5) / side
    else:
        x, y = torch.full((T,), 0.5, dtype=torch.float64), (t + 0.5) / side
    nb = E // 4
    f = torch.pow(torch.tensor(max(side / 2.0, 1.0), dtype=torch.float64), torch.arange(nb, dtype=torch.float64) / max(nb - 1, 1))
    ax, ay = 2 * math.pi * x[:, None] * f[None, :], 2 * math.pi * y[:, None] * f[None, :]
    return torch.stack([torch.sin(ax), torch.cos(ax), torch.sin(ay), torch.cos(ay)], dim=-1).reshape(T, E).float().contiguous()


class SirenGenerator(nn.Module):
    def __init__(self, latent=1024, image_size=32, channels=3, embed=384, heads=4, layers=4, siren_hidden=768,
                 omega_0=30.0, out_dtype=torch.float32, dropout=0.2, patch_size=0, fourier_features=False):
        """``patch_size == 0`` (default): the reference's v1 generator - one token per image row, each emitting
        ``channels * image_size`` values, assembled by a flat ``view`` (src/v1/generator.py:19,25,51,66-68).
        ``patch_size > 0`` (not in the reference): the same blocks on the discriminator's patch
        grid - ``(image_size / patch_size)**2`` tokens, each emitting one ``channels x P x P`` patch in conv1's
        (c, py, px) order, assembled by the un-patchify scatter.  Scales to 64x64 / 128x128 images (64 tokens)."""
        super().__init__()
        E, hd = embed, embed // heads
        if patch_size:
            if image_size % patch_size:
                raise ValueError("image_size must be a multiple of patch_size")
            T, out_features = (image_size // patch_size) ** 2, channels * patch_size * patch_size
        else:
            T, out_features = image_size, channels * image_size
        self.patch_size = int(patch_size)
        # optional Fourier positional input of the SIREN (north_star; not in the reference): a fixed [T, E] table added to
        # the final SLN output.  A non-persistent buffer: the state_dict stays the reference's.
        self.register_buffer("fourier_table", fourier_position_table(T, E, image_size, self.patch_size) if fourier_features else None,
                             persistent=False)
        self.latent, self.image_size, self.channels, self.out_dtype = latent, image_size, channels, out_dtype
        self.dropout_p = float(dropout)  # attention_dropout_rate = mlp_dropout = 0.2 in src/v1/config.py:36,39
        self.mapping_mlp = _mlp_holder(T * E, latent)
        self.embedding = nn.Parameter(torch.randn(T, E))
        blocks = []
        for _ in range(layers):
            b = _Holder()
            b.layer_norm_1 = _sln_holder(E)
            b.layer_norm_2 = _sln_holder(E)
            b.msha = _Holder()
            heads_l = []
            for _h in range(heads):
                a = _Holder()
                a.q, a.k, a.v = (_linear_holder(hd, E, bias=False) for _ in range(3))
                heads_l.append(a)
            b.msha.attention_heads = nn.ModuleList(heads_l)
            b.msha.output_linear = _linear_holder(E, E)
            b.mlp = _mlp_holder(E, E)
            blocks.append(b)
        self.transformer_layers = nn.ModuleList(blocks)
        self.sln = _sln_holder(E)
        s0, s1 = _Holder(), _Holder()
        s0.linear = _linear_holder(siren_hidden, E)
        s1.linear = _linear_holder(out_features, siren_hidden)
        self.output_network = nn.Sequential(s0, s1)
        self._dims = _lib.VgGenDims(latent, T, E, heads, layers, siren_hidden, out_features, float(omega_0),
                                    self.patch_size, channels, image_size)
        lay = flat.gen_layout(self._dims)
        self.reset_parameters()
        self._flat = FlatParams(dict(self.named_parameters()), flat.gen_slots(self._dims), lay.total)

    def reset_parameters(self):
        """Init distributions of the reference: nn.Linear default U(+-1/sqrt(in)); SIREN U(+-1/in) for
        the first layer and U(+-sqrt(6/in)/omega0) after (src/v1/siren.py:29-42); embedding, gamma,
        beta ~ N(0,1) (generator.py:24-26, spectral_layer_norm.py:16-17)."""
        w0 = float(self._dims.omega0)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name == "embedding" or name.endswith((".beta", ".gamma")):
                    p.normal_()
                elif "layer_norm.weight" in name:
                    p.fill_(1.0)
                elif "layer_norm.bias" in name:
                    p.zero_()
                elif name == "output_network.0.linear.weight":
                    p.uniform_(-1.0 / p.shape[1], 1.0 / p.shape[1])
                elif name == "output_network.1.linear.weight":
                    b = math.sqrt(6.0 / p.shape[1]) / w0
                    p.uniform_(-b, b)
                elif name.endswith("weight"):
                    b = 1.0 / math.sqrt(p.shape[1])
                    p.uniform_(-b, b)
                else:  # Linear bias: bound from the fan-in of its weight
                    fan_in = dict(self.named_parameters())[name[:-4] + "weight"].shape[1]
                    b = 1.0 / math.sqrt(fan_in)
                    p.uniform_(-b, b)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if hasattr(self, "_flat"):
            self._flat.named = dict(self.named_parameters())
            self._flat.rebuild()
        return out

    def zero_grad(self, set_to_none: bool = True):
        self._flat.zero_grad()

    def _ws_bytes(self, B):
        n = _lib.lib().vg_gen_ws_bytes(C.byref(self._dims), B)
        if n <= 0:
            raise RuntimeError("vg_gen_ws_bytes failed")
        return n

    def _net(self, drop=(0.0, 0)):
        fp = self._flat
        tab = self.fourier_table
        return _lib.VgGenNet(self._dims, fp.flat.data_ptr(), fp.shadow.data_ptr(), fp.grad.data_ptr(), float(drop[0]), int(drop[1]), None,
                             None if tab is None else tab.data_ptr())

    def forward(self, z):
        if not z.is_cuda:
            raise RuntimeError("SirenGenerator.forward: the HIP engine needs cuda tensors; there is no CPU fallback")
        if not self._flat.aliased():
            self._flat.named = dict(self.named_parameters())
            self._flat.rebuild()
        return _GenFn.apply(self, z, self.embedding, self.dropout_p if self.training else 0.0)
