"""The reference's ``src/v2/modules.py`` nn.Module surface on the MI355X HIP engine.

Same class names, constructor signatures, ``forward`` signatures and ``state_dict`` keys
(106 keys for ``ViTDiscriminator`` at 6 blocks) as the reference, so checkpoints written by
``src/v2/training.py:220-226,263`` load with ``strict=True``.  Compute is bf16 MFMA with fp32
accumulation in hand-written gfx950 kernels; there is no CPU path - a forward on CPU tensors
raises.  Citations are to /root/reference paths.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.nn as nn

from . import _lib, flat, ops, ops2
from .config import GENERATOR_KINDS, Config
from .flatparams import FlatParams
from .generator import SirenGenerator


class MovingAverage:  # src/v2/modules.py:9-21 (host-side scalar smoothing)
    def __init__(self, alpha=0.9):
        self.alpha = alpha
        self.value: Optional[float] = None

    def update(self, new_value: float):
        self.value = new_value if self.value is None else self.alpha * self.value + (1 - self.alpha) * new_value

    def get(self) -> float:
        return 0.0 if self.value is None else self.value


class EarlyStopping:  # src/v2/modules.py:24-45
    def __init__(self, patience=5, min_delta=2.0):
        self.patience, self.min_delta = patience, min_delta
        self.counter, self.best_score = 0, None

    def should_stop(self, current_score: float) -> bool:
        if self.best_score is None or current_score < self.best_score - self.min_delta:
            first = self.best_score is None
            self.best_score = current_score
            if not first:
                self.counter = 0
            return False
        self.counter += 1
        return self.counter >= self.patience


# --------------------------------------------------------------------------------------------
# building blocks (standalone HIP paths through ops.py)
# --------------------------------------------------------------------------------------------
class EmbedLayer(nn.Module):
    """src/v2/modules.py:67-100.  ``conv1`` only holds the [E,C,P,P] weight: a stride-P, kernel-P
    convolution is a per-patch GEMM, which is what runs."""

    def __init__(self, n_channels, embed_dim, image_size, patch_size, dropout=0.0):
        super().__init__()
        self.conv1 = nn.Conv2d(n_channels, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.pos_embedding = nn.Parameter(torch.zeros(1, (image_size // patch_size) ** 2, embed_dim), requires_grad=True)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim), requires_grad=True)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        B, Cc, IH, IW = x.shape
        P = self.conv1.kernel_size[0]
        E = self.conv1.out_channels
        gh, gw = IH // P, IW // P
        tiles = x.reshape(B, Cc, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, Cc * P * P)
        tok = ops.linear(tiles, self.conv1.weight.reshape(E, -1), self.conv1.bias)
        tok = tok + self.pos_embedding
        tok = torch.cat((self.cls_token.expand(B, 1, E).to(tok.dtype), tok), dim=1)
        return self.dropout(tok)


class SelfAttention(nn.Module):
    """src/v2/modules.py:103-162: three projections run as ONE [3E,E] GEMM, then the fused
    attention kernel, then the output projection."""

    def __init__(self, embed_dim, n_attention_heads):
        super().__init__()
        self.embed_dim = embed_dim
        self.n_attention_heads = n_attention_heads
        self.head_embed_dim = embed_dim // n_attention_heads
        width = self.head_embed_dim * self.n_attention_heads
        self.queries = nn.Linear(self.embed_dim, width)
        self.keys = nn.Linear(self.embed_dim, width)
        self.values = nn.Linear(self.embed_dim, width)
        self.out_projection = nn.Linear(width, self.embed_dim)

    def forward(self, x):
        w = torch.cat((self.queries.weight, self.keys.weight, self.values.weight), dim=0)
        b = torch.cat((self.queries.bias, self.keys.bias, self.values.bias), dim=0)
        qkv = ops.linear(x, w, b)
        ctx = ops.attention(qkv, self.n_attention_heads, 1.0 / float(self.head_embed_dim) ** 0.5)
        return ops.linear(ctx, self.out_projection.weight, self.out_projection.bias)


class Encoder(nn.Module):
    """Pre-LN transformer block, src/v2/modules.py:165-183."""

    def __init__(self, embed_dim, n_attention_heads, forward_mul, dropout=0.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dim)
        self.attention = SelfAttention(embed_dim, n_attention_heads)
        self.dropout1 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.fc1 = nn.Linear(embed_dim, embed_dim * forward_mul)
        self.activation = nn.GELU()
        self.fc2 = nn.Linear(embed_dim * forward_mul, embed_dim)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, x):
        h = ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x = x + self.dropout1(self.attention(h))
        h = ops.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        h = ops.mlp(h, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, "gelu")
        return x + self.dropout2(h)


class Classifier(nn.Module):
    """CLS row -> Linear -> Tanh -> Linear, src/v2/modules.py:186-199."""

    def __init__(self, embed_dim, n_classes):
        super().__init__()
        self.fc1 = nn.Linear(embed_dim, embed_dim)
        self.activation = nn.Tanh()
        self.fc2 = nn.Linear(embed_dim, n_classes)

    def forward(self, x):
        return ops.mlp(x[:, 0, :], self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, "tanh")


def vit_init_weights(m):
    """src/v2/modules.py:241-253 (usable with ``module.apply``)."""
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        nn.init.trunc_normal_(m.weight, mean=0.0, std=0.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.weight, 1)
        nn.init.constant_(m.bias, 0)
    elif isinstance(m, EmbedLayer):
        nn.init.trunc_normal_(m.cls_token, mean=0.0, std=0.02)
        nn.init.trunc_normal_(m.pos_embedding, mean=0.0, std=0.02)


# --------------------------------------------------------------------------------------------
# whole-network autograd node: ONE C call per forward and per backward
# --------------------------------------------------------------------------------------------
def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _fresh_seed() -> int:
    """Per-forward dropout seed drawn from torch's CPU generator (reproducible under torch.manual_seed)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class _VitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: "VisionTransformer", x, anchor, drop_p):
        fp = mod._flat
        fp.refresh_shadow()
        ctx.drop = (float(drop_p), _fresh_seed() if drop_p > 0 else 0)
        B = x.shape[0]
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        xin = x.detach().contiguous()
        ws = torch.empty(mod._ws_bytes(B), dtype=torch.uint8, device=x.device)
        logits = torch.empty(B, mod._dims.Kc, dtype=torch.float32, device=x.device)
        net = mod._net(need_grad=False, drop=ctx.drop)
        _lib.check(_lib.lib().vg_vit_forward(C.byref(net), B, xin.data_ptr(), int(xin.dtype == torch.bfloat16),
                                             ws.data_ptr(), logits.data_ptr(), _stream()), "vg_vit_forward")
        ctx.mod, ctx.ws, ctx.B, ctx.xdtype = mod, ws, B, x.dtype
        ctx.need_dx = x.requires_grad
        ctx.xshape = x.shape
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        mod = ctx.mod
        want_w = any(p.requires_grad for p in mod.parameters())
        if want_w:
            mod._flat.attach_grads()
        dl = dlogits.detach().float().contiguous()
        dimg = torch.empty(ctx.xshape, dtype=torch.bfloat16, device=dl.device) if ctx.need_dx else None
        net = mod._net(need_grad=want_w, drop=ctx.drop)
        _lib.check(_lib.lib().vg_vit_backward(C.byref(net), ctx.B, ctx.ws.data_ptr(), dl.data_ptr(),
                                              None if dimg is None else dimg.data_ptr(), int(want_w), _stream()),
                   "vg_vit_backward")
        ctx.ws = None
        return None, (None if dimg is None else dimg.to(ctx.xdtype)), None, None


class VisionTransformer(nn.Module):
    """src/v2/modules.py:202-238.  Parameters live in one flat buffer (flatparams.py); ``forward``
    is one fused pass of the HIP engine; in train mode with ``dropout > 0`` the three nn.Dropout sites of
    the reference (:80,:170,:176) are applied inside the GEMM epilogues with a counter-based mask (RNG-stream
    parity with torch is not defined; p is quantised to 1/256).  ``composed_forward`` runs the same network
    block by block through ops.py with torch's own dropout."""

    def __init__(self, n_channels, embed_dim, n_layers, n_attention_heads, forward_mul, image_size, patch_size,
                 n_classes, dropout=0.1):
        super().__init__()
        self.embedding = EmbedLayer(n_channels, embed_dim, image_size, patch_size, dropout=dropout)
        self.encoder = nn.ModuleList(
            [Encoder(embed_dim, n_attention_heads, forward_mul, dropout=dropout) for _ in range(n_layers)])
        self.norm = nn.LayerNorm(embed_dim)
        self.classifier = Classifier(embed_dim, n_classes)
        self.apply(vit_init_weights)
        self._dropout_p = float(dropout)
        # fp8 (e4m3) MFMA operands for the attention's Q.K^T and P.V (BASELINE.json's 128x128 configuration); off by default
        # - the reference's arithmetic is fp32 and the parity tiers are stated for bf16 storage.  Not a Config field: set
        # ``model.vit.attention_fp8 = True`` (GanEngine picks it up too).
        self.attention_fp8 = False
        self._dims = flat.vit_dims_struct(n_channels, image_size, patch_size, embed_dim, n_attention_heads, n_layers,
                                          forward_mul, n_classes)
        lay = flat.vit_layout(self._dims)  # raises for shapes the kernels do not cover
        object.__setattr__(self, "_flat", FlatParams(dict(self.named_parameters()), flat.vit_slots(self._dims, prefix=""),
                                                     lay.total))

    # -- storage plumbing --------------------------------------------------------------------
    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flat.named = dict(self.named_parameters())
        self._flat.rebuild()
        return out

    def zero_grad(self, set_to_none: bool = True):
        self._flat.zero_grad()

    def _ws_bytes(self, B: int) -> int:
        n = _lib.lib().vg_vit_ws_bytes(C.byref(self._dims), B)
        if n <= 0:
            raise RuntimeError("vg_vit_ws_bytes failed")
        return n

    def _net(self, need_grad: bool, drop=(0.0, 0)) -> _lib.VgVitNet:
        fp = self._flat
        return _lib.VgVitNet(self._dims, fp.flat.data_ptr(), fp.shadow.data_ptr(), fp.grad.data_ptr() if need_grad else None,
                             float(drop[0]), int(drop[1]), None, _lib.context() if need_grad else None, int(self.attention_fp8))

    # -- forward -----------------------------------------------------------------------------
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("VisionTransformer.forward: the HIP engine needs cuda tensors; there is no CPU fallback")
        if not self._flat.aliased():
            self._flat.named = dict(self.named_parameters())
            self._flat.rebuild()
        p = self._dropout_p if self.training else 0.0
        return _VitFn.apply(self, x, self.norm.weight, p)

    def composed_forward(self, x):
        """The same network through the per-operator HIP path (ops.py) and torch's nn.Dropout modules."""
        h = self.embedding(x)
        for block in self.encoder:
            h = block(h)
        # the classifier reads the CLS row only (:195), so normalising that row is equivalent to :236
        h = ops.layer_norm(h[:, :1, :], self.norm.weight, self.norm.bias, self.norm.eps)
        return self.classifier(h)


    def twice_differentiable_forward(self, x):
        """The same network through the twice-differentiable operator set (ops2.py): what ``gradient_penalty`` runs, since
        it differentiates the input gradient (``torch.autograd.grad(..., create_graph=True)``, src/v2/utils.py:132-139).
        Dropout, as in ``composed_forward``, is torch's own on the modules' nn.Dropout layers."""
        if not x.is_cuda:
            raise RuntimeError("VisionTransformer: the HIP engine needs cuda tensors; there is no CPU fallback")
        emb = self.embedding
        B, Cc, IH, IW = x.shape
        P, E = emb.conv1.kernel_size[0], emb.conv1.out_channels
        gh, gw = IH // P, IW // P
        # The operators exchange bf16-VALUED tensors in the caller's dtype; with an fp32 caller every operator boundary was a pair of
        # cast kernels over [B*S, E..4E] (8.7 ms of the 26 ms penalty step).  The pass therefore runs in bf16 from the patch tiles on:
        # the same values, the residual adds and autograd's gradient sums round once as before; only the input gradient is cast back.
        tiles = x.reshape(B, Cc, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, Cc * P * P).to(torch.bfloat16)
        tok = ops2.linear(tiles, emb.conv1.weight.reshape(E, -1), emb.conv1.bias) + emb.pos_embedding.to(torch.bfloat16)
        h = emb.dropout(torch.cat((emb.cls_token.expand(B, 1, E).to(tok.dtype), tok), dim=1))
        # where each block Linear's weight gradient lives in the flat buffer (q | k | v are contiguous there, and the four weights of a
        # block tile one region): inside ops2.deferred_weight_grads the penalty's weight gradients then go out grouped per block
        sl = self._flat.slots
        for i, blk in enumerate(self.encoder):
            a = blk.attention
            o_qkv, o_wo = sl[f"encoder.{i}.attention.queries.weight"][0], sl[f"encoder.{i}.attention.out_projection.weight"][0]
            o_w1, o_w2 = sl[f"encoder.{i}.fc1.weight"][0], sl[f"encoder.{i}.fc2.weight"][0]
            region = o_w2 + blk.fc2.weight.numel() - o_qkv
            slot = lambda off: ops2.WeightSlot(off, o_qkv, region)  # noqa: E731
            n1 = ops2.layer_norm(h, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
            w = torch.cat((a.queries.weight, a.keys.weight, a.values.weight), dim=0)
            b = torch.cat((a.queries.bias, a.keys.bias, a.values.bias), dim=0)
            ctx = ops2.attention(ops2.linear(n1, w, b, slot(o_qkv)), a.n_attention_heads, 1.0 / float(a.head_embed_dim) ** 0.5)
            if i == len(self.encoder) - 1:
                # top block: the classifier reads the CLS row only (:195), so behind the attention the row-local operators - and with
                # them their backward and double backward - run on the B CLS rows (the fused engine does the same, csrc/engine.hip)
                ctx, h = ctx[:, :1, :], h[:, :1, :]
            h = h + blk.dropout1(ops2.linear(ctx, a.out_projection.weight, a.out_projection.bias, slot(o_wo)))
            n2 = ops2.layer_norm(h, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            z = ops2.act(ops2.linear(n2, blk.fc1.weight, blk.fc1.bias, slot(o_w1)), "gelu")
            h = h + blk.dropout2(ops2.linear(z, blk.fc2.weight, blk.fc2.bias, slot(o_w2)))
        c = ops2.layer_norm(h[:, 0, :], self.norm.weight, self.norm.bias, self.norm.eps)   # only the CLS row reaches the classifier
        t = ops2.act(ops2.linear(c, self.classifier.fc1.weight, self.classifier.fc1.bias), "tanh")
        # Linear(E, classes_count): E multiply-adds per image and logit - the fused pass has a dedicated kernel for it; here
        # it is left to torch so that it is twice differentiable without a kernel of its own
        return torch.nn.functional.linear(t.float(), self.classifier.fc2.weight, self.classifier.fc2.bias)


# --------------------------------------------------------------------------------------------
# GAN wrappers
# --------------------------------------------------------------------------------------------
def _vit_from_config(config: Config) -> VisionTransformer:
    return VisionTransformer(
        n_channels=config.input_channels, embed_dim=config.embeddings_dimension,
        n_layers=config.transformer_blocks_count, n_attention_heads=config.attention_heads_count,
        forward_mul=config.mlp_ratio, image_size=config.image_size, patch_size=config.patch_size,
        n_classes=config.classes_count, dropout=config.dropout_rate)


class ViTGenerator(nn.Module):
    """src/v2/modules.py:344-372, including its tail: ``Linear(classes_count, batch_size)`` followed by
    a flat ``view(-1, C, IH, IW)`` which is only legal when B*batch_size is a multiple of C*IH*IW
     - reproduced faithfully, so it raises exactly where the reference raises.
    The working generator of this engine is ``vit_gan_amd.generator.SirenGenerator``."""

    def __init__(self, config: Config):
        super().__init__()
        self.vit = _vit_from_config(config)
        self.linear = nn.Linear(config.classes_count, config.batch_size)
        self.image_size = config.image_size
        self.input_channels = config.input_channels

    def zero_grad(self, set_to_none: bool = True):
        self.vit.zero_grad()
        for p in self.linear.parameters():
            p.grad = None

    def forward(self, x):
        x = self.vit(x)
        x = ops.linear(x, self.linear.weight, self.linear.bias)
        return x.view(-1, self.input_channels, self.image_size, self.image_size)


class ViTDiscriminator(nn.Module):
    """src/v2/modules.py:375-395."""

    def __init__(self, config: Config):
        super().__init__()
        self.vit = _vit_from_config(config)

    def zero_grad(self, set_to_none: bool = True):
        self.vit.zero_grad()

    def forward(self, x):
        return self.vit(x)


def generator_from_config(config: Config) -> nn.Module:
    """``config.generator_kind`` (the one extra Config field, default "v2" = reference behaviour) picks the generator."""
    kind = config.generator_kind
    if kind not in GENERATOR_KINDS:
        raise ValueError(f"generator_kind must be one of {GENERATOR_KINDS}, got {kind!r}")
    if kind == "v2":
        return ViTGenerator(config)
    if kind == "sln_siren":  # v1 defaults (src/v1/config.py:45-49,60-66): z 1024, E 384, 4 heads, 4 blocks, SIREN 768
        return SirenGenerator(image_size=config.image_size, channels=config.input_channels)
    return SirenGenerator(image_size=config.image_size, channels=config.input_channels, embed=config.embeddings_dimension,
                          heads=config.attention_heads_count, patch_size=config.patch_size)


class ViTGAN(nn.Module):
    """src/v2/modules.py:398-410.  ``Config(generator_kind="sln_siren")`` makes ``.generator`` the working SLN/SIREN
    network (latent input ``[B, generator.latent]``) instead of the reference's ViTGenerator, whose tail raises."""

    def __init__(self, config: Config):
        super().__init__()
        self.generator = generator_from_config(config)
        self.discriminator = ViTDiscriminator(config)

    def zero_grad(self, set_to_none: bool = True):
        self.generator.zero_grad()
        self.discriminator.zero_grad()

    def forward(self, z):
        generated_images = self.generator(z)
        discriminator_output = self.discriminator(generated_images)
        return generated_images, discriminator_output


# --------------------------------------------------------------------------------------------
# CNN toys of the reference (src/v2/modules.py:256-341,413-425): NOT on the hot path
# (training.py:145 builds ViTGAN).  Kept importable under their reference names as plain
# PyTorch modules; nothing here is accelerated.
# --------------------------------------------------------------------------------------------
def _cnn_stack(spec, final):
    layers = []
    for kind, cin, cout, norm, act in spec:
        conv = nn.Conv2d if kind == "down" else nn.ConvTranspose2d
        layers.append(conv(cin, cout, kernel_size=4, stride=2, padding=1, bias=False))
        if norm:
            layers.append(nn.BatchNorm2d(cout))
        if act is not None:
            layers.append(act())
    layers.extend(final)
    return nn.Sequential(*layers)


class Generator(nn.Module):
    def __init__(self, config: Config):
        super().__init__()
        c = config.input_channels
        relu = lambda: nn.ReLU(True)  # noqa: E731
        self.main = _cnn_stack([("down", c, 64, True, relu), ("down", 64, 128, True, relu), ("down", 128, 256, True, relu),
                                ("up", 256, 128, True, relu), ("up", 128, 64, True, relu), ("up", 64, c, False, None)],
                               [nn.Tanh()])

    def forward(self, input):
        return self.main(input)


class Discriminator(nn.Module):
    def __init__(self, config: Config):
        super().__init__()
        c = config.input_channels
        lrelu = lambda: nn.LeakyReLU(0.2, inplace=True)  # noqa: E731
        self.main = _cnn_stack([("down", c, 64, False, lrelu), ("down", 64, 128, True, lrelu), ("down", 128, 256, True, lrelu),
                                ("down", 256, 512, True, lrelu)],
                               [nn.Conv2d(512, 1, kernel_size=2, stride=1, padding=0, bias=False), nn.Sigmoid()])

    def forward(self, input):
        return self.main(input).view(-1, 1).squeeze(1)


class CNNGAN(nn.Module):
    def __init__(self, config: Config):
        super().__init__()
        self.generator = Generator(config)
        self.discriminator = Discriminator(config)

    def forward(self, z):
        generated_images = self.generator(z)
        return generated_images, self.discriminator(generated_images)


def load_pretrained_discriminator(vit_gan):
    """src/v2/modules.py:428-440 fetches torchvision's ViT-B/16 weights over the network and loads them
    with strict=False (no key matches); out of scope for the offline hot path."""
    raise NotImplementedError("load_pretrained_discriminator needs torchvision + network access (out of scope)")
