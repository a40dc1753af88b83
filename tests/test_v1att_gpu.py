"""v1 L2-distance attention on the GPU: C-ABI kernels and the module vs the pinned oracle."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("B,H,S,HE", [(3, 4, 65, 96), (2, 8, 65, 64), (4, 4, 17, 32), (2, 4, 32, 96), (1, 2, 80, 64)])
def test_l2_attention_kernels_vs_torch(B, H, S, HE):
    """vg_attention_l2_fwd / _bwd against fp32 torch (cdist scores) on bf16-rounded inputs: bf16 outputs within
    2^-7 of max|ref| (the tolerance of every bf16 operator test), gradients within 2^-6 (cdist's 1/dist factor)."""
    import gpu_util as u
    E = H * HE
    g = torch.Generator().manual_seed(B * 1000 + S)
    qkv = (torch.randn(B, S, 3 * E, generator=g) * 0.7).to(torch.bfloat16)
    do = torch.randn(B, S, E, generator=g).to(torch.bfloat16)
    scale = 1.0 / math.sqrt(E)
    ref_in = qkv.float().requires_grad_(True)
    q, k, v = (t.reshape(B, S, H, HE).transpose(1, 2) for t in ref_in.split(E, dim=-1))
    p = torch.softmax(torch.cdist(q, k, p=2) * scale, dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(B, S, E)
    (ref * do.float()).sum().backward()
    qd, dod = u.dev(qkv.reshape(B * S, 3 * E)), u.dev(do.reshape(B * S, E))
    out = torch.empty(B * S, E, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B * H * S, device="cuda")
    dq = torch.empty_like(qd)
    u.call("vg_attention_l2_fwd", u.ptr(qd), u.ptr(out), u.ptr(lse), B, H, S, HE, C.c_float(scale), u.stream())
    u.call("vg_attention_l2_bwd", u.ptr(qd), u.ptr(out), u.ptr(dod), u.ptr(lse), u.ptr(dq), B, H, S, HE, C.c_float(scale), u.stream())
    u.sync()
    u.assert_close(out.reshape(B, S, E), ref, 2.0 ** -7, "l2 attention out")
    u.assert_close(dq.reshape(B, S, 3 * E), ref_in.grad, 2.0 ** -6, "l2 attention dqkv")


@pytest.mark.parametrize("name", ["l2", "l2spec", "l2e128"])
def test_v1_mhsa_module_vs_golden_and_oracle(name):
    """The drop-in module (reference names / state_dict keys) against the reference's own output (golden fixture)
    and the oracle's gradients, on the fixture's weights."""
    import vit_gan_amd  # noqa: F401
    from cases import V1ATT_CASES
    from weights import make_input, make_state
    from oracle import v1att_oracle as ao
    from vit_gan_amd.v1_attention import MultiHeadSelfAttention, TransformerParameters
    import gpu_util as u

    c = V1ATT_CASES[name]
    npz = np.load(os.path.join(GOLD, f"v1att_{name}.npz"))
    tp = TransformerParameters(number_of_heads=c["heads"], input_features=c["embed"], lp=2, spectral_scaling=c["spectral"])
    M = MultiHeadSelfAttention(tp, output_size=c["embed"], head_dimension=c["head_dim"])
    shapes = {k: tuple(v.shape) for k, v in M.state_dict().items()}
    assert list(shapes) == [str(s) for s in npz["param_names"]]
    st_np = make_state(shapes, c["seed"], "v1att")
    M.load_state_dict({k: torch.from_numpy(v) for k, v in st_np.items()}, strict=True)
    if c["spectral"]:
        for h, row in zip(M.attention_heads, npz["init_spectrum"]):
            h.init_spectrum = [float(v) for v in row]  # the reference module's construction-time spectrum
    M = M.cuda()
    x = torch.from_numpy(make_input((c["batch"], c["seq"], c["embed"]), c["seed"]))
    xd = x.cuda().requires_grad_(True)
    out = M(xd)
    u.assert_close(out, torch.from_numpy(npz["out"]), 2.0 ** -6, "module output vs the reference's")
    R = torch.from_numpy(make_input(tuple(out.shape), c["seed"] + 1))
    (out * R.cuda()).sum().backward()
    st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in st_np.items()}
    xo = x.clone().requires_grad_(True)
    oo, used = ao.mhsa_l2_forward(st, xo, c["heads"], c["head_dim"], npz["init_spectrum"] if c["spectral"] else None)
    (oo * R).sum().backward()
    u.assert_close(xd.grad, xo.grad, 2.0 ** -5, "dx")
    got = dict(M.named_parameters())
    for k in ("attention_heads.0.q.weight", "attention_heads.1.k.weight", "attention_heads.3.v.weight", "output_linear.weight", "output_linear.bias"):
        u.assert_close(got[k].grad, used[k].grad, 2.0 ** -5, f"grad {k}", floor=1e-4)


@pytest.mark.parametrize("B,C,IH,P,ov", [(2, 3, 32, 8, 2), (1, 3, 16, 4, 1), (2, 1, 28, 4, 0), (3, 3, 64, 8, 2)])
def test_overlapping_tokeniser_vs_oracle(B, C, IH, P, ov):
    """vg_unfold_tokens_fwd is a pure gather: bit-exact against the oracle on bf16 inputs; the adjoint sums at most
    ceil(W/stride)^2 bf16 values per pixel in fp32 and rounds once: 2^-7 of max|ref|."""
    import vit_gan_amd  # noqa: F401
    from oracle import v1att_oracle as ao
    from vit_gan_amd import ops
    import gpu_util as u
    g = torch.Generator().manual_seed(IH + P)
    x = (torch.rand(B, C, IH, IH, generator=g) * 2 - 1).to(torch.bfloat16)
    ref_in = x.float().requires_grad_(True)
    ref = ao.unfold_tokens(ref_in, P, ov)
    xd = x.cuda().requires_grad_(True)
    tok = ops.unfold_tokens(xd, P, ov)
    assert tok.shape == ref.shape and tok.dtype == torch.bfloat16
    assert torch.equal(tok.float().cpu(), ref.detach())
    R = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
    (ref * R.float()).sum().backward()
    (tok.float() * R.cuda().float()).sum().backward()
    u.assert_close(xd.grad, ref_in.grad, 2.0 ** -7, "d images")
    # fp32 images take the same path (rounded to bf16 on the way in)
    tok32 = ops.unfold_tokens(x.float().cuda(), P, ov)
    assert tok32.dtype == torch.float32 and torch.equal(tok32.cpu(), ref.detach())
