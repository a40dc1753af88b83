"""Fused dropout: mask statistics, and forward/backward of the fused passes against the fp32 oracle run
with the SAME masks (extracted through vg_dropout_apply, which shares the kernels' hash and indexing)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _mask(u, shape, p, seed, site):
    ones = torch.ones(shape, dtype=torch.bfloat16, device="cuda")
    out = torch.empty_like(ones)
    u.call("vg_dropout_apply", u.ptr(ones), u.ptr(out), ones.numel(), p, seed, site, None, u.stream())
    u.sync()
    return out.float().cpu()


def _exact(m, p):
    """vg_dropout_apply on bf16 ones returns bf16(keep-scale); the fused epilogues multiply by the fp32 scale
    256 / (256 - round(256 p)): rebuild that exact multiplier from the mask's support."""
    thr = round(p * 256)
    return (m > 0).float() * (256.0 / (256.0 - thr))


def test_mask_statistics_and_determinism():
    import gpu_util as u
    p, n = 0.1, (4096, 384)
    m = _mask(u, n, p, 7, 3)
    thr = round(p * 256)
    keep = (256 - thr) / 256
    vals = torch.unique(m)
    assert len(vals) == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / keep) < 4e-3  # bf16 of 256/230
    frac = float((m > 0).float().mean())
    assert abs(frac - keep) < 3e-3, frac
    assert torch.equal(m, _mask(u, n, p, 7, 3))
    assert not torch.equal(m, _mask(u, n, p, 7, 4)) and not torch.equal(m, _mask(u, n, p, 8, 3))
    # columns and rows are both unbiased
    assert float((m > 0).float().mean(0).std()) < 0.02 and float((m > 0).float().mean(1).std()) < 0.03


@pytest.mark.parametrize("B", [3, 32])  # 32: M = 2080 rows, the K = 384 Linears (dropout + residual epilogue included) run on csrc/gemm_wr.hip
def test_vit_train_mode_matches_oracle_with_same_masks(B):
    import gpu_util as u
    from cases import VIT_CASES
    from weights import make_input, make_state
    from oracle import vit_oracle as vo
    from vit_gan_amd import _lib, flat

    c = dict(VIT_CASES["c1"])
    d = vo.VitDims(layers=c["layers"], classes=1)
    st_np = make_state(vo.vit_param_shapes(d), c["seed"], "vit")
    x = torch.from_numpy(make_input((B, 3, 32, 32), c["seed"], "uniform"))
    p, seed = 0.1, 1234
    S, E = d.seq, d.embed
    masks = {"embed": _exact(_mask(u, (B, S, E), p, seed, 0), p)}
    for l in range(d.layers):
        masks[("attn", l)] = _exact(_mask(u, (B, S, E), p, seed, 1 + 2 * l), p)
        masks[("mlp", l)] = _exact(_mask(u, (B, S, E), p, seed, 2 + 2 * l), p)
    from oracle import bf16_model as bm
    st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in st_np.items()}
    xr = x.clone().requires_grad_(True)
    out = vo.vit_forward(st, xr, d, masks=masks)
    R = torch.from_numpy(make_input(tuple(out.shape), c["seed"] + 1))
    (out * R).sum().backward()
    st_t = {k: torch.from_numpy(v).requires_grad_(True) for k, v in st_np.items()}
    xt = x.clone().requires_grad_(True)
    out_t = bm.vit_forward(st_t, xt, d, masks=masks)   # second reference: the rounding-faithful model with the same masks (tight tier: test_blocks_gpu.py)
    (out_t * R).sum().backward()
    out_eval = vo.vit_forward(st, x, d).detach()
    assert float((out.detach() - out_eval).abs().max()) > 1e-2  # dropout really changes the result

    dd = flat.vit_dims_struct(3, 32, 4, 384, 4, d.layers, 2, 1)
    lay, slots = flat.vit_layout(dd), flat.vit_slots(dd)
    P = flat.pack(slots, lay.total, st_np, device="cuda")
    Pb, G = P.to(torch.bfloat16), torch.zeros_like(P)
    net = _lib.VgVitNet(dd, P.data_ptr(), Pb.data_ptr(), G.data_ptr(), p, seed, None, _lib.context())
    ws = torch.empty(_lib.lib().vg_vit_ws_bytes(C.byref(dd), B), dtype=torch.uint8, device="cuda")
    logits = torch.empty(B, 1, device="cuda")
    X, Rd = x.cuda(), R.cuda()
    u.call("vg_vit_forward", C.byref(net), B, u.ptr(X), 0, u.ptr(ws), u.ptr(logits), u.stream())
    dimg = torch.empty(B, 3, 32, 32, dtype=torch.bfloat16, device="cuda")
    u.call("vg_vit_backward", C.byref(net), B, u.ptr(ws), u.ptr(Rd), u.ptr(dimg), 1, u.stream())
    u.sync()
    # 13 dropout sites scale surviving activations by 256/230: bf16 rounding noise grows accordingly (2^-4 here, 2^-5 in eval)
    u.assert_close(logits, out_t, 2.0 ** -4, "logits (train mode) vs the rounding-faithful model")
    u.assert_close(dimg, xt.grad, 2.0 ** -4, "d_img (train mode) vs the rounding-faithful model")
    u.assert_close(logits, out, 2.0 ** -4, "logits (train mode)")
    u.assert_close(dimg, xr.grad, 2.0 ** -4, "d_img (train mode)")
    grads = flat.unpack(slots, G)
    for k, prm in st.items():
        if float(prm.grad.abs().max()) < 1e-6:
            continue
        u.assert_close(grads[k], st_t[k].grad, 2.0 ** -4, f"grad {k} (train mode) vs the rounding-faithful model")
        u.assert_close(grads[k], prm.grad, 2.0 ** -4, f"grad {k} (train mode)")
    # staged backward == one-shot backward, bit for bit
    G2 = torch.zeros_like(P)
    net2 = _lib.VgVitNet(dd, P.data_ptr(), Pb.data_ptr(), G2.data_ptr(), p, seed, None, None)
    u.call("vg_vit_forward", C.byref(net2), B, u.ptr(X), 0, u.ptr(ws), u.ptr(logits), u.stream())
    for a, b in ((0, 3), (3, 5), (5, d.layers + 2)):
        u.call("vg_vit_backward_stages", C.byref(net2), B, u.ptr(ws), u.ptr(Rd), u.ptr(dimg), 1, a, b, u.stream())
    u.sync()
    assert torch.equal(G, G2), "staged backward must equal the one-shot backward bitwise"


def test_generator_train_mode_matches_oracle_with_same_masks():
    import gpu_util as u
    from weights import make_input, make_state
    from oracle import gen_oracle as go
    from vit_gan_amd import _lib, flat

    d = go.GenDims(layers=2)
    B, p, seed = 2, 0.2, 77
    st_np = make_state(go.gen_param_shapes(d), 21, "gen")
    z = torch.from_numpy(make_input((B, d.latent), 21))
    masks = {}
    for l in range(d.layers):
        masks[("attn", l)] = _exact(_mask(u, (B, d.tokens, d.embed), p, seed, 100 + 2 * l), p)
        masks[("mlp", l)] = _exact(_mask(u, (B, d.tokens, d.embed), p, seed, 101 + 2 * l), p)
    from oracle import bf16_model as bm
    st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in st_np.items()}
    out = go.gen_forward(st, z, d, masks=masks)
    R = torch.from_numpy(make_input(tuple(out.shape), 22)).to(torch.bfloat16).float()
    (out * R).sum().backward()
    st_t = {k: torch.from_numpy(v).requires_grad_(True) for k, v in st_np.items()}
    out_t = bm.gen_forward(st_t, z, d, masks=masks)
    (out_t * R).sum().backward()
    gd = _lib.VgGenDims(d.latent, d.tokens, d.embed, d.heads, d.layers, d.siren_hidden, d.out_features, d.omega0)
    lay, slots = flat.gen_layout(gd), flat.gen_slots(gd)
    P = flat.pack(slots, lay.total, st_np, device="cuda")
    Pb, G = P.to(torch.bfloat16), torch.zeros_like(P)
    net = _lib.VgGenNet(gd, P.data_ptr(), Pb.data_ptr(), G.data_ptr(), p, seed, None)
    ws = torch.empty(_lib.lib().vg_gen_ws_bytes(C.byref(gd), B), dtype=torch.uint8, device="cuda")
    img = torch.empty(B, 3, 32, 32, dtype=torch.bfloat16, device="cuda")
    Zd, Rd = z.cuda(), R.to(torch.bfloat16).cuda()
    u.call("vg_gen_forward", C.byref(net), B, u.ptr(Zd), u.ptr(ws), u.ptr(img), u.stream())
    u.call("vg_gen_backward", C.byref(net), B, u.ptr(ws), u.ptr(Rd), u.stream())
    u.sync()
    u.assert_close(img, out_t, 0.08, "generated image (train mode) vs the rounding-faithful model")
    u.assert_close(img, out, 0.08, "generated image (train mode)")
    grads = flat.unpack(slots, G)
    for k, prm in st.items():
        tol = 0.35 if k.endswith(("gamma", "beta")) else 0.12
        u.assert_close(grads[k], st_t[k].grad, tol, f"grad {k} (train mode) vs the rounding-faithful model", floor=1e-4)
        u.assert_close(grads[k], prm.grad, tol, f"grad {k} (train mode)", floor=1e-4)


def test_module_train_mode_runs_fused_dropout():
    import vit_gan_amd  # noqa: F401
    from vit_gan_amd.config import Config
    from vit_gan_amd.generator import SirenGenerator
    from vit_gan_amd.modules import ViTDiscriminator
    torch.manual_seed(0)
    D = ViTDiscriminator(Config(embeddings_dimension=128, transformer_blocks_count=2)).cuda().train()  # p = 0.1
    x = torch.randn(4, 3, 32, 32, device="cuda")
    torch.manual_seed(5); y1 = D(x)
    torch.manual_seed(5); y2 = D(x)
    torch.manual_seed(6); y3 = D(x)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)  # seeded from torch's generator
    D.eval()
    assert not torch.equal(D(x), y1)
    D.train(); D(x).sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in D.parameters())
    G = SirenGenerator(layers=1).cuda().train()
    z = torch.randn(2, 1024, device="cuda")
    a = G(z); G.eval(); b = G(z)
    assert a.shape == (2, 3, 32, 32) and not torch.equal(a, b)
