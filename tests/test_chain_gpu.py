"""Parity of the row-chain kernels (csrc/chain.hip, through the C ABI) against fp32 PyTorch.

vg_encoder_mlp_fwd replaces fc1 -> nn.GELU -> fc2 -> dropout -> residual add (src/v2/modules.py:181-182) and the LayerNorm
that reads the sum (the next block's norm1, :168): one launch, the hidden never read back.  Inputs are rounded to bf16 first;
the kernel rounds where the unfused kernels stored bf16 (a1, and the sum Y once), so the reference does the same: 2^-7 of
max|ref| for bf16 outputs, one code step (0.005) for the derivative bytes, 3e-5 for the statistics of the kernel's own Y.
Row counts: M = 16 * units, 8 units (one per wave) to a workgroup tile - partial tiles, several tiles per workgroup, and the
C2 launches (M = 16 640 / 33 280)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF_TOL = 2.0 ** -7
E, HID = 384, 768


def _u():
    import gpu_util
    return gpu_util


def _mask(u, M, p, seed, site):
    ones = torch.ones(M, E, dtype=u.BF, device="cuda")
    m = torch.empty_like(ones)
    u.call("vg_dropout_apply", u.ptr(ones), u.ptr(m), M * E, p, seed, site, None, u.stream())
    u.sync()
    return m.float().cpu()


def _gelu_grad(z):
    return 0.5 * (1.0 + torch.erf(z / math.sqrt(2.0))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)


def _pack_mlp(u, W1, W2):
    L = u._lib.lib()
    n = L.vg_encoder_mlp_image_elems()
    assert n == 2 * E * HID
    img = torch.empty(n, dtype=u.BF, device="cuda")
    d1, d2 = u.dev(W1, u.BF), u.dev(W2, u.BF)
    u.call("vg_encoder_mlp_pack", u.ptr(d1), u.ptr(d2), u.ptr(img), u.stream())
    u.sync()
    return img


MLP_SHAPES = [16, 48, 128, 144, 2080, 4160, 16640, 33280, 66560]


@pytest.mark.parametrize("M", MLP_SHAPES)
@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_encoder_mlp_fwd(M, drop):
    u = _u()
    g = torch.Generator().manual_seed(M + 7)
    xn = u.rbf(torch.randn(M, E, generator=g))
    W1 = u.rbf(torch.randn(HID, E, generator=g) / math.sqrt(E))
    b1 = torch.randn(HID, generator=g) * 0.1
    W2 = u.rbf(torch.randn(E, HID, generator=g) / math.sqrt(HID))
    b2 = torch.randn(E, generator=g) * 0.1
    R = u.rbf(torch.randn(M, E, generator=g))
    gam = 1.0 + 0.2 * torch.randn(E, generator=g)
    bet = 0.1 * torch.randn(E, generator=g)
    seed, site = 91, 4
    z = xn @ W1.t() + b1
    a1 = u.rbf(F.gelu(z))  # the hidden is stored (and multiplied) as bf16
    y = a1 @ W2.t() + b2
    if drop:
        y = y * _mask(u, M, drop, seed, site)
    y = u.rbf(y + R)
    yn = F.layer_norm(y, (E,), gam, bet, 1e-5)

    img = _pack_mlp(u, W1, W2)
    d = {k: u.dev(v, u.BF) for k, v in (("xn", xn), ("R", R))}
    db1, db2, dg, dbt = u.dev(b1), u.dev(b2), u.dev(gam), u.dev(bet)
    A1 = torch.full((M + 16, HID), 7.0, dtype=u.BF, device="cuda")
    Z8 = torch.full((M + 16, HID), 9, dtype=torch.uint8, device="cuda")
    Y = torch.full((M + 16, E), 7.0, dtype=u.BF, device="cuda")
    Yn = torch.full((M + 16, E), 7.0, dtype=u.BF, device="cuda")
    mean = torch.empty(M, device="cuda")
    rstd = torch.empty(M, device="cuda")
    u.call("vg_encoder_mlp_fwd", u.ptr(d["xn"]), u.ptr(img), u.ptr(db1), u.ptr(db2), u.ptr(d["R"]), u.ptr(A1), u.ptr(Z8), u.ptr(Y), u.ptr(Yn),
           u.ptr(mean), u.ptr(rstd), u.ptr(dg), u.ptr(dbt), M, 1e-5, drop, seed, site, None, u.stream())
    u.sync()
    u.assert_close(A1[:M], a1, BF_TOL, "a1")
    dz = (Z8[:M].float().cpu() - 27.0) * 0.005
    err = float((dz - _gelu_grad(z)).abs().max())
    assert err <= 0.0051, f"gelu' codes: max err {err:.4f}"
    # Y from the kernel's own hidden (one bf16 ulp of a1 apart from the reference's moves the sum by less than its own rounding)
    u.assert_close(Y[:M], y, BF_TOL * 1.5, "Y")
    yk = Y[:M].float().cpu()
    u.assert_close(Yn[:M], F.layer_norm(yk, (E,), gam, bet, 1e-5), BF_TOL, "Yn vs LN(own Y)")
    u.assert_close(mean, yk.mean(1), 3e-5, "mean", floor=1e-6)
    u.assert_close(rstd, 1.0 / torch.sqrt(yk.var(1, unbiased=False) + 1e-5), 3e-5, "rstd")
    u.assert_close(Yn[:M], yn, 2.0 ** -5, "Yn vs reference")
    assert bool((Y[M:] == 7.0).all()) and bool((Yn[M:] == 7.0).all()) and bool((A1[M:] == 7.0).all()) and bool((Z8[M:] == 9).all()), "rows beyond M were written"
    # bitwise repeatable, and without a LayerNorm behind it
    Y2 = torch.empty(M, E, dtype=u.BF, device="cuda")
    A2 = torch.empty(M, HID, dtype=u.BF, device="cuda")
    Z2 = torch.empty(M, HID, dtype=torch.uint8, device="cuda")
    u.call("vg_encoder_mlp_fwd", u.ptr(d["xn"]), u.ptr(img), u.ptr(db1), u.ptr(db2), u.ptr(d["R"]), u.ptr(A2), u.ptr(Z2), u.ptr(Y2), None,
           None, None, None, None, M, 1e-5, drop, seed, site, None, u.stream())
    u.sync()
    assert torch.equal(Y2, Y[:M]) and torch.equal(A2, A1[:M]) and torch.equal(Z2, Z8[:M])


def test_encoder_mlp_fwd_rows_do_not_depend_on_their_tile():
    """the same 2 080 rows as a problem of their own and as the head of 33 280 rows: bit-equal"""
    u = _u()
    g = torch.Generator().manual_seed(5)
    M1, M2 = 2080, 33280
    xn = u.rbf(torch.randn(M2, E, generator=g))
    W1 = u.rbf(torch.randn(HID, E, generator=g) / math.sqrt(E))
    W2 = u.rbf(torch.randn(E, HID, generator=g) / math.sqrt(HID))
    b1, b2 = torch.randn(HID, generator=g) * 0.1, torch.randn(E, generator=g) * 0.1
    R = u.rbf(torch.randn(M2, E, generator=g))
    gam, bet = torch.ones(E), torch.zeros(E)
    img = _pack_mlp(u, W1, W2)
    dx, dR = u.dev(xn, u.BF), u.dev(R, u.BF)
    db1, db2, dg, dbt = u.dev(b1), u.dev(b2), u.dev(gam), u.dev(bet)
    outs = []
    for M in (M1, M2):
        A1 = torch.empty(M, HID, dtype=u.BF, device="cuda"); Z8 = torch.empty(M, HID, dtype=torch.uint8, device="cuda")
        Y = torch.empty(M, E, dtype=u.BF, device="cuda"); Yn = torch.empty(M, E, dtype=u.BF, device="cuda")
        mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
        u.call("vg_encoder_mlp_fwd", u.ptr(dx), u.ptr(img), u.ptr(db1), u.ptr(db2), u.ptr(dR), u.ptr(A1), u.ptr(Z8), u.ptr(Y), u.ptr(Yn),
               u.ptr(mean), u.ptr(rstd), u.ptr(dg), u.ptr(dbt), M, 1e-5, 0.1, 3, 2, None, u.stream())
        u.sync()
        outs.append((A1[:M1].clone(), Z8[:M1].clone(), Y[:M1].clone(), Yn[:M1].clone(), mean[:M1].clone(), rstd[:M1].clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _mask_site(u, M, p, seed, site):
    return _mask(u, M, p, seed, site)


@pytest.mark.parametrize("M", [16, 128, 144, 2080, 16640, 32768, 33280])
@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_encoder_post_attention_fwd(M, drop):
    """vg_encoder_post_attention_fwd: out-projection + dropout + residual, norm2, fc1, GELU, fc2 + dropout + residual and the next norm1 in one
    launch (src/v2/modules.py:179-182, :168), every stored tensor against fp32 PyTorch with bf16 roundings where the kernel stores."""
    u = _u()
    g = torch.Generator().manual_seed(M + 11)
    ao = u.rbf(torch.randn(M, E, generator=g))
    x = u.rbf(torch.randn(M, E, generator=g))
    Wo = u.rbf(torch.randn(E, E, generator=g) / math.sqrt(E))
    bo = torch.randn(E, generator=g) * 0.1
    W1 = u.rbf(torch.randn(HID, E, generator=g) / math.sqrt(E))
    b1 = torch.randn(HID, generator=g) * 0.1
    W2 = u.rbf(torch.randn(E, HID, generator=g) / math.sqrt(HID))
    b2 = torch.randn(E, generator=g) * 0.1
    g2, be2 = 1.0 + 0.2 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    gn, ben = 1.0 + 0.2 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    seed, sa, sm = 17, 3, 4
    lin = ao @ Wo.t() + bo
    if drop:
        lin = lin * _mask(u, M, drop, seed, sa)
    xmid = u.rbf(lin + x)
    xn2 = u.rbf(F.layer_norm(xmid, (E,), g2, be2, 1e-5))
    z = xn2 @ W1.t() + b1
    a1 = u.rbf(F.gelu(z))
    y = a1 @ W2.t() + b2
    if drop:
        y = y * _mask(u, M, drop, seed, sm)
    y = u.rbf(y + xmid)
    yn = F.layer_norm(y, (E,), gn, ben, 1e-5)

    L = u._lib.lib()
    img = torch.empty(L.vg_encoder_post_attention_image_elems(), dtype=u.BF, device="cuda")
    dWo, dW1, dW2 = u.dev(Wo, u.BF), u.dev(W1, u.BF), u.dev(W2, u.BF)
    u.call("vg_encoder_post_attention_pack", u.ptr(dWo), u.ptr(dW1), u.ptr(dW2), u.ptr(img), u.stream())
    d_ao, d_x = u.dev(ao, u.BF), u.dev(x, u.BF)
    dv = [u.dev(t) for t in (bo, b1, b2, g2, be2, gn, ben)]
    XM = torch.full((M + 16, E), 7.0, dtype=u.BF, device="cuda"); XN = torch.full((M + 16, E), 7.0, dtype=u.BF, device="cuda")
    A1 = torch.full((M + 16, HID), 7.0, dtype=u.BF, device="cuda"); Z8 = torch.full((M + 16, HID), 9, dtype=torch.uint8, device="cuda")
    Y = torch.full((M + 16, E), 7.0, dtype=u.BF, device="cuda"); Yn = torch.full((M + 16, E), 7.0, dtype=u.BF, device="cuda")
    m2, r2, mn, rn = (torch.empty(M, device="cuda") for _ in range(4))

    def run(XM_, XN_, A1_, Z8_, Y_, Yn_):
        u.call("vg_encoder_post_attention_fwd", u.ptr(d_ao), u.ptr(d_x), u.ptr(img), u.ptr(dv[0]), u.ptr(dv[1]), u.ptr(dv[2]), u.ptr(dv[3]), u.ptr(dv[4]),
               u.ptr(dv[5]), u.ptr(dv[6]), u.ptr(XM_), u.ptr(XN_), u.ptr(m2), u.ptr(r2), u.ptr(A1_), u.ptr(Z8_), u.ptr(Y_), u.ptr(Yn_), u.ptr(mn), u.ptr(rn),
               M, 1e-5, drop, seed, sa, sm, None, u.stream())
        u.sync()
    run(XM, XN, A1, Z8, Y, Yn)
    u.assert_close(XM[:M], xmid, BF_TOL, "x_mid")
    xk = XM[:M].float().cpu()
    u.assert_close(XN[:M], F.layer_norm(xk, (E,), g2, be2, 1e-5), BF_TOL, "xn2 vs LN(own x_mid)")
    u.assert_close(m2, xk.mean(1), 3e-5, "mean2", floor=1e-6)
    u.assert_close(r2, 1.0 / torch.sqrt(xk.var(1, unbiased=False) + 1e-5), 3e-5, "rstd2")
    # the MLP on the kernel's own xn2 (one flipped bf16 rounding upstream would otherwise be charged to the stages behind it)
    xnk = XN[:M].float().cpu()
    zk = xnk @ W1.t() + b1
    a1k = u.rbf(F.gelu(zk))
    u.assert_close(A1[:M], a1k, BF_TOL, "a1 vs gelu(fc1(own xn2))")
    dz = (Z8[:M].float().cpu() - 27.0) * 0.005
    assert float((dz - _gelu_grad(zk)).abs().max()) <= 0.0051
    yk_ref = A1[:M].float().cpu() @ W2.t() + b2
    if drop:
        yk_ref = yk_ref * _mask(u, M, drop, seed, sm)
    u.assert_close(Y[:M], u.rbf(yk_ref + xk), BF_TOL, "Y vs fc2(own a1) + own x_mid")
    yk = Y[:M].float().cpu()
    u.assert_close(Yn[:M], F.layer_norm(yk, (E,), gn, ben, 1e-5), BF_TOL, "Yn vs LN(own Y)")
    u.assert_close(mn, yk.mean(1), 3e-5, "mean", floor=1e-6)
    u.assert_close(rn, 1.0 / torch.sqrt(yk.var(1, unbiased=False) + 1e-5), 3e-5, "rstd")
    # end to end against the reference chain (loose: three bf16 tensors deep)
    u.assert_close(Y[:M], y, 2.0 ** -5, "Y vs reference")
    u.assert_close(Yn[:M], yn, 2.0 ** -4, "Yn vs reference")
    for t, fill in ((XM, 7.0), (XN, 7.0), (A1, 7.0), (Y, 7.0), (Yn, 7.0)):
        assert bool((t[M:] == fill).all()), "rows beyond M were written"
    assert bool((Z8[M:] == 9).all())
    # bitwise repeatable
    outs2 = [torch.empty_like(t) for t in (XM, XN, A1, Z8, Y, Yn)]
    run(*outs2)
    for a_, b_ in zip((XM, XN, A1, Z8, Y, Yn), outs2):
        assert torch.equal(a_[:M], b_[:M])
