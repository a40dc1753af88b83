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
