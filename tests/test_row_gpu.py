"""Parity of the full-row Linear + LayerNorm kernels (csrc/gemm_row.hip, through the C ABI) against fp32 PyTorch.

vg_linear_ln_fwd replaces out_projection / fc2 + dropout + residual + the LayerNorm that reads the sum
(src/v2/modules.py:168,172,179-183); vg_linear_dgrad_ln_bwd replaces autograd of queries|keys|values / fc1 and of the
LayerNorm in front of them (modules.py:178-181).  Inputs are rounded to bf16 first; the kernels round the GEMM result to
bf16 once (as the unfused kernels did at their HBM round trip), so the references do the same: tolerance 2^-7 of max|ref|
for bf16 outputs (plus one ulp of the intermediate), 3e-5 for fp32 statistics, 1e-4 for the column sums.

Row counts: M = 16 * units over min(256, ceil(units / 2)) workgroups - the sizes below give tiles of 1..9 m-tiles, several
tiles per workgroup (M = 66 560: 16-17 units each) and the full C2 launches (M = 16 640 / 33 280: 4-5 and 8-9 units)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF_TOL = 2.0 ** -7
E = 384  # set per test by the `width` fixture below


@pytest.fixture(params=[384, 512], autouse=True)
def width(request):
    """Every test of this file runs at both embedding widths the kernel is built for: 384 (C1-C3; the entry points without suffix)
    and 512 (C4; the `_e` entry points that take the width, round 4) - 4 n-tiles per wave, tiles of at most 6 m-tiles, 32 lanes per
    row in the backward epilogue, the three column sums folded one after the other."""
    global E
    E = request.param
    yield
    E = 384


def _u():
    import gpu_util
    return gpu_util


def _rcall(u, name, *args):
    """row-kernel entry point at the current width"""
    if E == 384:
        u.call(name, *args)
    else:
        u.call(name + "_e", E, *args)


def _pack(u, W, K, transposed):
    L = u._lib.lib()
    n = L.vg_row_pack_elems(K) if E == 384 else L.vg_row_pack_elems_e(E, K)
    assert n == E * K
    Wp = torch.empty(n, dtype=u.BF, device="cuda")
    dW = u.dev(W, u.BF)
    _rcall(u, "vg_row_pack_weight", u.ptr(dW), W.shape[1], K, 1 if transposed else 0, u.ptr(Wp), u.stream())
    return Wp


def _mask(u, M, p, seed, site):
    ones = torch.ones(M, E, dtype=u.BF, device="cuda")
    m = torch.empty_like(ones)
    u.call("vg_dropout_apply", u.ptr(ones), u.ptr(m), M * E, p, seed, site, None, u.stream())
    u.sync()
    return m.float().cpu()


SHAPES = [(16, 384), (48, 768), (1040, 384), (2080, 1152), (2096, 768), (4160, 768), (16640, 384), (16640, 1152),
          (33280, 768), (33280, 1152), (66560, 384)]


@pytest.mark.parametrize("M,K", SHAPES)
@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_linear_ln_fwd(M, K, drop):
    u = _u()
    g = torch.Generator().manual_seed(M + K)
    A = u.rbf(torch.randn(M, K, generator=g))
    W = u.rbf(torch.randn(E, K, generator=g) / math.sqrt(K))
    b = torch.randn(E, generator=g) * 0.1
    R = u.rbf(torch.randn(M, E, generator=g))
    gam = 1.0 + 0.2 * torch.randn(E, generator=g)
    bet = 0.1 * torch.randn(E, generator=g)
    seed, site = 77, 5
    y = A @ W.t() + b
    if drop:
        y = y * _mask(u, M, drop, seed, site)
    y = u.rbf(y + R)  # the stored sum is bf16, and the statistics are those of the stored values
    yn = F.layer_norm(y, (E,), gam, bet, 1e-5)
    mu = y.mean(1)
    rs = 1.0 / torch.sqrt(y.var(1, unbiased=False) + 1e-5)

    Wp = _pack(u, W, K, False)
    dA, db, dR, dg, dbt = u.dev(A, u.BF), u.dev(b), u.dev(R, u.BF), u.dev(gam), u.dev(bet)
    Y = torch.full((M + 16, E), 7.0, dtype=u.BF, device="cuda")
    Yn = torch.full((M + 16, E), 7.0, dtype=u.BF, device="cuda")
    mean = torch.empty(M, device="cuda")
    rstd = torch.empty(M, device="cuda")
    _rcall(u, "vg_linear_ln_fwd", u.ptr(dA), u.ptr(Wp), u.ptr(db), u.ptr(dR), u.ptr(Y), u.ptr(Yn), u.ptr(mean), u.ptr(rstd), u.ptr(dg), u.ptr(dbt),
           M, K, 1e-5, drop, seed, site, None, u.stream())
    u.sync()
    u.assert_close(Y[:M], y, BF_TOL, "Y")
    # LayerNorm of a sum that may differ by one bf16 ulp from the reference's: compare against the LayerNorm of the kernel's own Y too
    yk = Y[:M].float().cpu()
    u.assert_close(Yn[:M], F.layer_norm(yk, (E,), gam, bet, 1e-5), BF_TOL, "Yn vs LN(own Y)")
    u.assert_close(mean, yk.mean(1), 3e-5, "mean", floor=1e-6)
    u.assert_close(rstd, 1.0 / torch.sqrt(yk.var(1, unbiased=False) + 1e-5), 3e-5, "rstd")
    u.assert_close(Yn[:M], yn, 2.0 ** -5, "Yn vs reference")
    u.assert_close(mean, mu, 2.0 ** -6, "mean vs reference", floor=1e-3)
    u.assert_close(rstd, rs, 2.0 ** -7, "rstd vs reference")
    assert bool((Y[M:] == 7.0).all()) and bool((Yn[M:] == 7.0).all()), "rows beyond M were written"
    # without a LayerNorm behind it (the last block's fc2)
    Y2 = torch.empty(M, E, dtype=u.BF, device="cuda")
    _rcall(u, "vg_linear_ln_fwd", u.ptr(dA), u.ptr(Wp), u.ptr(db), u.ptr(dR), u.ptr(Y2), None, None, None, None, None,
           M, K, 1e-5, drop, seed, site, None, u.stream())
    u.sync()
    assert torch.equal(Y2, Y[:M])


@pytest.mark.parametrize("M,K", SHAPES)
@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_linear_dgrad_ln_bwd(M, K, drop):
    u = _u()
    g = torch.Generator().manual_seed(M * 3 + K)
    dY = u.rbf(torch.randn(M, K, generator=g))
    W = u.rbf(torch.randn(K, E, generator=g) / math.sqrt(K))  # nn.Linear(E, K).weight: the Linear the LayerNorm feeds
    x = u.rbf(torch.randn(M, E, generator=g) * 1.5 + 0.3)
    gres = u.rbf(torch.randn(M, E, generator=g))
    gam = 1.0 + 0.2 * torch.randn(E, generator=g)
    seed, site = 31, 2
    mu = x.mean(1, keepdim=True)
    rs = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    xh = (x - mu) * rs
    dxn = u.rbf(dY @ W)  # the input gradient of the Linear is rounded to bf16 once (the unfused pair stored it)
    gg = dxn * gam
    dx = u.rbf(gres + rs * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True)))
    dxm = u.rbf(dx * _mask(u, M, drop, seed, site)) if drop else None
    terms = [dxn * xh, dxn, dxm if drop else dx]
    parts_ref = torch.cat([t.sum(0) for t in terms])

    WpT = _pack(u, W, K, True)
    L = u._lib.lib()
    nparts = L.vg_row_parts(M)
    assert nparts >= 1
    d_dY, d_x, d_gres, d_gam = u.dev(dY, u.BF), u.dev(x, u.BF), u.dev(gres, u.BF), u.dev(gam)
    d_mu, d_rs = u.dev(mu.flatten()), u.dev(rs.flatten())
    out = torch.full((M + 16, E), 7.0, dtype=u.BF, device="cuda")
    outm = torch.full((M + 16, E), 7.0, dtype=u.BF, device="cuda")
    part = torch.full((nparts + 1, 3 * E), 7.0, device="cuda")
    _rcall(u, "vg_linear_dgrad_ln_bwd", u.ptr(d_dY), u.ptr(WpT), u.ptr(d_x), u.ptr(d_mu), u.ptr(d_rs), u.ptr(d_gam), u.ptr(d_gres), u.ptr(out),
           u.ptr(outm) if drop else None, u.ptr(part), M, K, drop, seed, site, None, u.stream())
    u.sync()
    u.assert_close(out[:M], dx, BF_TOL * 1.5, "dx")  # one ulp of the bf16 d x_hat on top of the output rounding
    if drop:
        u.assert_close(outm[:M], dxm, BF_TOL * 1.5, "dxm")
    assert bool((out[M:] == 7.0).all()) and bool((part[nparts:] == 7.0).all()), "written past the end"
    got = part[:nparts].sum(0).cpu()
    for i, name in enumerate(("d gamma", "d beta", "colsum")):
        ref = parts_ref[i * E:(i + 1) * E]
        scale = float(ref.abs().max())
        # sums of M terms that each may differ by one bf16 ulp (2^-8 of the term): random-walk growth, 1.5 sigma-ish headroom
        tol = 2.0 ** -8 * math.sqrt(M) * 1.5 * float(terms[i].abs().max()) + 1e-4 * scale
        assert float((got[i * E:(i + 1) * E] - ref).abs().max()) <= tol, name
    # bitwise repeatable (no atomics, fixed fold order)
    out2 = torch.empty_like(out)
    part2 = torch.empty_like(part)
    _rcall(u, "vg_linear_dgrad_ln_bwd", u.ptr(d_dY), u.ptr(WpT), u.ptr(d_x), u.ptr(d_mu), u.ptr(d_rs), u.ptr(d_gam), u.ptr(d_gres), u.ptr(out2),
           None, u.ptr(part2), M, K, 0.0, seed, site, None, u.stream())
    u.sync()
    assert torch.equal(out2[:M], out[:M]) and torch.equal(part2[:nparts, :2 * E], part[:nparts, :2 * E])


def test_row_kernel_rejects_unsupported_shapes():
    u = _u()
    L = u._lib.lib()
    if E != 384:
        assert L.vg_row_pack_elems_e(E, 48) == -2 and L.vg_row_pack_elems_e(768, 768) == -2 and L.vg_row_pack_elems_e(E, 1024) == E * 1024
        a = torch.zeros(128, 768, dtype=u.BF, device="cuda")
        rc = L.vg_linear_ln_fwd_e(768, a.data_ptr(), a.data_ptr(), None, None, a.data_ptr(), None, None, None, None, None, 128, 768, 1e-5, 0.0, 0, 0, None,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == -3  # E = 768 does not fit the kernel's LDS ring
        return
    assert L.vg_row_parts(130) == 0 and L.vg_row_parts(8) == 0 and L.vg_row_parts(16) == 1 and L.vg_row_parts(33280) == 256
    assert L.vg_row_pack_elems(48) == -2
    a = torch.zeros(130, 384, dtype=u.BF, device="cuda")
    wp = torch.zeros(384 * 384, dtype=u.BF, device="cuda")
    y = torch.zeros(130, 384, dtype=u.BF, device="cuda")
    rc = L.vg_linear_ln_fwd(a.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), None, None, None, None, None, 130, 384, 1e-5, 0.0, 0, 0, None,
                            C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -3


@pytest.mark.parametrize("M,K", [(33280, 1152), (16640, 768)])
def test_row_kernel_race_screen(M, K):
    """The ring protocol (counted vmcnt per wave, one barrier per stage, fragments a stage ahead) under repetition: 40 launches of
    the same problem must be bit-identical (an early fragment read or a late DMA shows as a rare differing tile)."""
    u = _u()
    g = torch.Generator().manual_seed(5)
    A = u.dev(torch.randn(M, K, generator=g), u.BF)
    W = torch.randn(E, K, generator=g) / math.sqrt(K)
    Wp = _pack(u, u.rbf(W), K, False)
    R = u.dev(torch.randn(M, E, generator=g), u.BF)
    outs = []
    for i in range(40):
        Y = torch.empty(M, E, dtype=u.BF, device="cuda")
        _rcall(u, "vg_linear_ln_fwd", u.ptr(A), u.ptr(Wp), None, u.ptr(R), u.ptr(Y), None, None, None, None, None, M, K, 1e-5, 0.0, 0, 0, None, u.stream())
        outs.append(Y)
    u.sync()
    for i in range(1, 40):
        assert torch.equal(outs[0], outs[i]), f"launch {i} differs"


@pytest.mark.parametrize("K", [384, 1152])
def test_rows_do_not_depend_on_the_tile_they_land_in(K):
    """A row's result must be BIT-identical whatever the batch around it: the same 2 080 rows are run as a problem of their own
    (tiles of 2 m-tiles), as the head of 16 640 rows (4-5 m-tiles) and of 33 280 rows (8-9 m-tiles).  The epilogue is
    instantiated per tile height; this is what catches a multiply-add contracted in one instantiation and not in another."""
    u = _u()
    g = torch.Generator().manual_seed(11)
    Mbig, Ms = 33280, 2080
    A = u.dev(torch.randn(Mbig, K, generator=g), u.BF)
    W = u.rbf(torch.randn(E, K, generator=g) / math.sqrt(K))
    Wp = _pack(u, W, K, False)
    WpT = _pack(u, u.rbf(torch.randn(K, E, generator=g) / math.sqrt(K)), K, True)
    R = u.dev(torch.randn(Mbig, E, generator=g), u.BF)
    X = u.dev(torch.randn(Mbig, E, generator=g) * 1.5 + 0.3, u.BF)
    gam, bet, b = u.dev(1.0 + 0.2 * torch.randn(E, generator=g)), u.dev(0.1 * torch.randn(E, generator=g)), u.dev(0.1 * torch.randn(E, generator=g))
    mu = X.float().mean(1).contiguous()
    rs = (1.0 / torch.sqrt(X.float().var(1, unbiased=False) + 1e-5)).contiguous()
    L = u._lib.lib()
    outs = []
    for M in (Ms, 16640, Mbig):
        Y = torch.empty(M, E, dtype=u.BF, device="cuda"); Yn = torch.empty_like(Y)
        mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
        _rcall(u, "vg_linear_ln_fwd", u.ptr(A), u.ptr(Wp), u.ptr(b), u.ptr(R), u.ptr(Y), u.ptr(Yn), u.ptr(mean), u.ptr(rstd), u.ptr(gam), u.ptr(bet),
               M, K, 1e-5, 0.1, 3, 1, None, u.stream())
        dx = torch.empty(M, E, dtype=u.BF, device="cuda"); dxm = torch.empty_like(dx)
        part = torch.empty(L.vg_row_parts(M), 3 * E, device="cuda")
        _rcall(u, "vg_linear_dgrad_ln_bwd", u.ptr(A), u.ptr(WpT), u.ptr(X), u.ptr(mu), u.ptr(rs), u.ptr(gam), u.ptr(R), u.ptr(dx), u.ptr(dxm), u.ptr(part),
               M, K, 0.1, 3, 1, None, u.stream())
        u.sync()
        outs.append([t[:Ms].clone() for t in (Y, Yn, mean, rstd, dx, dxm)])
    for other in outs[1:]:
        for name, a_, b_ in zip(("Y", "Yn", "mean", "rstd", "dx", "dxm"), outs[0], other):
            assert torch.equal(a_, b_), name


@pytest.mark.parametrize("M,K", [(2080, 384), (16640, 768), (33280, 384)])
def test_fused_forward_is_bit_identical_to_the_unfused_pair(M, K):
    """vg_linear_ln_fwd against vg_linear_fwd (+ residual) followed by vg_layernorm_fwd: the sum, the statistics and the
    normalised rows agree BIT for bit (same k order in the MFMA chains, one rounding of the sum, the statistics written with
    the arithmetic forms norm.hip compiles to).  This is what lets batches that take the fused path (rows in whole units of 16)
    and batches that do not (tests/test_fullsize_gpu.py: 8 images) produce identical logits."""
    u = _u()
    g = torch.Generator().manual_seed(M + K)
    A = u.dev(torch.randn(M, K, generator=g), u.BF)
    W = u.dev(torch.randn(E, K, generator=g) / math.sqrt(K), u.BF)
    b = u.dev(torch.randn(E, generator=g) * 0.1)
    R = u.dev(torch.randn(M, E, generator=g), u.BF)
    gam, bet = u.dev(1 + 0.2 * torch.randn(E, generator=g)), u.dev(0.1 * torch.randn(E, generator=g))
    Wp = _pack(u, W.float().cpu(), K, False)
    Y = torch.empty(M, E, dtype=u.BF, device="cuda"); Yn = torch.empty_like(Y)
    mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    _rcall(u, "vg_linear_ln_fwd", u.ptr(A), u.ptr(Wp), u.ptr(b), u.ptr(R), u.ptr(Y), u.ptr(Yn), u.ptr(mean), u.ptr(rstd), u.ptr(gam), u.ptr(bet),
           M, K, 1e-5, 0.0, 0, 0, None, u.stream())
    Y2 = torch.empty_like(Y); Yn2 = torch.empty_like(Y); mean2 = torch.empty_like(mean); rstd2 = torch.empty_like(rstd)
    u.call("vg_linear_fwd", u.ptr(A), u.ptr(W), u.ptr(b), u.ptr(R), u.ptr(Y2), None, None, M, E, K, 0, 0.0, u.stream())
    u.call("vg_layernorm_fwd", u.ptr(Y2), E, u.ptr(gam), u.ptr(bet), u.ptr(Yn2), E, u.ptr(mean2), u.ptr(rstd2), M, E, 1e-5, u.stream())
    u.sync()
    assert torch.equal(Y, Y2) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2) and torch.equal(Yn, Yn2)


# ---- the same kernels with the v1 generator's self-modulated LayerNorm (src/v1/spectral_layer_norm.py:19-20) in the epilogue ----
SLN_SHAPES = [(64, 384), (96, 768), (1024, 384), (2048, 1152), (8192, 384), (8192, 1152)]


@pytest.mark.parametrize("M,K", SLN_SHAPES)
@pytest.mark.parametrize("table", [False, True])
def test_linear_sln_fwd(M, K, table):
    u = _u()
    T = 32
    g = torch.Generator().manual_seed(M + K + int(table))
    A = u.rbf(torch.randn(M, K, generator=g))
    W = u.rbf(torch.randn(E, K, generator=g) / math.sqrt(K))
    b = torch.randn(E, generator=g) * 0.1
    R = u.rbf(torch.randn(M, E, generator=g))
    tab = torch.randn(T, E, generator=g)
    wmod = u.rbf(torch.randn(M, E, generator=g))
    lw, lb = 1.0 + 0.2 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    sc = torch.tensor([0.7, -0.3])
    drop, seed, site = 0.2, 5, 101
    y = (A @ W.t() + b) * _mask(u, M, drop, seed, site)
    y = u.rbf(y + (tab.repeat(M // T, 1) if table else R))
    ln = F.layer_norm(y, (E,), lw, lb, 1e-5)
    yn = wmod * (sc[0] * ln + sc[1])
    Wp = _pack(u, W, K, False)
    dA, db, dR, dtab, dwm, dlw, dlb, dsc = u.dev(A, u.BF), u.dev(b), u.dev(R, u.BF), u.dev(tab), u.dev(wmod, u.BF), u.dev(lw), u.dev(lb), u.dev(sc)
    Y = torch.empty(M, E, dtype=u.BF, device="cuda"); Yn = torch.empty_like(Y)
    mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    _rcall(u, "vg_linear_sln_fwd", u.ptr(dA), u.ptr(Wp), u.ptr(db), None if table else u.ptr(dR), u.ptr(dtab) if table else None, T, u.ptr(Y), u.ptr(Yn),
           u.ptr(mean), u.ptr(rstd), u.ptr(dwm), u.ptr(dlw), u.ptr(dlb), u.ptr(dsc), C.c_void_p(dsc.data_ptr() + 4), M, K, 1e-5, drop, seed, site, None, u.stream())
    u.sync()
    u.assert_close(Y, y, BF_TOL, "Y")
    yk = Y.float().cpu()
    u.assert_close(Yn, wmod * (sc[0] * F.layer_norm(yk, (E,), lw, lb, 1e-5) + sc[1]), BF_TOL, "Yn vs SLN(own Y)")
    u.assert_close(mean, yk.mean(1), 3e-5, "mean", floor=1e-6)
    u.assert_close(rstd, 1.0 / torch.sqrt(yk.var(1, unbiased=False) + 1e-5), 3e-5, "rstd")
    u.assert_close(Yn, yn, 2.0 ** -5, "Yn vs reference")


@pytest.mark.parametrize("M,K", SLN_SHAPES)
@pytest.mark.parametrize("bcast,acc", [(0, 0), (32, 1)])
def test_linear_dgrad_sln_bwd(M, K, bcast, acc):
    u = _u()
    g = torch.Generator().manual_seed(M * 3 + K + bcast)
    dY = u.rbf(torch.randn(M, K, generator=g))
    W = u.rbf(torch.randn(K, E, generator=g) / math.sqrt(K))
    hx = u.rbf(torch.randn(bcast if bcast else M, E, generator=g) * 1.5 + 0.3)
    x = hx.repeat(M // bcast, 1) if bcast else hx
    gres = u.rbf(torch.randn(M, E, generator=g))
    wmod = u.rbf(torch.randn(M, E, generator=g))
    lw, lb = 1.0 + 0.2 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    gs, bs = 0.7, -0.3
    dw0 = torch.randn(M, E, generator=g)
    drop, seed, site = 0.2, 9, 100
    mu = x.mean(1, keepdim=True)
    rs = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    xh = (x - mu) * rs
    dxn = u.rbf(dY @ W)
    l = xh * lw + lb
    dw_ref = dxn * (gs * l + bs) + (dw0 if acc else 0.0)
    dgs_ref, dbs_ref = float((dxn * wmod * l).sum()), float((dxn * wmod).sum())
    de = dxn * wmod * gs
    gg = de * lw
    dx = u.rbf(gres + rs * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True)))
    dxm = u.rbf(dx * _mask(u, M, drop, seed, site))
    terms = [de * xh, de, dxm]
    WpT = _pack(u, W, K, True)
    L = u._lib.lib()
    nparts = L.vg_row_parts(M)
    PW = 3 * E + 64
    d_dY, d_h, d_gres, d_wm, d_lw, d_lb = u.dev(dY, u.BF), u.dev(hx, u.BF), u.dev(gres, u.BF), u.dev(wmod, u.BF), u.dev(lw), u.dev(lb)
    d_sc = u.dev(torch.tensor([gs, bs]))
    d_mu, d_rs = u.dev(mu.flatten()), u.dev(rs.flatten())
    out = torch.empty(M, E, dtype=u.BF, device="cuda"); outm = torch.empty_like(out)
    dwa = u.dev(dw0.clone())
    part = torch.full((nparts + 1, PW), 7.0, device="cuda")
    _rcall(u, "vg_linear_dgrad_sln_bwd", u.ptr(d_dY), u.ptr(WpT), u.ptr(d_h), bcast, u.ptr(d_wm), u.ptr(d_mu), u.ptr(d_rs), u.ptr(d_lw), u.ptr(d_lb),
           u.ptr(d_sc), C.c_void_p(d_sc.data_ptr() + 4), u.ptr(d_gres), u.ptr(out), u.ptr(outm), u.ptr(dwa), acc, u.ptr(part), M, K, drop, seed, site, None, u.stream())
    u.sync()
    u.assert_close(out, dx, BF_TOL * 1.5, "dh")
    u.assert_close(outm, dxm, BF_TOL * 1.5, "dhm")
    u.assert_close(dwa, dw_ref, BF_TOL, "dw_acc")  # fp32 products of a bf16 GEMM result that may differ from the reference's by one ulp
    assert bool((part[nparts:] == 7.0).all())
    got = part[:nparts].sum(0).cpu()
    for i, name in enumerate(("d lw", "d lb", "colsum")):
        ref = terms[i].sum(0)
        tol = 2.0 ** -8 * math.sqrt(M) * 1.5 * float(terms[i].abs().max()) + 1e-4 * float(ref.abs().max())
        assert float((got[i * E:(i + 1) * E] - ref).abs().max()) <= tol, name
    for val, ref, t in ((float(got[3 * E]), dgs_ref, dxn * wmod * l), (float(got[3 * E + 1]), dbs_ref, dxn * wmod)):
        assert abs(val - ref) <= 2.0 ** -7 * float(t.norm()) + 1e-4 * abs(ref), (val, ref)
