"""TIGHT parity tier, stage by stage: every stage of the full-depth networks against oracle/bf16_model.py at 2^-6
(ONE number, `TIGHT` below: two bf16 ulps of a tensor's largest element; the SLN scalars additionally at 2^-7 of the 2-norm of
their terms, `SLN_TIGHT`).

bf16 storage makes a deep network chaotic at the ulp level: one flipped rounding decision moves every downstream value by
a fraction of an ulp and flips more decisions, so after two encoder blocks ANY two correct bf16 implementations differ by
about as much as either differs from fp32 arithmetic (tests/parity_tiers.py prints the table: depth 1 matches the
rounding-faithful model to 2e-7 / 1e-3, depth 2 no better than the fp32 oracle).  A whole-network comparison therefore
cannot be tighter than the LOOSE tier of test_net_gpu.py.  What CAN be held tight is each stage on its own: the engine
runs the FULL network (6 blocks, real inter-stage plumbing, parity scratch sets, staged backward), and each stage of the
model is fed the tensors the engine really produced for that stage (its saved input activations from the workspace, the
upstream gradient it handed to that stage) - teacher forcing.  A mis-scaled gradient, a wrong residual or a stale buffer
in any single stage shows up as an O(1) relative error here; legitimate differences are fp32 summation order and sparse
one-ulp flips inside ONE stage.  One bf16 ulp of the largest element of a tensor is between 2^-8 and 2^-7 of max|ref|,
so the bound is TWO ulps of the largest element, 2^-6 (1.6e-2); the median over the ~120 tensors of a network is ~1e-3
(printed).  Sums of random-sign terms (the SLN scalars gamma / beta, R*E terms each) are judged against the 2-norm of
their terms when that exceeds the sum itself: a mis-scaled scalar is off by |sum| ~ that norm, rounding noise by 2^-8 of it.
On top of that rule every one of the 18 SLN scalar gradients (9 sites x gamma, beta; src/v1/spectral_layer_norm.py:7-20) must
agree with the model to 2^-7 of that 2-norm whatever the size of the sum (the table is printed: profiles/r03_stage_parity_summary.txt).
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TIGHT = 2.0 ** -6
SLN_TIGHT = 2.0 ** -7  # |d gamma - ref|, |d beta - ref| as a fraction of the 2-norm of the terms they sum


def _view(ws, off, shape, dtype):
    n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
    return ws[off:off + n].view(dtype).view(*shape)


def _leaf(t):
    return t.detach().float().cpu().clone().requires_grad_(True)


def _rel_err(got, ref, scale=None, floor=1e-6):
    """max|got - ref| / max|ref| (or / scale) - recorded per tensor, judged together by _report"""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape and bool(torch.isfinite(got).all())
    s = float(ref.abs().max()) if scale is None else scale
    return float((got - ref).abs().max()) / max(s, floor)


def _report(title, worst):
    top = sorted(worst.items(), key=lambda kv: -kv[1])
    vals = sorted(worst.values())
    print(f"{title}: {len(worst)} tensors checked stage by stage; median {vals[len(vals) // 2]:.2e}, worst",
          [(k, f"{v:.2e}") for k, v in top[:8]])
    bad = [(k, v) for k, v in top if v > TIGHT]
    assert not bad, f"{title}: beyond 2^-6 of max|ref|: {[(k, f'{v:.2e}') for k, v in bad[:12]]} ({len(bad)} tensors)"


# B = 32 (M = 2080 rows = 65 units of 32): the K = 384 Linears of that case run on the weights-in-registers kernel
# "c5-fp8": the C5 geometry (E = 768, 12 heads of 64, 3 blocks deep here) with e4m3 attention operands (VgVitNet.attn_fp8) against the
# model's fp8 mode: the same 2^-6 per stage as the bf16 cases - the rounding-faithful tier of the fp8 network
@pytest.mark.parametrize("case,B,dropout", [("c1", 3, 0.0), ("c1", 2, 0.1), ("c4", 2, 0.0), ("c1", 32, 0.1), ("c5-fp8", 2, 0.0), ("c5-fp8", 2, 0.1)])
def test_vit_every_stage_against_the_model(case, B, dropout):
    import gpu_util as u
    from cases import VIT_CASES
    from weights import make_input, make_state
    from oracle import bf16_model as bm, vit_oracle as vo
    from vit_gan_amd import _lib, flat

    fp8 = case.endswith("-fp8")
    c = dict(VIT_CASES["c5"], layers=3) if fp8 else VIT_CASES[case]
    d = vo.VitDims(channels=c["channels"], image=c["image"], patch=c["patch"], embed=c["embed"], heads=c["heads"],
                   layers=c["layers"], mlp_ratio=c["mlp_ratio"], classes=c["classes"])
    L, S, E = d.layers, d.seq, d.embed
    st_np = make_state(vo.vit_param_shapes(d), c["seed"], "vit")
    x = torch.from_numpy(make_input((B, d.channels, d.image, d.image), c["seed"], "uniform"))
    dd = flat.vit_dims_struct(d.channels, d.image, d.patch, E, d.heads, L, d.mlp_ratio, d.classes)
    lay, slots = flat.vit_layout(dd), flat.vit_slots(dd)
    P = flat.pack(slots, lay.total, st_np, device="cuda")
    Pb, G = P.to(torch.bfloat16), torch.zeros_like(P)
    seed = 4321
    net = _lib.VgVitNet(dd, P.data_ptr(), Pb.data_ptr(), G.data_ptr(), dropout, seed, None, _lib.context(), 1 if fp8 else 0, 0)
    wm = _lib.VgVitWsMap()
    u.call("vg_vit_ws_map", C.byref(dd), B, C.byref(wm))
    ws = torch.zeros(_lib.lib().vg_vit_ws_bytes(C.byref(dd), B), dtype=torch.uint8, device="cuda")
    assert wm.total == ws.numel()
    logits = torch.empty(B, d.classes, device="cuda")
    X = x.cuda()
    u.call("vg_vit_forward", C.byref(net), B, u.ptr(X), 0, u.ptr(ws), u.ptr(logits), u.stream())
    u.sync()
    M = B * S
    Xs = [_view(ws, wm.X + l * M * E * 2, (B, S, E), torch.bfloat16).float().cpu().clone() for l in range(L)]
    # X[L]: the engine computes the top block's output on the CLS rows only (nothing else of it is ever read, modules.py:195)

    def cls_only(off):
        full = torch.zeros(B, S, E)
        full[:, 0] = _view(ws, off, (B, E), torch.bfloat16).float().cpu()
        return full
    Xs.append(cls_only(wm.xtop))
    R = torch.from_numpy(make_input((B, d.classes), c["seed"] + 1))
    Rd = R.cuda()
    dimg = torch.empty(B, d.channels, d.image, d.image, dtype=torch.bfloat16, device="cuda")
    gins = {}
    u.call("vg_vit_backward_stages", C.byref(net), B, u.ptr(ws), u.ptr(Rd), u.ptr(dimg), 1, 0, 1, u.stream())
    u.sync()
    gins[L] = cls_only(wm.dxtop)   # dL/dX[L]: its CLS rows; exactly zero on every other row
    for l in range(L - 1, -1, -1):
        stage = L - l
        u.call("vg_vit_backward_stages", C.byref(net), B, u.ptr(ws), u.ptr(Rd), u.ptr(dimg), 1, stage, stage + 1, u.stream())
        u.sync()
        gins[l] = _view(ws, wm.gin[(l & 1) ^ 1], (B, S, E), torch.bfloat16).float().cpu().clone()  # dL/dX[l]
    u.call("vg_vit_backward_stages", C.byref(net), B, u.ptr(ws), u.ptr(Rd), u.ptr(dimg), 1, L + 1, L + 2, u.stream())
    u.sync()
    grads = {k: v.clone() for k, v in flat.unpack(slots, G).items()}

    masks = {}
    if dropout > 0:
        keep = 256.0 / (256.0 - round(dropout * 256))

        def mask(site):
            ones = torch.ones(B, S, E, dtype=torch.bfloat16, device="cuda")
            out = torch.empty_like(ones)
            u.call("vg_dropout_apply", u.ptr(ones), u.ptr(out), ones.numel(), dropout, seed, site, None, u.stream())
            u.sync()
            return (out.float().cpu() > 0).float() * keep
        masks["embed"] = mask(0)
        for l in range(L):
            masks[("attn", l)], masks[("mlp", l)] = mask(1 + 2 * l), mask(2 + 2 * l)

    worst = {}

    def check(got, ref, what, scale=None):
        worst[what] = _rel_err(got, ref, scale)
        if worst[what] > 0.5:  # an O(1) miss: say what the two sides look like
            g_, r_ = got.detach().float().cpu(), ref.detach().float().cpu()
            print(f"{what}: |got| max {float(g_.abs().max()):.4e} mean {float(g_.abs().mean()):.4e}; |ref| max {float(r_.abs().max()):.4e} mean "
                  f"{float(r_.abs().mean()):.4e}; got/ref at ref's largest element {float(g_.flatten()[r_.abs().argmax()] / r_.flatten()[r_.abs().argmax()]):.4f}")

    def check_params(st, keys, what):
        for k in keys:
            # keys.bias: the exact gradient is identically zero (softmax is invariant to a key shift); what both sides hold
            # is rounding noise - measure it against the sibling queries.bias gradient
            scale = float(st[k.replace("keys", "queries")].grad.abs().max()) if k.endswith("keys.bias") else None
            check(grads[k], st[k].grad, f"{what}: grad {k}", scale)

    # ---- head: X[L] -> logits, dlogits -> dL/dX[L] ----
    st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in st_np.items()}
    xin = _leaf(Xs[L])
    out = bm.vit_head(st, bm.stored(xin))
    check(logits, out, "head: logits")
    (out * R).sum().backward()
    check(gins[L], xin.grad, "head: dL/dX[L]")
    check_params(st, [k for k in st if k.startswith(("vit.norm.", "vit.classifier."))], "head")
    # ---- encoder blocks, each on the engine's own X[l] and dL/dX[l+1] ----
    for l in range(L):
        st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in st_np.items()}
        xin = _leaf(Xs[l])
        out = bm.vit_block(st, bm.stored(xin), d, f"vit.encoder.{l}.", masks.get(("attn", l)), masks.get(("mlp", l)), fp8)
        if l == L - 1:
            check(Xs[l + 1][:, 0], out[:, 0], f"block {l}: X[l+1] (CLS rows)")
        else:
            check(Xs[l + 1], out, f"block {l}: X[l+1]")
        out.backward(gins[l + 1])
        check(gins[l], xin.grad, f"block {l}: dL/dX[l]")
        check_params(st, [k for k in st if k.startswith(f"vit.encoder.{l}.")], f"block {l}")
    # ---- embedding: image -> X[0], dL/dX[0] -> d image ----
    st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in st_np.items()}
    xin = x.clone().requires_grad_(True)
    out = bm.vit_embed(st, xin, d, mask=masks.get("embed"))
    check(Xs[0], out, "embedding: X[0]")
    out.backward(gins[0])
    check(dimg, xin.grad, "embedding: d image")
    check_params(st, [k for k in st if k.startswith("vit.embedding.")], "embedding")
    _report(f"ViT {case} B={B} p={dropout}", worst)


@pytest.mark.parametrize("B,dropout,patch", [(3, 0.0, 0), (2, 0.2, 0), (2, 0.0, 4), (32, 0.2, 0)])
def test_generator_every_stage_against_the_model(B, dropout, patch):
    import gpu_util as u
    from weights import make_input, make_state
    from oracle import bf16_model as bm, gen_oracle as go
    from vit_gan_amd import _lib, flat

    d = go.GenDims(tokens=64 if patch else 32, patch=patch)
    L, T, E = d.layers, d.tokens, d.embed
    st_np = make_state(go.gen_param_shapes(d), 21, "gen")
    z = torch.from_numpy(make_input((B, d.latent), 21))
    gd = _lib.VgGenDims(d.latent, T, E, d.heads, L, d.siren_hidden, d.out_features, d.omega0, patch, d.channels, d.image)
    lay, slots = flat.gen_layout(gd), flat.gen_slots(gd)
    P = flat.pack(slots, lay.total, st_np, device="cuda")
    Pb, G = P.to(torch.bfloat16), torch.zeros_like(P)
    seed = 99
    net = _lib.VgGenNet(gd, P.data_ptr(), Pb.data_ptr(), G.data_ptr(), dropout, seed, None, None)
    wm = _lib.VgGenWsMap()
    u.call("vg_gen_ws_map", C.byref(gd), B, C.byref(wm))
    ws = torch.zeros(_lib.lib().vg_gen_ws_bytes(C.byref(gd), B), dtype=torch.uint8, device="cuda")
    assert wm.total == ws.numel()
    img = torch.empty(B, d.channels, d.image, d.image, dtype=torch.bfloat16, device="cuda")
    Zd = z.cuda()
    u.call("vg_gen_forward", C.byref(net), B, u.ptr(Zd), u.ptr(ws), u.ptr(img), u.stream())
    u.sync()
    R_ = B * T
    wmod = _view(ws, wm.wmod, (B, T, E), torch.bfloat16).float().cpu().clone()
    hs = [_view(ws, wm.hout + l * R_ * E * 2, (B, T, E), torch.bfloat16).float().cpu().clone() for l in range(L)]
    Rimg = torch.from_numpy(make_input((B, d.channels, d.image, d.image), 22)).to(torch.bfloat16).float()
    Rd = Rimg.to(torch.bfloat16).cuda()

    def run(a, b):
        u.call("vg_gen_backward_stages", C.byref(net), B, u.ptr(ws), u.ptr(Rd), a, b, u.stream())
        u.sync()

    def gbuf(i):
        return _view(ws, wm.g[i], (B, T, E), torch.bfloat16).float().cpu().clone()

    def dw():
        return _view(ws, wm.dw_acc, (B, T, E), torch.float32).cpu().clone()
    g_in, dws = {}, {}
    run(0, 1)
    g_in[L], dws[L] = gbuf(0), dw()                       # dL/d h_L ; dw after the final SLN
    for l in range(L - 1, -1, -1):
        run(L - l, L - l + 1)
        done = L - l                                       # blocks processed so far
        g_in[l], dws[l] = gbuf(2 if done % 2 else 0), dw()  # dL/d h entering block l ; cumulative dw
    run(L + 1, L + 2)
    grads = {k: v.clone() for k, v in flat.unpack(slots, G).items()}

    masks = {}
    if dropout > 0:
        keep = 256.0 / (256.0 - round(dropout * 256))

        def mask(site):
            ones = torch.ones(B, T, E, dtype=torch.bfloat16, device="cuda")
            out = torch.empty_like(ones)
            u.call("vg_dropout_apply", u.ptr(ones), u.ptr(out), ones.numel(), dropout, seed, site, None, u.stream())
            u.sync()
            return (out.float().cpu() > 0).float() * keep
        for l in range(L):
            masks[("attn", l)], masks[("mlp", l)] = mask(100 + 2 * l), mask(101 + 2 * l)

    worst = {}

    def check(got, ref, what, floor=1e-6, scale=None):
        worst[what] = _rel_err(got, ref, scale, floor)

    sln_rows = []

    def scalar_scale(taps, k, st):
        """SLN scalars are sums of R*E random-sign terms (d gamma = sum dy w LN(h), d beta = sum dy w): the scale of such a
        sum's rounding noise - and of the error a mis-scaling would cause - is the 2-norm of its terms."""
        if not k.endswith((".gamma", ".beta")):
            return None
        y, w_, ln = taps[k.rsplit(".", 1)[0] + "."]
        t = y.grad * w_ * (ln if k.endswith(".gamma") else 1.0)
        got, ref, nrm = float(grads[k].reshape(-1)[0]), float(st[k].grad.reshape(-1)[0]), float(t.norm())
        sln_rows.append((k, got, ref, nrm, abs(got - ref) / max(nrm, 1e-12)))
        return max(float(st[k].grad.abs().max()), float(t.norm()))

    def fresh():
        return {k: torch.from_numpy(v).requires_grad_(True) for k, v in st_np.items()}
    # ---- mapping: z -> w ----
    st = fresh()
    check(wmod, bm.gen_mapping(st, z, d), "mapping: w")
    # ---- head: h_L, w -> image ; d image -> dL/d h_L, its share of dw, SLN / SIREN gradients ----
    st = fresh()
    hin, win = _leaf(hs[L - 1]), _leaf(wmod)
    taps = {}
    out = bm.rows_to_image(bm.gen_head(st, bm.stored(hin), win, d, taps=taps), d)
    check(img, out, "head: image")
    (out * Rimg).sum().backward()
    check(g_in[L], hin.grad, "head: dL/d h_L")
    check(dws[L], win.grad, "head: dw of the final SLN")
    for k in [k for k in st if k.startswith(("sln.", "output_network."))]:
        check(grads[k], st[k].grad, f"head: grad {k}", scale=scalar_scale(taps, k, st))
    # ---- blocks ----
    for l in range(L - 1, -1, -1):
        st = fresh()
        win = _leaf(wmod)
        if l == 0:
            h_sln, h_res = bm.gen_embedding(st, B, d)
        else:
            hin = _leaf(hs[l - 1])
            h_sln = h_res = bm.stored(hin)
        taps = {}
        out = bm.gen_block(st, f"transformer_layers.{l}.", h_sln, h_res, win, d, masks.get(("attn", l)), masks.get(("mlp", l)), taps=taps)
        check(hs[l], out, f"block {l}: h_out")
        out.backward(g_in[l + 1])
        check(dws[l] - dws[l + 1], win.grad, f"block {l}: dw of its two SLNs", floor=1e-5)
        if l == 0:   # embedding gradient = batch sum of the per-sample bf16 gradients
            check(grads["embedding"], st["embedding"].grad, "block 0: grad embedding")
        else:
            check(g_in[l], hin.grad, f"block {l}: dL/d h_in")
        for k in [k for k in st if k.startswith(f"transformer_layers.{l}.")]:
            check(grads[k], st[k].grad, f"block {l}: grad {k}", scale=scalar_scale(taps, k, st))
    # ---- mapping backward: the fp32 sum of dw over all 2L+1 uses -> weight (bf16 cast of the sum) and bias (fp32 sum) ----
    st = fresh()
    wout = bm.gen_mapping(st, z, d)
    wout.backward(dws[0])
    check(grads["mapping_mlp.model.0.0.weight"], st["mapping_mlp.model.0.0.weight"].grad, "mapping: grad weight")
    check(grads["mapping_mlp.model.0.0.bias"], st["mapping_mlp.model.0.0.bias"].grad, "mapping: grad bias")
    _report(f"generator B={B} p={dropout} patch={patch}", worst)
    # every SLN scalar against the 2-norm of its terms (all 2L + 1 sites, gamma and beta)
    assert len(sln_rows) == 2 * (2 * L + 1), len(sln_rows)
    print(f"SLN scalars, generator B={B} p={dropout} patch={patch}: name, HIP, model, 2-norm of terms, |diff| / norm")
    for k, got, ref, nrm, rel in sln_rows:
        print(f"  {k:44s} {got:+.5e} {ref:+.5e} {nrm:.4e} {rel:.2e}")
    bad = [(k, f"{rel:.2e}") for k, _, _, _, rel in sln_rows if rel > SLN_TIGHT]
    assert not bad, f"SLN scalar gradients beyond 2^-7 of the 2-norm of their terms: {bad}"
    # the staged backward is the one-shot backward, bit for bit
    G2 = torch.zeros_like(P)
    net2 = _lib.VgGenNet(gd, P.data_ptr(), Pb.data_ptr(), G2.data_ptr(), dropout, seed, None, None)
    u.call("vg_gen_forward", C.byref(net2), B, u.ptr(Zd), u.ptr(ws), u.ptr(img), u.stream())
    u.call("vg_gen_backward", C.byref(net2), B, u.ptr(ws), u.ptr(Rd), u.stream())
    u.sync()
    assert torch.equal(G, G2), "staged generator backward must equal the one-shot backward bitwise"
