#!/usr/bin/env python3
"""Diagnostic: per-tensor error of the HIP ViT / generator passes against both parity tiers (the fp32 oracle and the
rounding-faithful bf16 model) as a function of depth.  Prints max|err| / max|ref| per tensor; run on the GPU box."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ is one of the places allowed to import the oracle
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import gpu_util as u  # noqa: E402
from weights import make_input, make_state  # noqa: E402
from oracle import bf16_model as bm, gen_oracle as go, vit_oracle as vo  # noqa: E402
from vit_gan_amd import _lib, flat  # noqa: E402


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


def vit(layers, B=2, embed=384, heads=4, image=32, patch=4):
    d = vo.VitDims(embed=embed, heads=heads, layers=layers, classes=1, image=image, patch=patch)
    st_np = make_state(vo.vit_param_shapes(d), 11, "vit")
    x = torch.from_numpy(make_input((B, 3, image, image), 11, "uniform"))
    refs = []
    for fwd in (vo.vit_forward, bm.vit_forward):
        st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in st_np.items()}
        xr = x.clone().requires_grad_(True)
        out = fwd(st, xr, d)
        R = torch.from_numpy(make_input(tuple(out.shape), 12))
        (out * R).sum().backward()
        refs.append((out, xr.grad, {k: p.grad for k, p in st.items()}))
    dd = flat.vit_dims_struct(3, image, patch, embed, heads, layers, 2, 1)
    lay, slots = flat.vit_layout(dd), flat.vit_slots(dd)
    P = flat.pack(slots, lay.total, st_np, device="cuda")
    Pb, G = P.to(torch.bfloat16), torch.zeros_like(P)
    net = _lib.VgVitNet(dd, P.data_ptr(), Pb.data_ptr(), G.data_ptr(), 0.0, 0, None, None)
    ws = torch.empty(_lib.lib().vg_vit_ws_bytes(C.byref(dd), B), dtype=torch.uint8, device="cuda")
    logits = torch.empty(B, 1, device="cuda")
    X, Rd = x.cuda(), R.cuda()
    u.call("vg_vit_forward", C.byref(net), B, u.ptr(X), 0, u.ptr(ws), u.ptr(logits), u.stream())
    dimg = torch.empty(B, 3, image, image, dtype=torch.bfloat16, device="cuda")
    u.call("vg_vit_backward", C.byref(net), B, u.ptr(ws), u.ptr(Rd), u.ptr(dimg), 1, u.stream())
    u.sync()
    grads = flat.unpack(slots, G)
    print(f"--- ViT L={layers} E={embed} H={heads} image {image} patch {patch}: tensor | vs fp32 oracle | vs bf16 model")
    print(f"logits {rel(logits, refs[0][0]):.2e} {rel(logits, refs[1][0]):.2e}")
    print(f"d_img  {rel(dimg, refs[0][1]):.2e} {rel(dimg, refs[1][1]):.2e}")
    rows = []
    for k in refs[0][2]:
        if float(refs[0][2][k].abs().max()) < 1e-6:
            continue
        rows.append((rel(grads[k], refs[1][2][k]), rel(grads[k], refs[0][2][k]), k))
    rows.sort(reverse=True)
    for t, l, k in rows[:8]:
        print(f"  {k:50s} {l:.2e} {t:.2e}")
    print(f"  median over {len(rows)} tensors: fp32 {sorted(r[1] for r in rows)[len(rows) // 2]:.2e}  model {sorted(r[0] for r in rows)[len(rows) // 2]:.2e}")


def gen(layers, B=2):
    d = go.GenDims(layers=layers)
    st_np = make_state(go.gen_param_shapes(d), 21, "gen")
    z = torch.from_numpy(make_input((B, d.latent), 21))
    refs = []
    for fwd in (go.gen_forward, bm.gen_forward):
        st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in st_np.items()}
        out = fwd(st, z, d)
        R = torch.from_numpy(make_input(tuple(out.shape), 22)).to(torch.bfloat16).float()
        (out * R).sum().backward()
        refs.append((out, {k: p.grad for k, p in st.items()}))
    gd = _lib.VgGenDims(d.latent, d.tokens, d.embed, d.heads, d.layers, d.siren_hidden, d.out_features, d.omega0, 0, 3, 32)
    lay, slots = flat.gen_layout(gd), flat.gen_slots(gd)
    P = flat.pack(slots, lay.total, st_np, device="cuda")
    Pb, G = P.to(torch.bfloat16), torch.zeros_like(P)
    net = _lib.VgGenNet(gd, P.data_ptr(), Pb.data_ptr(), G.data_ptr(), 0.0, 0, None, None)
    ws = torch.empty(_lib.lib().vg_gen_ws_bytes(C.byref(gd), B), dtype=torch.uint8, device="cuda")
    img = torch.empty(B, 3, 32, 32, dtype=torch.bfloat16, device="cuda")
    Zd, Rd = z.cuda(), R.to(torch.bfloat16).cuda()
    u.call("vg_gen_forward", C.byref(net), B, u.ptr(Zd), u.ptr(ws), u.ptr(img), u.stream())
    u.call("vg_gen_backward", C.byref(net), B, u.ptr(ws), u.ptr(Rd), u.stream())
    u.sync()
    grads = flat.unpack(slots, G)
    print(f"--- generator L={layers}: tensor | vs fp32 oracle | vs bf16 model")
    print(f"image {rel(img, refs[0][0]):.2e} {rel(img, refs[1][0]):.2e}   frac of pixels off by > 2^-7: "
          f"{float(((img.float().cpu() - refs[1][0].detach()).abs() > 2 ** -7).float().mean()):.4f}")
    rows = sorted(((rel(grads[k], refs[1][1][k]), rel(grads[k], refs[0][1][k]), k) for k in refs[0][1]), reverse=True)
    for t, l, k in rows[:8]:
        print(f"  {k:50s} {l:.2e} {t:.2e}")
    print(f"  median over {len(rows)} tensors: fp32 {sorted(r[1] for r in rows)[len(rows) // 2]:.2e}  model {sorted(r[0] for r in rows)[len(rows) // 2]:.2e}")


if __name__ == "__main__":
    import sys as _s
    if len(_s.argv) > 1 and _s.argv[1] == "c5":
        vit(1, B=1, embed=768, heads=12, image=128, patch=16)
        vit(1, B=1, embed=768, heads=12, image=32, patch=4)
        vit(1, B=1, embed=384, heads=4, image=128, patch=16)
        vit(1, B=2, embed=512, heads=8, image=64, patch=8)
    else:
        for L in (1, 2, 6):
            vit(L)
        vit(6, embed=128)
        for L in (1, 4):
            gen(L)
