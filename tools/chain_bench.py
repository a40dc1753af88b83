#!/usr/bin/env python3
"""Micro-benchmark of the row-chain kernels (csrc/chain.hip) against the launches they replace, same process, same box:
the MLP half of an encoder block  fc1 + GELU (vg_linear_gelu_fwd, gemm_wr.hip)  +  fc2 + dropout + residual + LayerNorm
(vg_linear_ln_fwd, gemm_row.hip)  versus  vg_encoder_mlp_fwd.  MS=33280,16640 REPS=20 python tools/chain_bench.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_gan_amd  # noqa: F401
from vit_gan_amd import _lib

L = _lib.lib()
BF = torch.bfloat16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
reps = int(os.environ.get("REPS", "20"))
E, HID = 384, 768


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def timeit(fn):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return t[len(t) // 2]


def back_to_back(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M in [int(m) for m in os.environ.get("MS", "32768,33280,16384,16640").split(",")]:
    xn = torch.randn(M, E, device="cuda").to(BF)
    w1 = (torch.randn(HID, E, device="cuda") * 0.05).to(BF); w2 = (torch.randn(E, HID, device="cuda") * 0.04).to(BF)
    b1 = torch.zeros(HID, device="cuda"); b2 = torch.zeros(E, device="cuda")
    res = torch.randn(M, E, device="cuda").to(BF)
    a1 = torch.empty(M, HID, device="cuda", dtype=BF); z8 = torch.empty(M, HID, device="cuda", dtype=torch.uint8)
    y = torch.empty(M, E, device="cuda", dtype=BF); yn = torch.empty(M, E, device="cuda", dtype=BF)
    mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    gam = torch.ones(E, device="cuda"); bet = torch.zeros(E, device="cuda")
    img = torch.empty(L.vg_encoder_mlp_image_elems(), device="cuda", dtype=BF)
    _lib.check(L.vg_encoder_mlp_pack(p(w1), p(w2), p(img), st), "pack")
    w2p = torch.empty(E * HID, device="cuda", dtype=BF)
    _lib.check(L.vg_row_pack_weight(p(w2), HID, HID, 0, p(w2p), st), "rowpack")

    def chain():
        _lib.check(L.vg_encoder_mlp_fwd(p(xn), p(img), p(b1), p(b2), p(res), p(a1), p(z8), p(y), p(yn), p(mean), p(rstd), p(gam), p(bet),
                                        M, 1e-5, 0.1, 1, 3, None, st), "chain")

    def fc1():
        _lib.check(L.vg_linear_gelu_fwd(p(xn), p(w1), p(b1), p(a1), p(z8), M, HID, E, st), "fc1")

    def fc2():
        _lib.check(L.vg_linear_ln_fwd(p(a1), p(w2p), p(b2), p(res), p(y), p(yn), p(mean), p(rstd), p(gam), p(bet), M, HID, 1e-5, 0.1, 1, 3, None, st), "fc2")

    def pair():
        fc1(); fc2()

    # the chain with the out-projection + residual + norm2 in front, against its three launches
    ao = torch.randn(M, E, device="cuda").to(BF); xin = torch.randn(M, E, device="cuda").to(BF)
    wo = (torch.randn(E, E, device="cuda") * 0.05).to(BF); bo = torch.zeros(E, device="cuda")
    xmid = torch.empty(M, E, device="cuda", dtype=BF); xn2 = torch.empty(M, E, device="cuda", dtype=BF)
    m2 = torch.empty(M, device="cuda"); r2 = torch.empty(M, device="cuda")
    img2 = torch.empty(L.vg_encoder_post_attention_image_elems(), device="cuda", dtype=BF)
    _lib.check(L.vg_encoder_post_attention_pack(p(wo), p(w1), p(w2), p(img2), st), "pack2")
    wop = torch.empty(E * E, device="cuda", dtype=BF)
    _lib.check(L.vg_row_pack_weight(p(wo), E, E, 0, p(wop), st), "rowpack")

    def chain2():
        _lib.check(L.vg_encoder_post_attention_fwd(p(ao), p(xin), p(img2), p(bo), p(b1), p(b2), p(gam), p(bet), p(gam), p(bet), p(xmid), p(xn2), p(m2), p(r2),
                                                   p(a1), p(z8), p(y), p(yn), p(mean), p(rstd), M, 1e-5, 0.1, 1, 2, 3, None, st), "chain2")

    def outp():
        _lib.check(L.vg_linear_ln_fwd(p(ao), p(wop), p(bo), p(xin), p(xmid), p(xn2), p(m2), p(r2), p(gam), p(bet), M, E, 1e-5, 0.1, 1, 2, None, st), "outp")

    def triple():
        outp(); fc1(); fc2()

    t_c2, t_0 = timeit(chain2), timeit(outp)
    b_c2, b_t = back_to_back(chain2), back_to_back(triple)
    print(f"M={M}: out-projection + norm2 + MLP chain {t_c2:.1f} us (back to back {b_c2:.1f})   out-proj+ln {t_0:.1f} + fc1 + fc2 (three launches back to back {b_t:.1f})", flush=True)
    t_c, t_1, t_2 = timeit(chain), timeit(fc1), timeit(fc2)
    b_c, b_p = back_to_back(chain), back_to_back(pair)
    mb = M * (E * 2 * 4 + HID * 3) / 1e6
    print(f"M={M}: chain {t_c:.1f} us (back to back {b_c:.1f})   fc1+gelu {t_1:.1f} + fc2+ln {t_2:.1f} = {t_1 + t_2:.1f} (pair back to back {b_p:.1f})"
          f"   chain: {mb / t_c / 1e3:.2f} TB/s of {mb:.0f} MB, {4 * M * E * HID / t_c / 1e6:.0f} TFLOP/s", flush=True)
