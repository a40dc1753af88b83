#!/usr/bin/env python3
"""Micro-benchmark of the full-row kernels (csrc/gemm_row.hip): time against the contraction depth K at the C2 row counts.
The slope over K is the time per 32-deep stage; the intercept is prologue + epilogue.  Diagnostic libraries (make var
SRC=gemm_row NAME=rowdbg, VITGAN_HIP_LIB=...) read VG_ROW_DBG: 1 = A from L2 (lda 0), 2 = W stage 0 re-read, 4 = A stage 0 re-read."""
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
E = 384


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


tag = os.environ.get("VG_ROW_DBG", "0")
for M in [int(m) for m in os.environ.get("MS", "33280,16640").split(",")]:
    row = []
    for K in (384, 768, 1152, 2304):
        a = torch.randn(M, K, device="cuda").to(BF)
        w = (torch.randn(E, K, device="cuda") * 0.05).to(BF)
        wp = torch.empty(E * K, device="cuda", dtype=BF)
        _lib.check(L.vg_row_pack_weight(p(w), K, K, 0, p(wp), st), "pack")
        bias = torch.zeros(E, device="cuda"); res = torch.randn(M, E, device="cuda").to(BF)
        y = torch.empty(M, E, device="cuda", dtype=BF); yn = torch.empty(M, E, device="cuda", dtype=BF)
        mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
        gam = torch.ones(E, device="cuda"); bet = torch.zeros(E, device="cuda")
        t_plain = timeit(lambda: L.vg_linear_ln_fwd(p(a), p(wp), p(bias), p(res), p(y), None, None, None, None, None, M, K, 1e-5, 0.0, 0, 0, None, st))
        t_ln = timeit(lambda: L.vg_linear_ln_fwd(p(a), p(wp), p(bias), p(res), p(y), p(yn), p(mean), p(rstd), p(gam), p(bet), M, K, 1e-5, 0.1, 1, 3, None, st))
        x = torch.randn(M, E, device="cuda").to(BF); gres = torch.randn(M, E, device="cuda").to(BF)
        dx = torch.empty(M, E, device="cuda", dtype=BF); dxm = torch.empty(M, E, device="cuda", dtype=BF)
        part = torch.empty(L.vg_row_parts(M), 3 * E, device="cuda")
        t_bwd = timeit(lambda: L.vg_linear_dgrad_ln_bwd(p(a), p(wp), p(x), p(mean), p(rstd), p(gam), p(gres), p(dx), p(dxm), p(part), M, K, 0.1, 1, 3, None, st))
        row.append((K, t_plain, t_ln, t_bwd))
        del a, w, wp
    print(f"dbg={tag} M={M}: " + "  ".join(f"K={k}: fwd {tp:.1f} fwd+ln {tl:.1f} bwd+ln {tb:.1f}" for k, tp, tl, tb in row))
    (k0, a0, _, _), (k1, a1, _, _) = row[0], row[-1]
    print(f"   per 32-deep stage: {(a1 - a0) / ((k1 - k0) / 32) * 1e3:.0f} ns; intercept (K -> 0): {a0 - (a1 - a0) / (k1 - k0) * k0:.1f} us")
