#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernel family at the hot-path shapes (run under rocprofv3 for PMC)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_gan_amd
from vit_gan_amd import _lib

L = _lib.lib()
BF = torch.bfloat16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
reps = int(os.environ.get("REPS", "30"))


def timeit(fn, flops, name):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:44s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s")


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


B = int(os.environ.get("B", "512"))
M = int(os.environ.get("M", B * 65))
for (N, K, act, pre, res, nm) in [(1152, 384, 0, False, False, "NT qkv"), (384, 384, 0, False, True, "NT out+res"),
                                  (768, 384, 1, True, False, "NT fc1+gelu+pre"), (384, 768, 0, False, True, "NT fc2+res")]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.zeros(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=BF)
    prb = torch.empty(M, N, device="cuda", dtype=BF) if pre else None
    r = torch.randn(M, N, device="cuda").to(BF) if res else None
    timeit(lambda: L.vg_linear_fwd(p(a), p(w), p(bias), p(r), p(out), p(prb), None, M, N, K, act, 0.0, st), 2.0 * M * N * K, f"{nm} M={M} N={N} K={K}")
for (N, K, mul, nm) in [(384, 768, 4, "NN fc2 dgrad*gelu'"), (768, 384, 0, "NN fc1 dgrad"), (1152, 384, 0, "NN qkv dgrad"), (384, 384, 0, "NN out dgrad")]:
    dy = torch.randn(M, N, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    z = torch.randn(M, K, device="cuda").to(BF); dx = torch.empty(M, K, device="cuda", dtype=BF)
    timeit(lambda: L.vg_linear_dgrad(p(dy), p(w), p(dx), M, N, K, mul, p(z), None, 0.0, st), 2.0 * M * N * K, f"{nm} M={M} N={N} K={K}")
for (N, K, splits, nm) in [(1152, 384, 4, "TN qkv wgrad"), (768, 384, 6, "TN fc1 wgrad"), (384, 768, 6, "TN fc2 wgrad"), (384, 384, 8, "TN out wgrad")]:
    dy = torch.randn(M, N, device="cuda").to(BF); x = torch.randn(M, K, device="cuda").to(BF)
    dw = torch.zeros(N, K, device="cuda"); slab = torch.empty(splits * N * K, device="cuda")
    timeit(lambda: L.vg_linear_wgrad(p(dy), p(x), p(dw), p(slab), slab.numel(), M, N, K, splits, 1, st), 2.0 * M * N * K, f"{nm} M={M} N={N} K={K} s={splits}")
