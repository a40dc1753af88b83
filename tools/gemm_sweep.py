#!/usr/bin/env python3
"""Forward-GEMM time against the reduction length K at the hot-path M, N: separates the per-launch fixed cost
(intercept) from the per-k-step cost (slope)."""
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
M = int(os.environ.get("M", "33280"))
N = int(os.environ.get("N", "1152"))
for K in (32, 128, 384, 768, 1536, 3072):
    a = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)

    def f():
        _lib.check(L.vg_linear_fwd(a.data_ptr(), w.data_ptr(), None, None, out.data_ptr(), None, None, M, N, K, 0, 0.0, st), "vg_linear_fwd")
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"M={M} N={N} K={K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  per k-step {us / (K / 32):6.2f} us")
