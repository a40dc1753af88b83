#!/usr/bin/env python3
"""Calibration, not product code: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the Linear shapes of one
ViT block of the C2 step (fused 2B pass, M = 33 280), plain GEMM without any epilogue, per-dispatch HIP event pairs like
tools/gemm_bench.py - the ceiling a hand-written kernel with a fused epilogue is compared against in DESIGN.md s5."""
import torch

M = 33280
BF = torch.bfloat16
shapes = [("qkv fwd           [M,384]x[384,1152]", (M, 384), (384, 1152), False),
          ("out fwd           [M,384]x[384,384] ", (M, 384), (384, 384), False),
          ("fc1 fwd           [M,384]x[384,768] ", (M, 384), (384, 768), False),
          ("fc2 fwd           [M,768]x[768,384] ", (M, 768), (768, 384), False),
          ("qkv dgrad         [M,1152]x[1152,384]", (M, 1152), (1152, 384), False),
          ("qkv wgrad  [1152,M]x[M,384] (A^T)    ", (M, 1152), (M, 384), True),
          ("fc1 wgrad  [768,M]x[M,384] (A^T)     ", (M, 768), (M, 384), True)]
for name, sa, sb, ta in shapes:
    a = torch.randn(*sa, device="cuda").to(BF)
    b = (torch.randn(*sb, device="cuda") * 0.05).to(BF)
    fn = (lambda: torch.matmul(a.t(), b)) if ta else (lambda: torch.matmul(a, b))
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    torch.cuda.synchronize()
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    us = t[len(t) // 2]
    k = sa[0] if ta else sa[1]
    m, n = (sa[1], sb[1]) if ta else (sa[0], sb[1])
    print(f"{name}: {us:7.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s")
