#!/usr/bin/env python3
"""Micro-benchmark of the fused attention kernels at the hot-path shape (B images x 4 heads x 65 tokens x 96)."""
import ctypes as C
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_gan_amd  # noqa: F401
from vit_gan_amd import _lib

L = _lib.lib()
B = int(os.environ.get("B", "512")); H = int(os.environ.get("H", "4")); S = int(os.environ.get("S", "65")); HE = int(os.environ.get("HE", "96"))
E = H * HE
reps = int(os.environ.get("REPS", "30"))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
qkv = torch.randn(B * S, 3 * E, device="cuda").to(torch.bfloat16)
o = torch.empty(B * S, E, device="cuda", dtype=torch.bfloat16)
do = torch.randn(B * S, E, device="cuda").to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
lse = torch.empty(B * H * S, device="cuda")
scale = 1.0 / math.sqrt(HE)
p = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, nbytes, name):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:28s} B={B} H={H} S={S} HE={HE} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s algorithmic")


nq, no = qkv.numel() * 2, o.numel() * 2
timeit(lambda: _lib.check(L.vg_attention_fwd(p(qkv), p(o), p(lse), B, H, S, HE, scale, st), "fwd"), nq + no, "attention fwd")
timeit(lambda: _lib.check(L.vg_attention_bwd(p(qkv), p(o), p(do), p(lse), p(dqkv), B, H, S, HE, scale, st), "bwd"), 2 * nq + 2 * no, "attention bwd")
