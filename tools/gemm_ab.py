#!/usr/bin/env python3
"""A/B of GEMM builds in ONE process (cdna_hip_programming.md rule 24): libraries given on the command line are loaded
side by side and timed in interleaved rounds at the hot-path shapes; prints median / min per shape and build.

    python tools/gemm_ab.py vit-gan_amd/libvitgan_hip.so vit-gan_amd/libvitgan_hip_<name>.so [...]
Only vg_linear_fwd / vg_linear_dgrad are called (their signatures are stable across ABI versions)."""
import ctypes as C
import os
import statistics
import sys

import torch  # noqa: F401  (first: one HIP runtime for torch and the libraries)

BF = torch.bfloat16
P = C.c_void_p
libs = []
for path in sys.argv[1:]:
    h = C.CDLL(os.path.abspath(path))
    h.vg_linear_fwd.argtypes = [P] * 7 + [C.c_int] * 4 + [C.c_float, P]
    h.vg_linear_dgrad.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, P, P, C.c_float, P]
    h.vg_linear_wgrad.argtypes = [P, P, P, P, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]  # ABI >= 4
    libs.append((os.path.basename(path), h))
st = P(torch.cuda.current_stream().cuda_stream)
ROUNDS, REPS = int(os.environ.get("ROUNDS", "7")), int(os.environ.get("REPS", "20"))


def p(t):
    return None if t is None else P(t.data_ptr())


def bench(name, calls, flops):
    res = {n: [] for n, _ in libs}
    for n, fn in calls:
        for _ in range(3):
            fn()
    for _ in range(ROUNDS):
        for n, fn in calls:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / REPS * 1e3)
    line = f"{name:34s}"
    for n, _ in libs:
        med, mn = statistics.median(res[n]), min(res[n])
        line += f" | {n[-18:]:>18s} {med:7.1f} us (min {mn:6.1f}) {flops / med / 1e6:6.0f} TF"
    print(line, flush=True)


for M in (33280, 16640):
    for (N, K, act, pre, res, nm) in [(1152, 384, 0, False, False, "NT qkv"), (384, 384, 0, False, True, "NT out+res"),
                                      (768, 384, 1, True, False, "NT fc1+gelu+pre"), (384, 768, 0, False, True, "NT fc2+res")]:
        a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
        bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=BF)
        prb = torch.empty(M, N, device="cuda", dtype=BF) if pre else None
        r = torch.randn(M, N, device="cuda").to(BF) if res else None
        bench(f"{nm} M={M}", [(n, (lambda h=h: h.vg_linear_fwd(p(a), p(w), p(bias), p(r), p(out), p(prb), None, M, N, K, act, 0.0, st))) for n, h in libs],
              2.0 * M * N * K)
    for (N, K, mul, nm) in [(384, 768, 4, "NN fc2 dgrad*gelu'"), (768, 384, 0, "NN fc1 dgrad"), (1152, 384, 0, "NN qkv dgrad"), (384, 384, 0, "NN out dgrad")]:
        dy = torch.randn(M, N, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
        z = torch.randn(M, K, device="cuda").to(BF); dx = torch.empty(M, K, device="cuda", dtype=BF)
        bench(f"{nm} M={M}", [(n, (lambda h=h: h.vg_linear_dgrad(p(dy), p(w), p(dx), M, N, K, mul, p(z), None, 0.0, st))) for n, h in libs], 2.0 * M * N * K)
    # weight gradients dW[N, K] = dY[M, N]^T X[M, K] (split-K slabs + the fold); WG_SPLITS K slices (default 10)
    SPL = int(os.environ.get("WG_SPLITS", "10"))
    for (N, K, nm) in [(1152, 384, "TN qkv wgrad"), (768, 384, "TN fc1 wgrad"), (384, 768, "TN fc2 wgrad"), (384, 384, "TN out wgrad")]:
        dy = torch.randn(M, N, device="cuda").to(BF); x = torch.randn(M, K, device="cuda").to(BF)
        dw = torch.empty(N, K, device="cuda"); slab = torch.empty(SPL * N * K, device="cuda")
        bench(f"{nm} M={M}", [(n, (lambda h=h: h.vg_linear_wgrad(p(dy), p(x), p(dw), p(slab), slab.numel(), M, N, K, SPL, 0, st))) for n, h in libs],
              2.0 * M * N * K)
