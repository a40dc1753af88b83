#!/usr/bin/env python3
"""Micro-benchmark of the GEMM launches of one ViT block of the C2 step, through the C ABI, each on the kernel instantiation the
engine really uses for it (round 3: the full-row kernels of csrc/gemm_row.hip for every Linear whose output is the embedding).
Run under rocprofv3 for PMC (tools/pmc_traffic.py recovers the shapes by position from SHAPES below).

  name                         entry point                  kernel
  wr qkv fwd                   vg_linear_fwd                vg_gemm_wr_kernel<0,0,0>
  row out+res+ln fwd           vg_linear_ln_fwd  (K=384)    vg_gemm_row_kernel<0>
  wr fc1+gelu+gelu' bytes      vg_linear_gelu_fwd           vg_gemm_wr_kernel<0,1,2>
  row fc2+res+ln fwd           vg_linear_ln_fwd  (K=768)    vg_gemm_row_kernel<0>
  wr fc2 dgrad*stored          vg_linear_dgrad mul=8        vg_gemm_wr_kernel<1,8,0>
  row fc1 dgrad+ln bwd         vg_linear_dgrad_ln_bwd K=768 vg_gemm_row_kernel<1>
  wr out dgrad                 vg_linear_dgrad              vg_gemm_wr_kernel<1,0,0>
  row qkv dgrad+ln bwd         vg_linear_dgrad_ln_bwd K=1152 vg_gemm_row_kernel<1>
  tn qkv/fc1/fc2/out wgrad     vg_linear_wgrad              vg_gemm_tn384_kernel<6>
"""
import ctypes as C
import os
import sys

E = 384
B = int(os.environ.get("B", "512"))
M = int(os.environ.get("M", B * 65))
# name, flops, algorithmic bytes (every operand once, bf16 unless fp32 statistics / slabs are named)
SHAPES = [
    ("wr qkv fwd", 2.0 * M * 1152 * 384, 2 * (M * 384 + 1152 * 384 + M * 1152)),
    ("row out+res+ln fwd", 2.0 * M * 384 * 384, 2 * (M * 384 + 384 * 384 + 3 * M * 384) + 8 * M),
    ("wr fc1+gelu+gelu' bytes", 2.0 * M * 768 * 384, 2 * (M * 384 + 768 * 384 + M * 768) + M * 768),  # gelu' as one byte per element
    ("row fc2+res+ln fwd", 2.0 * M * 384 * 768, 2 * (M * 768 + 384 * 768 + 3 * M * 384) + 8 * M),
    ("wr fc2 dgrad*stored", 2.0 * M * 768 * 384, 2 * (M * 384 + 768 * 384 + M * 768) + M * 768),
    ("row fc1 dgrad+ln bwd", 2.0 * M * 384 * 768, 2 * (M * 768 + 384 * 768 + 4 * M * 384) + 8 * M),
    ("wr out dgrad", 2.0 * M * 384 * 384, 2 * (M * 384 + 384 * 384 + M * 384)),
    ("row qkv dgrad+ln bwd", 2.0 * M * 384 * 1152, 2 * (M * 1152 + 384 * 1152 + 4 * M * 384) + 8 * M),
    ("tn qkv wgrad", 2.0 * M * 1152 * 384, 2 * (M * 1152 + M * 384) + 4 * 28 * 1152 * 384),   # + the fp32 K-slice slabs it writes
    ("tn fc1 wgrad", 2.0 * M * 768 * 384, 2 * (M * 768 + M * 384) + 4 * 42 * 768 * 384),
    ("tn fc2 wgrad", 2.0 * M * 384 * 768, 2 * (M * 384 + M * 768) + 4 * 42 * 384 * 768),
    ("tn out wgrad", 2.0 * M * 384 * 384, 2 * (M * 384 + M * 384) + 4 * 64 * 384 * 384),
]


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import vit_gan_amd  # noqa: F401
    from vit_gan_amd import _lib
    L = _lib.lib()
    BF = torch.bfloat16
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    reps = int(os.environ.get("REPS", "30"))

    def p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def rnd(*shape, s=1.0):
        return (torch.randn(*shape, device="cuda") * s).to(BF)

    def timeit(fn, idx):
        name, flops, byts = SHAPES[idx]
        for _ in range(3):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        torch.cuda.synchronize()
        for e0, e1 in ev:
            e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
        us = t[len(t) // 2]
        print(f"{name:24s} M={M}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {byts / us / 1e3:7.1f} GB/s (algorithmic)")

    def packed(w, K, tr):
        wp = torch.empty(E * K, device="cuda", dtype=BF)
        _lib.check(L.vg_row_pack_weight(p(w), w.shape[1], K, tr, p(wp), st), "pack")
        return wp

    x384, x768, x1152 = rnd(M, 384), rnd(M, 768), rnd(M, 1152)
    res = rnd(M, 384)
    gam, bet, bias384 = torch.ones(E, device="cuda"), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    mean, rstd = torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")
    o384, o384b, o768, o768b, o1152 = (torch.empty(M, n, device="cuda", dtype=BF) for n in (384, 384, 768, 768, 1152))
    wqkv, wo, w1, w2 = rnd(1152, 384, s=0.05), rnd(384, 384, s=0.05), rnd(768, 384, s=0.05), rnd(384, 768, s=0.05)
    b1152, b768 = torch.zeros(1152, device="cuda"), torch.zeros(768, device="cuda")
    wo_p, w2_p, wqkv_t, w1_t = packed(wo, 384, 0), packed(w2, 768, 0), packed(wqkv, 1152, 1), packed(w1, 768, 1)
    part = torch.empty(L.vg_row_parts(M), 3 * E, device="cuda")
    timeit(lambda: L.vg_linear_fwd(p(x384), p(wqkv), p(b1152), None, p(o1152), None, None, M, 1152, 384, 0, 0.0, st), 0)
    timeit(lambda: L.vg_linear_ln_fwd(p(x384), p(wo_p), p(bias384), p(res), p(o384), p(o384b), p(mean), p(rstd), p(gam), p(bet), M, 384, 1e-5, 0.1, 1, 1, None, st), 1)
    code = torch.empty(M, 768, device="cuda", dtype=torch.uint8)
    timeit(lambda: L.vg_linear_gelu_fwd(p(x384), p(w1), p(b768), p(o768), p(code), M, 768, 384, st), 2)
    timeit(lambda: L.vg_linear_ln_fwd(p(x768), p(w2_p), p(bias384), p(res), p(o384), p(o384b), p(mean), p(rstd), p(gam), p(bet), M, 768, 1e-5, 0.1, 1, 2, None, st), 3)
    timeit(lambda: L.vg_linear_dgrad(p(x384), p(w2), p(o768), M, 384, 768, 8, p(code), None, 0.0, st), 4)
    timeit(lambda: L.vg_linear_dgrad_ln_bwd(p(x768), p(w1_t), p(x384), p(mean), p(rstd), p(gam), p(res), p(o384), p(o384b), p(part), M, 768, 0.1, 1, 1, None, st), 5)
    timeit(lambda: L.vg_linear_dgrad(p(x384), p(wo), p(o384), M, 384, 384, 0, None, None, 0.0, st), 6)
    timeit(lambda: L.vg_linear_dgrad_ln_bwd(p(x1152), p(wqkv_t), p(x384), p(mean), p(rstd), p(gam), p(res), p(o384), p(o384b), p(part), M, 1152, 0.1, 1, 0, None, st), 7)
    # K slices so that a single problem fills the chip like the engine's grouped launch does (~252 workgroups of 128 x 384 tiles)
    for i, (N, K, splits, dy, x) in enumerate([(1152, 384, 28, x1152, x384), (768, 384, 42, x768, x384), (384, 768, 42, x384, x768), (384, 384, 64, x384, x384)]):
        dw = torch.zeros(N, K, device="cuda"); slab = torch.empty(splits * N * K, device="cuda")
        timeit(lambda: L.vg_linear_wgrad(p(dy), p(x), p(dw), p(slab), slab.numel(), M, N, K, splits, 1, st), 8 + i)


if __name__ == "__main__":
    main()
