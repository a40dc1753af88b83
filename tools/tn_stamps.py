#!/usr/bin/env python3
"""Cycle breakdown of a stage of the 128x384 weight-gradient kernel from in-kernel s_memtime stamps.
Needs the diagnostic build:  make -C vit-gan_amd/csrc var SRC=gemm_tn NAME=tnst DEFS=-DVG_TN_STAMPS  (the product carries none)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, N, K, SPL = 33280, int(os.environ.get("N", "1152")), 384, int(os.environ.get("WG_SPLITS", "10"))
nwg = (N // 128) * (K // 384) * SPL
stamps = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device="cuda")
os.environ["VG_STAMP_PTR"] = hex(stamps.data_ptr())
L = C.CDLL(os.path.join(root, "vit-gan_amd", "libvitgan_hip_tnst.so"))
P = C.c_void_p
L.vg_linear_wgrad.argtypes = [P, P, P, P, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]
BF = torch.bfloat16
dy = torch.randn(M, N, device="cuda").to(BF); x = torch.randn(M, K, device="cuda").to(BF)
dw = torch.empty(N, K, device="cuda"); slab = torch.empty(SPL * N * K, device="cuda")
st = P(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    L.vg_linear_wgrad(P(dy.data_ptr()), P(x.data_ptr()), P(dw.data_ptr()), P(slab.data_ptr()), slab.numel(), M, N, K, SPL, 0, st)
torch.cuda.synchronize()
t = stamps.cpu().numpy().reshape(nwg, 8, 8).astype(np.float64)
steps = t[:, :, 5]
per = t[:, :, :5] / steps[:, :, None]
print(f"N {N}: {nwg} workgroups, {int(steps[0,0])} stages each; cycles per stage (mean over waves): "
      f"vmcnt+barrier {per[:,:,0].mean():.0f} | before MFMA {per[:,:,1].mean():.0f} | MFMA {per[:,:,2].mean():.0f} | "
      f"MFMA {per[:,:,3].mean():.0f} | total {per[:,:,4].mean():.0f}")
for w in range(8):
    print(f"  wave {w}: " + " ".join(f"{per[:, w, i].mean():7.0f}" for i in range(5)))
