#!/usr/bin/env python3
"""Per-workgroup timeline of the GEMM kernel from in-kernel s_memrealtime stamps.

Needs the diagnostic build (`make -C vit-gan_amd/csrc dbg` -> libvitgan_hip_dbg.so, compiled with -DVG_STAMPS); the
product library carries no stamps.  Runs the QKV-forward shape (M = 33280, N = 1152, K from $K, default 384) and prints
the median time a workgroup spends in each segment.  VG_GEMM_WM=2 forces 128-row tiles, VERBOSE=1 prints every segment.
"""
import ctypes as C
import os

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(root, "vit-gan_amd", "libvitgan_hip_dbg.so"))
BF = torch.bfloat16
M, N, K = 33280, 1152, int(os.environ.get("K", "384"))
a = torch.randn(M, K, device="cuda").to(BF)
w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
out = torch.empty(M, N, device="cuda", dtype=BF)
rows = 128 if os.environ.get("VG_GEMM_WM") == "2" else 256
nwg = ((M + rows - 1) // rows) * (N // 128)
stamps = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
os.environ["VG_STAMP_PTR"] = hex(stamps.data_ptr())  # read by the diagnostic launcher
L.vg_linear_fwd.argtypes = [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_float, C.c_void_p]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    L.vg_linear_fwd(a.data_ptr(), w.data_ptr(), None, None, out.data_ptr(), None, None, M, N, K, 0, 0.0, s)
torch.cuda.synchronize()
t = stamps.cpu().numpy().reshape(nwg, 8).astype(np.float64)
rel = (t - t[:, 0].min()) / 100.0  # us: s_memrealtime ticks at 100 MHz
d = np.diff(t, axis=1) / 100.0
print("K", K, "tile rows", rows, "workgroups", nwg, "| span us", round(float(rel[:, 7].max()), 1), "| main loop per k-step us",
      round(float(np.median(d[:, 3])) / (K / 32), 3), "| epilogue us", round(float(np.median(d[:, 4] + d[:, 5] + d[:, 6])), 2))
if os.environ.get("VERBOSE"):
    for i, n in enumerate(["setup", "prologue DMA", "first wait", "main loop", "ep: -", "ep: loads", "ep: body"]):
        print(f"{n:14s} mean {d[:, i].mean():7.2f} us  p50 {np.median(d[:, i]):7.2f}  max {d[:, i].max():7.2f}")
    print("workgroup lifetime mean", round(float((t[:, 7] - t[:, 0]).mean() / 100.0), 2), "us")
