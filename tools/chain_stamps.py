#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the row-chain kernel (a `make var SRC=chain NAME=chd64 DEFS=-DCH_DBG=64` build):
per stage the cycles a wave spends before the barrier (LDS-DMA wait + barrier skew) and in the stage body.
VITGAN_HIP_LIB=vit-gan_amd/libvitgan_hip_chd64.so python tools/chain_stamps.py [M]"""
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
E, HID, S = 384, 768, 48
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


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
nwg = min(256, (M // 16 + 7) // 8)
stamps = torch.zeros(nwg, 8, 104, dtype=torch.int64, device="cuda")
L.vg_chain_dbg_stamps.restype = None
L.vg_chain_dbg_stamps.argtypes = [C.c_void_p]
L.vg_chain_dbg_stamps(p(stamps))


def run():
    _lib.check(L.vg_encoder_mlp_fwd(p(xn), p(img), p(b1), p(b2), p(res), p(a1), p(z8), p(y), p(yn), p(mean), p(rstd), p(gam), p(bet),
                                    M, 1e-5, 0.1, 1, 3, None, st), "chain")


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
t = stamps.cpu().numpy()
print(f"M={M}: {us:.1f} us for the launch (event pair)")
for wg in (0, nwg // 2, nwg - 1):
    for w in (0, 5):
        r = t[wg, w]
        t0 = r[0]
        total = r[2 * S + 1] - t0
        wait = [int(r[2 * s + 1] - r[2 * s]) for s in range(S)]
        body = [int(r[2 * s + 2] - r[2 * s + 1]) for s in range(S)]
        print(f"wg {wg} wave {w}: {total} ticks in all; stream {r[2 * S] - t0}, epilogue {r[2 * S + 1] - r[2 * S]}")
        print("   wait :", " ".join(f"{x:4d}" for x in wait))
        print("   body :", " ".join(f"{x:4d}" for x in body))
# the s_memtime tick: derive from the launch time of the slowest workgroup
tot = (t[:, :, 2 * S + 1] - t[:, :, 0]).max()
print(f"longest wave: {tot} ticks; if the launch is ~{us:.0f} us that is <= {tot / us:.0f} ticks per us")
