"""Attention double backward: write the outputs of the loaded library to a file / compare with a saved file (A/B across builds) and time it."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.getcwd())
import torch
import vit_gan_amd  # noqa: F401
from vit_gan_amd import _lib
L = _lib.lib()
out = {}
for (B, H, S, HE) in ((256, 4, 65, 96), (16, 8, 65, 64), (8, 12, 65, 64), (8, 4, 33, 32), (3, 4, 17, 96)):
    g = torch.Generator().manual_seed(B + S)
    E = H * HE
    qkv = torch.randn(B * S, 3 * E, generator=g).to(torch.bfloat16).cuda()
    do = torch.randn(B * S, E, generator=g).to(torch.bfloat16).cuda()
    u = torch.randn(B * S, 3 * E, generator=g).to(torch.bfloat16).cuda()
    o = torch.empty(B * S, E, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B * H * S, device="cuda")
    ddo = torch.zeros_like(do); dq2 = torch.zeros_like(qkv)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    sc = 1 / math.sqrt(HE)
    _lib.check(L.vg_attention_fwd(p(qkv), p(o), p(lse), B, H, S, HE, sc, st), "fwd")
    fn = lambda: _lib.check(L.vg_attention_bwd_bwd(p(qkv), p(do), p(lse), p(u), p(ddo), p(dq2), B, H, S, HE, sc, st), "bwd_bwd")
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} H={H} S={S} HE={HE}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us")
    out[(B, H, S, HE)] = (ddo.float().cpu(), dq2.float().cpu())
f = sys.argv[1]
if os.path.exists(f):
    ref = torch.load(f)
    for k, (a, b) in out.items():
        ra, rb = ref[k]
        print(k, "d(dO) max|diff|/max|ref| %.2e" % (float((a - ra).abs().max()) / float(ra.abs().max())), "d(qkv) %.2e" % (float((b - rb).abs().max()) / float(rb.abs().max())),
              "finite", bool(torch.isfinite(a).all() and torch.isfinite(b).all()))
else:
    torch.save(out, f)
