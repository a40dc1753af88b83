"""Print a checksum of the attention backward's output on fixed inputs (compare across library builds: VITGAN_HIP_LIB)."""
import ctypes as C, hashlib, math, os, sys
sys.path.insert(0, os.getcwd())
import torch
import vit_gan_amd  # noqa: F401
from vit_gan_amd import _lib
L = _lib.lib()
for (B, H, S, HE) in ((64, 4, 65, 96), (16, 8, 65, 64), (8, 12, 80, 64), (8, 4, 33, 32)):
    g = torch.Generator().manual_seed(B + S)
    E = H * HE
    qkv = torch.randn(B * S, 3 * E, generator=g).to(torch.bfloat16).cuda()
    do = torch.randn(B * S, E, generator=g).to(torch.bfloat16).cuda()
    o = torch.empty(B * S, E, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B * H * S, device="cuda")
    dqkv = torch.zeros_like(qkv)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(L.vg_attention_fwd(p(qkv), p(o), p(lse), B, H, S, HE, 1 / math.sqrt(HE), st), "fwd")
    _lib.check(L.vg_attention_bwd(p(qkv), p(o), p(do), p(lse), p(dqkv), B, H, S, HE, 1 / math.sqrt(HE), st), "bwd")
    torch.cuda.synchronize()
    print(B, H, S, HE, hashlib.sha256(dqkv.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16])
