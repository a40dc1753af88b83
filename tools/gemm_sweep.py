import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vit_gan_amd
from vit_gan_amd import _lib
L=_lib.lib(); BF=torch.bfloat16
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
def p(t): return None if t is None else C.c_void_p(t.data_ptr())
M=int(os.environ.get("M","33280")); N=int(os.environ.get("N","1152"))
for K in (32,128,384,768,1536,3072):
    a=torch.randn(M,K,device="cuda").to(BF); w=(torch.randn(N,K,device="cuda")*0.05).to(BF); out=torch.empty(M,N,device="cuda",dtype=BF)
    f=lambda: L.vg_linear_fwd(p(a),p(w),None,None,p(out),None,None,M,N,K,0,0.0,st)
    for _ in range(3): f()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    print(f"M={M} N={N} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF  per-32k-step {us/(K/32):6.2f} us")
