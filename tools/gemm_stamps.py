"""Diagnostic: per-workgroup timeline of the GEMM kernel (needs `make -C vit-gan_amd/csrc dbg`)."""
import ctypes as C, os, sys
import torch
import numpy as np
root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L=C.CDLL(os.path.join(root,"vit-gan_amd","libvitgan_hip_dbg.so"))
BF=torch.bfloat16
M,N,K=33280,1152,int(os.environ.get("K","384"))
a=torch.randn(M,K,device="cuda").to(BF); w=(torch.randn(N,K,device="cuda")*0.05).to(BF); out=torch.empty(M,N,device="cuda",dtype=BF)
nwg=((M+127)//128 if os.environ.get("VG_GEMM_WM")=="2" else (M+255)//256)*(N//128)
st=torch.zeros(nwg*8,dtype=torch.int64,device="cuda")
os.environ["VG_STAMP_PTR"]=hex(st.data_ptr())
L.vg_linear_fwd.argtypes=[C.c_void_p]*7+[C.c_int]*4+[C.c_float,C.c_void_p]
s=C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    L.vg_linear_fwd(a.data_ptr(),w.data_ptr(),None,None,out.data_ptr(),None,None,M,N,K,0,0.0,s)
torch.cuda.synchronize()
t=st.cpu().numpy().reshape(nwg,8).astype(np.float64)
t0=t[:,0].min()
rel=(t[:,:8]-t0)/100.0  # us (100 MHz)
print("K",K,"dbg",os.environ.get("VG_GEMM_DBG","0"),"WM",os.environ.get("VG_GEMM_WM","4"),"nwg",nwg,"kernel span us",round(rel[:,7].max(),1), "mainloop per-step us", round(float(np.median(d[:,3]))/(K/32),3)) if False else None
d=np.diff(t[:,:8],axis=1)/100.0
print('K',K,'dbg',os.environ.get('VG_GEMM_DBG','0'),'WM',os.environ.get('VG_GEMM_WM','4'),'nwg',nwg,'span us',round(float(rel[:,7].max()),1),'mainloop/step us',round(float(np.median(d[:,3]))/(K/32),3),'epilogue us',round(float(np.median(d[:,4]+d[:,5]+d[:,6])),2))
names=["setup","issue3","first-wait","mainloop","ep-barrier","ep-prefetch","ep-body"]
for i,n in enumerate(names):
    if os.environ.get("VERBOSE"): print(f"{n:10s} mean {d[:,i].mean():7.2f} us  p50 {np.median(d[:,i]):7.2f}  max {d[:,i].max():7.2f}")
if os.environ.get("VERBOSE"): print("wg lifetime mean", (t[:,7]-t[:,0]).mean()/100.0)
order=np.argsort(t[:,0]); 
if os.environ.get("VERBOSE"): print("start times (us) of WGs by order: ", [round(float(rel[order[i],0]),2) for i in (0,100,255,256,511,512,700,1000,nwg-1) if i<nwg])
