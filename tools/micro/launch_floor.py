"""Dispatch-to-dispatch floor of dependent tiny kernels: eager stream vs hipGraph replay (vg_zero_tick on 4 floats)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import vit_gan_amd  # noqa: F401
from vit_gan_amd import _lib
L = _lib.lib()
g = torch.zeros(1024, device="cuda")
N = 400
def run():
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(N):
        L.vg_zero_tick(C.c_void_p(g.data_ptr()), 4, None, st)
for name in ("eager", "graph"):
    if name == "graph":
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            run()
        fn = gr.replay
    else:
        fn = run
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / (10 * N):.2f} us per dependent tiny kernel (host enqueue {1e6 * (t1 - t0) / (10 * N):.2f} us each)")
