"""Print the loss differences between the two-stream and the unfused single-stream schedules (run from a tree's root)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_engine_gpu as t
B, n = 8, int(os.environ.get("N", "3"))
runs = {}
for name, kw in (("unfused", dict(fuse_real_fake=False, use_graph=False, concurrent_wgrad=False)), ("two_stream", dict(two_stream=True, use_graph=False))):
    eng, D, G, _ = t._bench_like(B, d_dropout=0.0, g_dropout=0.0, **kw)
    runs[name] = t._run_steps(eng, n, B)
(l_u, s_u), (l_t, s_t) = runs["unfused"], runs["two_stream"]
print((l_t - l_u).abs())
print("state", float((s_t[0] - s_u[0]).abs().max()), float(((s_t[0] - s_u[0]).abs() < 2e-5).float().mean()))
gr = {}
for name, kw in (("unfused", dict(fuse_real_fake=False, use_graph=False, concurrent_wgrad=False)), ("two_stream", dict(two_stream=True, use_graph=False))):
    eng, D, G, _ = t._bench_like(B, d_dropout=0.0, g_dropout=0.0, **kw)
    t._run_steps(eng, 1, B)
    gr[name] = (eng.vit._flat.grad.clone().cpu(), eng.gen._flat.grad.clone().cpu())
for i, nm in enumerate(("D", "G")):
    a, b = gr["unfused"][i], gr["two_stream"][i]
    print(nm, "grad max", float(a.abs().max()), "max diff", float((a - b).abs().max()), "frac equal", float((a == b).float().mean()))
