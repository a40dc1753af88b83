#!/usr/bin/env python3
"""Soak test: N full-size steps (C2, B=256, hipGraph replay, reference dropout) on synthetic data; losses and weights must stay finite.
GP=10 STEPS=3000 python tools/soak.py: the Wasserstein step with the gradient penalty (one C call) and the reference's clipping (training.py:78,104)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_gan_amd  # noqa: F401
from vit_gan_amd.config import Config
from vit_gan_amd.engine import GanEngine
from vit_gan_amd.generator import SirenGenerator
from vit_gan_amd.modules import ViTDiscriminator

N = int(os.environ.get("STEPS", "2000"))
torch.manual_seed(0)
D = ViTDiscriminator(Config(embeddings_dimension=384, classes_count=1, batch_size=256)).cuda().train()
G = SirenGenerator().cuda().train()
GP = float(os.environ.get("GP", "0"))
LOSS = os.environ.get("LOSS", "wasserstein" if GP else "ns")
eng = (GanEngine(D, G, batch=256, use_graph=True, loss=LOSS, gp_weight=GP, clip_d=5.0, clip_g=0.5, gp_autograd=bool(int(os.environ.get("GP_AUTOGRAD", "0"))))
       if (GP or LOSS != "ns") else GanEngine(D, G, batch=256, use_graph=True))
gen = torch.Generator(device="cuda").manual_seed(1)
reals = [torch.rand(256, 3, 32, 32, device="cuda", generator=gen) * 2 - 1 for _ in range(8)]
t0 = time.perf_counter()
hist = []
first_bad = None
for i in range(N):
    l = eng.step(reals[i % 8])
    if i % (N // 10) == 0 or i == N - 1:
        hist.append([round(x, 4) for x in l.tolist()])
    if i % 100 == 0 and first_bad is None and not bool(torch.isfinite(l).all()):
        first_bad = i
    if i % 5000 == 0 and i:  # (a progress line: a silent run of many minutes looks hung to a job runner)
        print(f"  step {i}: {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ok = all(torch.isfinite(p).all() for p in list(D.parameters()) + list(G.parameters()))
print(f"gp_weight {GP} (C call: {bool(eng.gp_c_call)}; last penalty {float(eng.gp_loss):.4f}); " if GP else "", end="")
print(f"loss {LOSS}; first non-finite loss seen at step {first_bad}; ", end="")
print(f"{N} steps in {dt:.1f} s ({N * 256 / dt:.0f} img/s incl. 10 host syncs); finite weights: {ok}")
for h in hist:
    print(h)
assert ok
