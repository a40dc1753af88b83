#!/usr/bin/env python3
"""Is the penalty step deterministic run to run?  N steps (C2, B=256, graph replay), losses every 100 steps and a checksum of D's weights.
SYNC_EVERY changes how far the host may run ahead of the device (a host/device race would show as a dependence on it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vit_gan_amd  # noqa: F401
from vit_gan_amd.config import Config
from vit_gan_amd.engine import GanEngine
from vit_gan_amd.generator import SirenGenerator
from vit_gan_amd.modules import ViTDiscriminator

N = int(os.environ.get("STEPS", "600"))
SYNC = int(os.environ.get("SYNC_EVERY", "100"))
GRAPH = bool(int(os.environ.get("GRAPH", "1")))
torch.manual_seed(0)
D = ViTDiscriminator(Config(embeddings_dimension=384, classes_count=1, batch_size=256)).cuda().train()
G = SirenGenerator().cuda().train()
GP = float(os.environ.get("GP", "10"))
eng = GanEngine(D, G, batch=256, use_graph=GRAPH, loss="wasserstein", gp_weight=GP, clip_d=5.0, clip_g=0.5,
                gp_autograd=bool(int(os.environ.get("GP_AUTOGRAD", "0"))))
gen = torch.Generator(device="cuda").manual_seed(1)
if int(os.environ.get("FIXED_EPS", "0")):
    eng.gp_epsilon = torch.rand(256, 1, 1, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
reals = [torch.rand(256, 3, 32, 32, device="cuda", generator=gen) * 2 - 1 for _ in range(8)]
keep = []
for i in range(N):
    l = eng.step(reals[i % 8])
    if i % int(os.environ.get("PRINT_EVERY", "100")) == 0:
        keep.append((i, l.clone(), eng.gp_loss.clone()))
    if i % SYNC == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
for i, l, g in keep:
    print(i, [round(x, 5) for x in l.tolist()], round(float(g), 5))
print("checksum", float(D.vit._flat.flat.double().abs().sum()), float(G._flat.flat.double().abs().sum()))
