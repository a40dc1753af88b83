#!/usr/bin/env python3
"""Where does the first non-finite value of a long penalty run come from?  Replays the step to START, then checks every step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vit_gan_amd  # noqa: F401
from vit_gan_amd.config import Config
from vit_gan_amd.engine import GanEngine
from vit_gan_amd.generator import SirenGenerator
from vit_gan_amd.modules import ViTDiscriminator

START, N = int(os.environ.get("START", "2550")), int(os.environ.get("STEPS", "2750"))
torch.manual_seed(0)
D = ViTDiscriminator(Config(embeddings_dimension=384, classes_count=1, batch_size=256)).cuda().train()
G = SirenGenerator().cuda().train()
eng = GanEngine(D, G, batch=256, use_graph=True, loss="wasserstein", gp_weight=float(os.environ.get("GP", "10")), clip_d=5.0, clip_g=0.5,
                gp_autograd=bool(int(os.environ.get("GP_AUTOGRAD", "0"))))
gen = torch.Generator(device="cuda").manual_seed(1)
reals = [torch.rand(256, 3, 32, 32, device="cuda", generator=gen) * 2 - 1 for _ in range(8)]
fd, fg = D.vit._flat, G._flat
for i in range(N):
    l = eng.step(reals[i % 8])
    if i % 100 == 0:
        torch.cuda.synchronize()
    if i >= START or i % 250 == 0:
        torch.cuda.synchronize()
        st = dict(step=i, losses=[round(x, 3) for x in l.tolist()], gp=round(float(eng.gp_loss), 4),
                  dw=float(fd.flat.abs().max()), dg=float(fd.grad.abs().max()), gw=float(fg.flat.abs().max()), gg=float(fg.grad.abs().max()),
                  fake=float(eng.imgs[256:].float().abs().max()), dfake=float(eng.dfake.float().abs().max()),
                  clipn=[round(float(eng.clip_scratch[0, 0]), 3), round(float(eng.clip_scratch[1, 0]), 3)])
        print(st, flush=True)
        if not all(v == v and abs(v) < 1e30 for v in (st["dw"], st["dg"], st["gw"], st["gg"], st["gp"])):
            break
