#!/usr/bin/env python3
"""Every schedule / loss option of GanEngine under hipGraph replay at full size: N replays with the host synchronising after every step and
never (ordinary stream work between the replays); the two must agree bit for bit and stay finite.  (The screen that would have caught the
memset node of the penalty call: DESIGN 7, hardening.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vit_gan_amd  # noqa: F401
from vit_gan_amd.config import Config
from vit_gan_amd.engine import GanEngine
from vit_gan_amd.generator import SirenGenerator
from vit_gan_amd.modules import ViTDiscriminator

N, B = int(os.environ.get("STEPS", "200")), 256
MODES = {
    "default": {},
    "concurrent_wgrad": dict(concurrent_wgrad=True),
    "two_stream": dict(two_stream=True),
    "unfused real / fake": dict(fuse_real_fake=False),
    "hinge": dict(loss="hinge"),
    "wasserstein + clipping + diversity + instance noise": dict(loss="wasserstein", clip_d=5.0, clip_g=0.5, diversity_weight=0.1, instance_noise=0.1),
    "wasserstein + gp (C call) + instance noise": dict(loss="wasserstein", clip_d=5.0, clip_g=0.5, gp_weight=10.0, instance_noise=0.1),
    "wasserstein + gp (C call), unfused real / fake": dict(loss="wasserstein", gp_weight=10.0, fuse_real_fake=False),
    "wasserstein + gp (C call), weight gradients on a side stream": dict(loss="wasserstein", gp_weight=10.0, concurrent_wgrad=True),
    "wasserstein + gp (C call), dense top block, no dropout": dict(loss="wasserstein", gp_weight=10.0, dense_top_block=True, d_dropout=0.0, g_dropout=0.0),
    "wasserstein + gp (operator set)": dict(loss="wasserstein", gp_weight=10.0, gp_autograd=True),
    "dense top block": dict(dense_top_block=True),
    "no dropout": dict(d_dropout=0.0, g_dropout=0.0),
}
bad = []
for name, kw in MODES.items():
    out = []
    for sync_every in (False, True):
        torch.manual_seed(0)
        D = ViTDiscriminator(Config(embeddings_dimension=384, classes_count=1, batch_size=B)).cuda().train()
        G = SirenGenerator().cuda().train()
        eng = GanEngine(D, G, batch=B, use_graph=True, **kw)
        gen = torch.Generator(device="cuda").manual_seed(1)
        reals = [torch.rand(B, 3, 32, 32, device="cuda", generator=gen) * 2 - 1 for _ in range(4)]
        keep = []
        for i in range(N):
            l = eng.step(reals[i % 4])
            if i % 20 == 0:
                keep.append(l.clone())
            if sync_every or i == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        assert eng.graph_active, eng.graph_fallback_reason
        out.append((D.vit._flat.flat.detach().clone(), G._flat.flat.detach().clone(), [round(x, 4) for x in l.tolist()]))
        eng.close()
    same = torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    fin = bool(torch.isfinite(out[0][0]).all()) and bool(torch.isfinite(out[0][1]).all())
    print(f"{name:55s} bit-identical: {same}  finite: {fin}  last losses {out[0][2]}", flush=True)
    if not (same and fin):
        bad.append(name)
print("FAILED: " + ", ".join(bad) if bad else f"all {len(MODES)} modes: {N} replays independent of host synchronisation, finite")
sys.exit(1 if bad else 0)
