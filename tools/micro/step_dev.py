"""Deviation of the engine step from the step oracle (d_real, d_fake, g at steps 0 and 1) over losses, schedules and data seeds: what the
tolerances of tests/test_engine_gpu.py::test_engine_step_matches_oracle rest on, and why its hinge case does not use seed 0.  Run from the repo root."""
import sys, torch
import os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); import vit_gan_amd  # noqa
from test_engine_gpu import _build
from vit_gan_amd.engine import GanEngine
for loss in ("ns", "hinge"):
    for fuse in (True, False):
        for seed in (0, 1, 2):
            B = 8
            D, G, oracle = _build(B, loss)
            eng = GanEngine(D, G, batch=B, loss=loss, fuse_real_fake=fuse)
            g = torch.Generator().manual_seed(seed)
            out = []
            for it in range(2):
                real = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
                losses = eng.step(real.cuda()); torch.cuda.synchronize()
                z = eng.z.detach().cpu().clone()
                ref = oracle.step(real, z)
                got = losses.cpu().tolist()
                out.append([round(abs(got[0] - ref["d_real"]), 4), round(abs(got[1] - ref["d_fake"]), 4), round(abs(got[2] - ref["g"]), 4)])
            print(loss, fuse, seed, out, flush=True)
