#!/usr/bin/env python3
"""Capture the data-parallel step (RCCL collectives inside the hipGraph, one-rank group) N times in one process: every capture must succeed
while the RCCL watchdog thread keeps polling the events of the warm-up step's collectives."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import vit_gan_amd  # noqa: F401
from vit_gan_amd.config import Config
from vit_gan_amd.engine import GanEngine
from vit_gan_amd.generator import SirenGenerator
from vit_gan_amd.modules import ViTDiscriminator

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29641")
dist.init_process_group("nccl", rank=0, world_size=1)
N, B = int(os.environ.get("CAPTURES", "30")), 64
torch.manual_seed(0)
D = ViTDiscriminator(Config(embeddings_dimension=384, classes_count=1, batch_size=B)).cuda().train()
G = SirenGenerator().cuda().train()
real = torch.rand(B, 3, 32, 32, device="cuda") * 2 - 1
t0 = time.time()
for c in range(N):
    eng = GanEngine(D, G, batch=B, use_graph=True, exchange_single_rank=True, shard_mapping_update=bool(c & 1))
    print(c, "engine", flush=True)
    for k in range(3):
        eng.step(real)
        print(c, "step", k, flush=True)
    torch.cuda.synchronize()
    assert eng.graph_active, (c, eng.graph_fallback_reason)
    eng.close()
    del eng
    print(c, "closed", flush=True)
print(f"{N} captures with collectives, all replayed; {time.time() - t0:.1f} s")
dist.destroy_process_group()
