#!/usr/bin/env python3
"""The data-parallel step as the driver's multi-GPU bench runs it - RCCL collectives captured in the step's hipGraph - on a ONE-rank group
(all a one-GPU box can do): N replays, with the host synchronising after every step and never; the two must agree bit for bit and stay finite.
SHARD=1: the sharded update of the mapping layer (reduce-scatter + all-gather) instead of the all-reduce."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import vit_gan_amd  # noqa: F401
from vit_gan_amd.config import Config
from vit_gan_amd.engine import GanEngine
from vit_gan_amd.generator import SirenGenerator
from vit_gan_amd.modules import ViTDiscriminator

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29631")
dist.init_process_group("nccl", rank=0, world_size=1)
N, B = int(os.environ.get("STEPS", "400")), 256
shard = bool(int(os.environ.get("SHARD", "0")))
out = []
for sync_every in (False, True):
    torch.manual_seed(0)
    D = ViTDiscriminator(Config(embeddings_dimension=384, classes_count=1, batch_size=B)).cuda().train()
    G = SirenGenerator().cuda().train()
    eng = GanEngine(D, G, batch=B, use_graph=True, exchange_single_rank=True, shard_mapping_update=shard)
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
    eng.gather_master()
    print(f"sync every step: {sync_every}; graph active: {eng.graph_active} ({eng.graph_fallback_reason}); shard: {eng.shard_map}; last losses {[round(x, 4) for x in l.tolist()]}", flush=True)
    out.append((D.vit._flat.flat.detach().clone(), G._flat.flat.detach().clone()))
    eng.close()
ok = torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and bool(torch.isfinite(out[0][0]).all()) and bool(torch.isfinite(out[0][1]).all())
print("bit-identical and finite:", ok)
dist.destroy_process_group()
assert ok
