"""N > 1 path on CPU: two gloo ranks.  (a) GradSync range all-reduce; (b) sharding the batch over ranks and
averaging the summed gradients reproduces the single-process gradient of the global batch (the data-parallel
contract of SURVEY 8e), checked with the fp32 oracle as the per-rank compute."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import vit_gan_amd  # noqa: F401
        from vit_gan_amd.dist import GradSync, shard_batch
        from oracle import step_oracle as so, vit_oracle as vo

        torch.set_num_threads(2)
        sync = GradSync(None, torch.device("cpu"))
        assert sync.world == world and not sync.overlap
        # (a) ranges
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        sync.reduce_range(flat, 100, 900)
        sync.wait()
        ref = torch.arange(1000, dtype=torch.float32)
        tot = sum(r + 1 for r in range(world))
        assert torch.equal(flat[100:900], ref[100:900] * tot) and torch.equal(flat[:100], ref[:100] * (rank + 1))
        # (a') bf16-compressed exchange: the sum of the bf16-rounded contributions, written back to the fp32 buffer
        flat = (torch.arange(1000, dtype=torch.float32) * 0.37 + 0.011) * (rank + 1)
        want = sum(((torch.arange(1000, dtype=torch.float32) * 0.37 + 0.011) * (r + 1)).to(torch.bfloat16) for r in range(world))
        sync.reduce_range(flat, 200, 800, compress=True)
        sync.wait()
        untouched = (torch.arange(1000, dtype=torch.float32) * 0.37 + 0.011) * (rank + 1)
        assert torch.equal(flat[200:800], want[200:800].float()), "bf16 exchange"
        assert torch.equal(flat[:200], untouched[:200]) and torch.equal(flat[800:], untouched[800:])
        # (a'') the sharded update's exchange (GanEngine(shard_mapping_update=True)): after the reduce-scatter this rank's share holds
        #       the SUM over the ranks; after "updating" only that share, the all-gather gives every rank every share
        base = torch.arange(1000, dtype=torch.float32) * 0.5 + 1.0
        flat = base * (rank + 1)
        a, b = sync.share(200, 800)
        assert (b - a) * world == 600 and a == 200 + rank * (600 // world)
        sync.reduce_scatter_range(flat, 200, 800)
        sync.wait()
        assert torch.equal(flat[a:b], base[a:b] * tot), "this rank's share must hold the sum"
        assert torch.equal(flat[:200], base[:200] * (rank + 1)) and torch.equal(flat[800:], base[800:] * (rank + 1))
        upd = torch.zeros(1000)
        upd[a:b] = flat[a:b] * 2 + rank          # each rank updates its own share only
        sync.all_gather_range(upd, 200, 800)
        sync.wait()
        for r in range(world):
            per = 600 // world
            lo_, hi_ = 200 + r * per, 200 + (r + 1) * per
            assert torch.equal(upd[lo_:hi_], base[lo_:hi_] * tot * 2 + r), "all-gather of the updated shares"
        assert float(upd[:200].abs().max()) == 0.0 and float(upd[800:].abs().max()) == 0.0
        # (b) sharded D gradient == global-batch gradient
        d = vo.VitDims(embed=128, heads=4, layers=1, classes=1)
        st = {k: v.clone().requires_grad_(True) for k, v in vo.init_vit_state(d, 7).items()}
        GB = 8
        g = torch.Generator().manual_seed(3)
        real = torch.rand(GB, 3, 32, 32, generator=g) * 2 - 1
        lo, hi = shard_batch(GB, rank, world)
        so.d_loss_real(vo.vit_forward(st, real[lo:hi], d), "ns").backward()
        names = list(st)
        flat_g = torch.cat([st[k].grad.reshape(-1) for k in names])
        sync.reduce_range(flat_g, 0, flat_g.numel())
        sync.wait()
        flat_g /= world  # the 1/world factor the fused AdamW applies as `gscale`
        if rank == 0:
            st2 = {k: v.detach().clone().requires_grad_(True) for k, v in st.items()}
            so.d_loss_real(vo.vit_forward(st2, real, d), "ns").backward()
            ref_g = torch.cat([st2[k].grad.reshape(-1) for k in names])
            err = float((flat_g - ref_g).abs().max()) / float(ref_g.abs().max())
            out.put(("ok", err))
    except Exception as e:  # surface the failure in the parent
        out.put(("err", f"rank {rank}: {type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_gradient_exchange():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    status, val = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert status == "ok", val
    assert val < 1e-5, f"sharded gradient differs from the global-batch gradient: rel err {val}"


def test_shard_batch():
    from vit_gan_amd.dist import shard_batch
    assert [shard_batch(2048, r, 8) for r in (0, 7)] == [(0, 256), (1792, 2048)]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


def test_backward_pieces_tile_the_gradient_buffer_exactly_once():
    """Host logic of the overlapped D exchange: stage ranges are contiguous and increasing, gradient ranges tile
    [0, total) from the top down, and a range is only released once every block in it has run."""
    from vit_gan_amd.dist import backward_pieces
    layer0, stride = 1000, 77
    for L in (1, 2, 3, 6, 12):
        total = layer0 + L * stride + 555  # embedding | L blocks | head
        for chunks in (1, 2, 3, 4, 50):
            plan = backward_pieces(L, chunks, layer0, stride, total)
            assert plan[0][0] == 0 and plan[-1][1] == L + 2 and plan[0][3] == total and plan[-1][2] == 0
            for (a0, a1, lo, hi), nxt in zip(plan, plan[1:] + [None]):
                assert a0 < a1 and lo < hi
                if nxt is not None:
                    assert nxt[0] == a1 and nxt[3] == lo
                    blocks_done = a1 - 1                      # stages 1..a1-1 = blocks L-1 .. L-blocks_done
                    assert lo == layer0 + (L - blocks_done) * stride
            assert len(plan) == max(1, min(chunks, L))
