#!/usr/bin/env python3
"""images/sec of the full ViTGAN G+D step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config C2 of BASELINE.json / SURVEY 8): CIFAR-shaped 3x32x32, patch 4 (64+1 tokens), E=384,
4 heads, 6 blocks discriminator (src/v2) + SLN/SIREN generator (src/v1), per-GPU batch 256, bf16 MFMA
compute with fp32 accumulation and fp32 master weights, one full alternating step per "step":
3 D forward + 3 D backward (the third without weight gradients) + G forward + G backward + 2 AdamW.
Weak scaling: the per-GPU batch is fixed, gradients are all-reduced over RCCL.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(seconds_budget=30.0):
    """The reference's algorithm (oracle = fp32 CPU restatement pinned to the reference's outputs) timed on this box's
    host cores: config C1 (B=64, fp32), same step structure; 2 warm-up steps, then 5 timed steps unless the
    time budget runs out first (never fewer than 2)."""
    import torch
    from oracle import gen_oracle as go, step_oracle as so, vit_oracle as vo

    B = 64
    dd, gd = vo.VitDims(classes=1), go.GenDims()
    oracle = so.GanStepOracle(vo.init_vit_state(dd, 0), go.init_gen_state(gd, 1), dd, gd)
    g = torch.Generator().manual_seed(1234)
    real = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
    z = torch.randn(B, gd.latent, generator=g)
    for _ in range(2):
        oracle.step(real, z)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.step(real, z)
        n += 1
        el = time.perf_counter() - t0
        if n >= 5 or (n >= 2 and el > seconds_budget):
            break
    return {"value": round(n * B / el, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full G+D steps after 2 warm-ups at config C1 (B=64, fp32, torch {torch.__version__} CPU, "
                      f"{_cpu_model()}, os.cpu_count={os.cpu_count()})"}


def vit_flops_per_image(image, patch, embed, layers=6, mlp_ratio=2, classes=1, channels=3):
    """Algorithmic matmul FLOPs of one discriminator forward per image (F_D = 961.72 MFLOP at C2)."""
    N = (image // patch) ** 2
    S, E, r = N + 1, embed, mlp_ratio
    per_layer = 2 * S * E * 3 * E + 2 * S * S * E + 2 * S * S * E + 2 * S * E * E + 2 * (2 * S * E * r * E)
    return 2 * N * (channels * patch * patch) * E + layers * per_layer + 2 * E * E + 2 * E * classes


def vit_dead_flops_per_image(image, patch, embed, mlp_ratio=2, fp8_attention=False):
    """Matmul FLOPs of one discriminator forward per image that the reference spends on values nobody reads: the classifier takes the
    CLS row of the top block only (src/v2/modules.py:195), so that block's out-projection, fc1, fc2 and - with dot-product attention -
    every query but the CLS one matter for 1 of its S rows.  The engine does not compute the other S - 1 (csrc/engine.hip, "pruned tail";
    the same rows' gradients are exactly zero in the backward), which is 10.5 % of F_D at C2."""
    N = (image // patch) ** 2
    S, E, r = N + 1, embed, mlp_ratio
    row_local = 2 * S * E * E + 2 * (2 * S * E * r * E)
    attn = 0 if fp8_attention else 4 * S * S * E
    return (row_local + attn) * (S - 1) // S


def gen_flops_per_image(latent, tokens, embed, layers, siren_hidden, out_features):
    """Algorithmic matmul FLOPs of one generator forward per image (F_G1 = 243.79 MFLOP at the v1 defaults)."""
    Z, T, E, O = latent, tokens, embed, siren_hidden
    per_layer = 2 * T * E * 3 * E + 4 * T * T * E + 2 * T * E * E + 2 * T * E * E
    return 2 * Z * T * E + layers * per_layer + 2 * T * E * O + 2 * T * O * out_features


PEAK_HBM_GBS = 8000.0      # HBM3E, MI355X_MICROARCH.md "Chip-level parameters" (6.3 TB/s is what a streaming copy reaches)
TRAFFIC_FILE = "profiles/r04_gemm_pmc_traffic.json"      # tools/pmc_traffic.py over tools/gemm_bench.py (separate --pmc passes)
STEP_PMC_FILE = "profiles/r04_step_pmc_summary.json"     # tools/pmc_summary.py over the step (separate --pmc passes)
E = 384


def _per_dispatch_us(torch, fn, reps):
    """PER-DISPATCH time: every launch sits between its own pair of HIP events on the stream the kernel runs on (an event
    completes only when the launch before it has finished, so consecutive launches do not overlap inside a pair: this is what
    rocprofv3's per-dispatch AverageNs measures).  Returns (mean, median, min) in microseconds."""
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return sum(t) / len(t), t[len(t) // 2], t[0]


def step_roofline(torch, B, reps=30):
    """Roofline of the step's heaviest kernels, timed live at their heaviest shapes (the fused real+fake pass: M = 2B*65 rows).

    Dominant kernel (rocprof: profiles/r03_bench_b256_kernel_stats.csv) = vg_gemm_row_kernel<1>, the full-row input-gradient
    GEMM with the LayerNorm backward in its epilogue (csrc/gemm_row.hip), at the QKV shape: dx = gres + LN'(dqkv[M,1152] Wqkv),
    dxm = dx * mask.  Algorithmic work per launch: 2*M*384*1152 flops; bytes = dqkv + packed W + x + gres + dx + dxm (bf16) +
    mean, rstd (fp32) = 2*(M*1152 + 384*1152 + 4*M*384) + 8*M.  Intensity 163 flop/B is under the chip's ridge (2500 TFLOP/s /
    8 TB/s = 312): by the roofline model it is HBM-bound, `frac` is taken against the 8 TB/s HBM peak and the fraction of the
    dense bf16 MFMA peak is reported beside it.
    `traffic` = HBM bytes per launch of this kernel from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gfx950
    FETCH_SIZE x2 correction) over tools/gemm_bench.py, read from the committed summary named in `traffic_source` - PMC
    counters cannot be read from inside the process.
    `kernels`: the same figures for the five heaviest instantiations of the step (shape, per-dispatch time, TFLOP/s, GB/s, both
    fractions), each through the C ABI entry point the engine uses for it."""
    import ctypes as C
    from vit_gan_amd import _lib
    L = _lib.lib()
    BF = torch.bfloat16
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    M = 2 * B * 65

    def p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def rnd(*shape, s=1.0):
        return (torch.randn(*shape, device="cuda") * s).to(BF)

    def packed(w, K, tr):
        wp = torch.empty(E * K, device="cuda", dtype=BF)
        _lib.check(L.vg_row_pack_weight(p(w), w.shape[1], K, tr, p(wp), st), "vg_row_pack_weight")
        return wp

    x384, x768, x1152, res = rnd(M, 384), rnd(M, 768), rnd(M, 1152), rnd(M, 384)
    gam, bet, bias = torch.ones(E, device="cuda"), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    mean, rstd = torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")
    o384, o384b, o768, o1152 = (torch.empty(M, n, device="cuda", dtype=BF) for n in (384, 384, 768, 1152))
    wqkv, w1, w2 = rnd(1152, 384, s=0.05), rnd(768, 384, s=0.05), rnd(384, 768, s=0.05)
    wqkv_t, w2_p = packed(wqkv, 1152, 1), packed(w2, 768, 0)
    b768 = torch.zeros(768, device="cuda")
    code = torch.empty(M, 768, device="cuda", dtype=torch.uint8)
    part = torch.empty(L.vg_row_parts(M), 3 * E, device="cuda")
    lse = torch.zeros(2 * B * 4 * 65, device="cuda")
    dw = torch.zeros(1152, 384, device="cuda")
    WG_SPLITS = 28  # 9 tiles of 128 x 384 x 28 K slices = 252 workgroups, as the engine's grouped launch fills the chip (12 slices x 21 tiles)
    slab = torch.empty(WG_SPLITS * 1152 * 384, device="cuda")
    S, H, HE = 65, 4, 96

    def chk(rc, what):
        _lib.check(rc, what)

    specs = [
        ("vg_gemm_row_kernel<1>: QKV input gradient + LayerNorm backward (dx, dxm)", [M, 384, 1152],
         lambda: chk(L.vg_linear_dgrad_ln_bwd(p(x1152), p(wqkv_t), p(x384), p(mean), p(rstd), p(gam), p(res), p(o384), p(o384b), p(part), M, 1152,
                                              0.1, 1, 0, None, st), "vg_linear_dgrad_ln_bwd"),
         2.0 * M * 384 * 1152, 2 * (M * 1152 + 384 * 1152 + 4 * M * 384) + 8 * M, "row qkv dgrad+ln bwd"),
        ("vg_gemm_row_kernel<0>: fc2 + dropout + residual + next LayerNorm", [M, 384, 768],
         lambda: chk(L.vg_linear_ln_fwd(p(x768), p(w2_p), p(bias), p(res), p(o384), p(o384b), p(mean), p(rstd), p(gam), p(bet), M, 768, 1e-5,
                                        0.1, 1, 2, None, st), "vg_linear_ln_fwd"),
         2.0 * M * 384 * 768, 2 * (M * 768 + 384 * 768 + 3 * M * 384) + 8 * M, "row fc2+res+ln fwd"),
        ("vg_gemm_tn384_kernel<6>: QKV weight gradient (one of the four problems of a block's grouped launch; + its slab fold)", [1152, 384, M],
         lambda: chk(L.vg_linear_wgrad(p(x1152), p(x384), p(dw), p(slab), slab.numel(), M, 1152, 384, WG_SPLITS, 1, st), "vg_linear_wgrad"),
         2.0 * M * 1152 * 384, 2 * (M * 1152 + M * 384) + 4 * WG_SPLITS * 1152 * 384, "tn qkv wgrad"),
        ("vg_attn_bwd2_kernel<96,5>: fused attention backward, one workgroup per (image, head), two LDS images", [2 * B, H, S, HE],
         lambda: chk(L.vg_attention_bwd(p(x1152), p(x384), p(res), p(lse), p(o1152), 2 * B, H, S, HE, 1.0 / HE ** 0.5, st), "vg_attention_bwd"),
         10.0 * 2 * B * H * S * S * HE, 2 * (M * 1152 * 2 + M * 384 * 2) + 4 * 2 * B * H * S, None),
        ("vg_gemm_wr_kernel<0,1,2>: fc1 + GELU, gelu' for the backward as one byte per element", [M, 768, 384],
         lambda: chk(L.vg_linear_gelu_fwd(p(x384), p(w1), p(b768), p(o768), p(code), M, 768, 384, st), "vg_linear_gelu_fwd"),
         2.0 * M * 768 * 384, 2 * (M * 384 + 768 * 384 + M * 768) + M * 768, "wr fc1+gelu+gelu' bytes"),
    ]
    # Figures quoted from committed PMC summaries are tied to the tree they were measured on: the summary carries the hash of the
    # kernel / engine sources (tools/tree_hash.py) and, where known, the commit; when this tree hashes differently the quoted
    # fields are null and `*_stale` says why (nothing used to tie the file to the tree being benchmarked).
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from tree_hash import tree_hash
    here = tree_hash(ROOT)
    traffic, traffic_meta = {}, {}
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            traffic_meta = json.load(f)
        traffic = traffic_meta["kernels"] if traffic_meta.get("tree_hash") == here else {}
    except (OSError, KeyError, ValueError):
        pass
    kernels = []
    for name, shape, fn, flops, byts, tkey in specs:
        mean_us, med_us, min_us = _per_dispatch_us(torch, fn, reps)
        tf, gbs = flops / mean_us / 1e6, byts / mean_us / 1e3
        rec = traffic.get(tkey) if tkey else None
        kernels.append({"kernel": name, "shape": shape, "avg_launch_us": round(mean_us, 2), "median_launch_us": round(med_us, 2),
                        "min_launch_us": round(min_us, 2), "tflops": round(tf, 1), "gbs": round(gbs, 1),
                        "frac_mfma": round(tf / PEAK_BF16_TFLOPS, 4), "frac_hbm": round(gbs / PEAK_HBM_GBS, 4),
                        "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": byts,
                        "traffic": rec["hbm_bytes_per_launch"] if rec and rec.get("rows_M") == M else None})
    d = kernels[0]
    out = {"bound": "hbm", "achieved": d["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": d["frac_hbm"],
           "traffic": d["traffic"],
           "traffic_source": (TRAFFIC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/gemm_bench.py)") if d["traffic"] else None,
           "mfma": {"achieved": d["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": d["frac_mfma"]},
           "intensity_flop_per_byte": round(d["algorithmic_flops_per_launch"] / d["algorithmic_bytes_per_launch"], 1),
           "ridge_flop_per_byte": round(PEAK_BF16_TFLOPS * 1e3 / PEAK_HBM_GBS, 1),
           "kernel": d["kernel"], "shape_M_N_K": d["shape"],
           "algorithmic_flops_per_launch": d["algorithmic_flops_per_launch"], "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
           "avg_launch_us": d["avg_launch_us"], "median_launch_us": d["median_launch_us"], "min_launch_us": d["min_launch_us"],
           "timing": f"per-dispatch HIP event pairs on the kernel's stream, {reps} launches", "kernels": kernels,
           "tree_hash": here}
    if traffic_meta and not traffic:
        out["traffic_stale"] = f"{TRAFFIC_FILE} was measured on tree {traffic_meta.get('tree_hash')} (commit {traffic_meta.get('commit')}), not on this one"
    out["hbm_GB_per_step"] = out["mfma_util_percent_step"] = None
    try:
        with open(os.path.join(ROOT, STEP_PMC_FILE)) as f:
            sp = json.load(f)
        out["step_pmc_source"] = STEP_PMC_FILE + " (rocprofv3 --pmc passes over bench.py, tools/pmc_summary.py; B = 256)"
        out["step_pmc_tree_hash"], out["step_pmc_commit"] = sp.get("tree_hash"), sp.get("commit")
        if sp.get("tree_hash") == here:
            out["hbm_GB_per_step"] = sp["hbm"]["total_GB_per_step"]
            out["mfma_util_percent_step"] = sp["mfma_util_percent"]["<whole step, all vg_ kernels>"]["util"]
        else:
            out["step_pmc_stale"] = "measured on another tree than the one being benchmarked: the two quoted fields are null"
    except (OSError, KeyError, ValueError):
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 256 for C2, 128 for C4 / C5)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: FIXED global batch split over the ranks (config C3 of BASELINE.json: --global-batch 2048 -> "
                         "1024 / 512 / 256 per GPU at 2 / 4 / 8 GPUs); default 0 = weak scaling, the per-GPU batch is fixed")
    ap.add_argument("--workload", default="c2", choices=["c2", "c4", "c5"],
                    help="c2 (default, the metric's configuration): 32x32 patch 4 E=384 4 heads, v1 row-token generator; "
                         "c4: 64x64 patch 8 E=512 8 heads; c5: 128x128 patch 16 E=768 12 heads (bf16 attention) - "
                         "both with the patch-grid generator (SURVEY 8f f1); extra measurements, not the headline")
    ap.add_argument("--loss", default="ns", choices=["ns", "hinge", "wasserstein"])
    ap.add_argument("--gp", type=float, default=0.0,
                    help="weight of the WGAN-GP gradient penalty in the discriminator step (src/v2/training.py:101-106; 10 with --loss wasserstein is "
                         "the reference's unreached recipe): an extra measurement, not the headline - the penalty runs through torch autograd over "
                         "the twice-differentiable operators; the step with it is captured and replayed as a hipGraph like the plain one")
    ap.add_argument("--graph", type=int, default=-1,
                    help="1: replay the step as a hipGraph (on > 1 GPU the capture includes the RCCL all-reduces); 0: eager; -1 (default): "
                         "on - the engine falls back to eager, loudly, when the process group cannot be captured (gloo)")
    ap.add_argument("--compress-mapping-grad", type=int, default=0,
                    help="data parallel only, 1: exchange the generator's 12.6 M-parameter mapping gradient as bf16 (halves the one exchange "
                         "that cannot hide behind compute, but the sum is then formed in bf16); default 0: the exact fp32 all-reduce, the "
                         "reference's arithmetic - what a multi-GPU headline must be measured on (VERDICT r3 weak 7)")
    ap.add_argument("--shard-mapping-update", type=int, default=0,
                    help="data parallel only, 1: reduce-scatter the mapping gradient in fp32, AdamW on each rank's share, all-gather the updated bf16 "
                         "shadow (exact sums, 3/4 of the all-reduce's bytes, 1/world of AdamW's traffic on that layer); default 0: all-reduce")
    ap.add_argument("--no-fuse", action="store_true", help="run D(real) and D(fake) as two passes like the reference")
    ap.add_argument("--dropout", type=int, default=1, help="1: reference train-mode dropout (D 0.1 at 13 sites, G 0.2 at 8 sites), 0: none")
    ap.add_argument("--fp8-attention", type=int, default=-1,
                    help="1: fp8 (e4m3) MFMA operands for the attention's Q.K^T and P.V; -1 (default): on for --workload c5 (BASELINE.json "
                         "configs[4] names fp8 MFMA attention), off otherwise")
    ap.add_argument("--two-stream", type=int, default=0, help="1: run the step as two concurrent chains on two HIP streams (single GPU)")
    ap.add_argument("--dense-top-block", type=int, default=0,
                    help="1: compute every row of the top encoder block (the reference's operator graph row for row); default 0: only the CLS rows "
                         "the classifier reads behind that block's attention - same logits and gradients (DESIGN.md s3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the C4 / C5 measurements that follow the headline's timed region on one GPU (JSON field extra_workloads)")
    ap.add_argument("--no-roofline", action="store_true",
                    help="profiling aid: skip the dominant-kernel timing leg, so a rocprofv3 run of this command contains the step's launches only")
    ap.add_argument("--single-stream", action="store_true", help="(default since round 2; kept for the profiling scripts) weight gradients on the main stream")
    ap.add_argument("--concurrent-wgrad", type=int, default=0,
                    help="1: the discriminator's weight gradients on a side stream beside its input gradients (the default until the "
                         "persistent weights-in-registers GEMMs: their workgroups hold the CUs for a whole launch, and the side stream "
                         "measured 6.70 vs 6.67 ms/step)")
    ap.add_argument("--roofline-only", action="store_true", help="profiling aid: run only the dominant-kernel timing leg and print its object")
    ap.add_argument("--rehearse-exchange", type=int, default=0,
                    help="1 (with --gpus 1): run the multi-GPU code path of this file on ONE rank - a one-rank RCCL group, the gradient exchange calls "
                         "executed (identity collectives) and captured in the step's hipGraph, the barriers and the MAX all-reduce of the timing; "
                         "what a one-GPU box can rehearse of the driver's N > 1 run")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the default and the measured path); gloo only to rehearse the multi-rank code path on a box "
                         "with fewer GPUs than ranks (ranks then share devices: local_rank %% device_count)")
    args = ap.parse_args()
    if args.roofline_only:
        import torch
        import vit_gan_amd  # noqa: F401
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU path")
        print(json.dumps({"roofline": step_roofline(torch, args.batch or 256)}))
        return

    import torch
    import torch.distributed as dist
    import vit_gan_amd  # noqa: F401
    from vit_gan_amd.config import Config
    from vit_gan_amd.engine import GanEngine
    from vit_gan_amd.generator import SirenGenerator
    from vit_gan_amd.modules import ViTDiscriminator

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU path")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dp = world > 1 or (bool(args.rehearse_exchange) and world == 1)  # a process group exists: barriers, exchanged gradients, MAX over ranks
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local)

    GEO = {"c2": dict(image=32, patch=4, embed=384, heads=4, batch=256, gpatch=0),
           "c4": dict(image=64, patch=8, embed=512, heads=8, batch=128, gpatch=8),
           "c5": dict(image=128, patch=16, embed=768, heads=12, batch=128, gpatch=16)}
    use_graph = True if args.graph < 0 else bool(args.graph)

    def make(workload, B, fp8_flag, loss=None, gp=None):
        """Discriminator, generator and engine of one workload (random init of that architecture, synthetic data)."""
        geo = GEO[workload]
        torch.manual_seed(0)  # identical init on every rank (v1 config.py:61 seed 0)
        cfg = Config(embeddings_dimension=geo["embed"], attention_heads_count=geo["heads"], transformer_blocks_count=6, mlp_ratio=2,
                     patch_size=geo["patch"], image_size=geo["image"], input_channels=3, classes_count=1,
                     dropout_rate=0.1 if args.dropout else 0.0, batch_size=B)
        D = ViTDiscriminator(cfg).to(dev).train()                       # Config default dropout_rate = 0.1 (src/v2/utils.py:30)
        fp8 = (workload == "c5") if fp8_flag < 0 else bool(fp8_flag)
        D.vit.attention_fp8 = fp8
        if workload == "c2":
            G = SirenGenerator(dropout=0.2 if args.dropout else 0.0).to(dev).train()  # src/v1/config.py:36,39
        else:
            G = SirenGenerator(image_size=geo["image"], embed=geo["embed"], heads=geo["heads"], patch_size=geo["gpatch"],
                               dropout=0.2 if args.dropout else 0.0).to(dev).train()
        eng = GanEngine(D, G, batch=B, loss=args.loss if loss is None else loss, fuse_real_fake=not args.no_fuse, use_graph=use_graph, seed=1000 + rank,
                        concurrent_wgrad=bool(args.concurrent_wgrad) and not args.single_stream, two_stream=bool(args.two_stream) and world == 1,
                        compress_mapping_grad=bool(args.compress_mapping_grad) and dp, shard_mapping_update=bool(args.shard_mapping_update) and dp,
                        gp_weight=args.gp if gp is None else gp, exchange_single_rank=bool(args.rehearse_exchange) and world == 1,
                        dense_top_block=bool(args.dense_top_block))
        return geo, G, eng, fp8

    def step_flops(geo, G, fp8):
        f_d = vit_flops_per_image(geo["image"], geo["patch"], geo["embed"])
        gd = G._dims
        f_g = gen_flops_per_image(gd.Z, gd.T, gd.E, gd.L, gd.O, gd.CW)
        f_step = 8 * f_d + 3 * f_g  # algorithmic FLOPs per real image (the reference's operator graph)
        f_exec = f_step - (0 if args.dense_top_block else 8 * vit_dead_flops_per_image(geo["image"], geo["patch"], geo["embed"], fp8_attention=fp8))
        return f_step, f_exec  # (the second: what the engine really multiplies)

    B = args.batch or GEO[args.workload]["batch"]
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not divisible by {world} ranks")
        B = args.global_batch // world
    geo, G, eng, fp8_attn = make(args.workload, B, args.fp8_attention)
    IMG = geo["image"]
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    reals = [torch.rand(B, 3, IMG, IMG, device=dev, generator=gen) * 2 - 1 for _ in range(4)]  # resident synthetic batches
    torch.manual_seed(4321 + rank)  # noise stream

    for i in range(args.warmup):
        eng.step(reals[i % 4])
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    torch.cuda.synchronize()
    # EXACTLY args.steps steps between the barriers; events at the boundaries of (up to) 6 equal windows inside that region
    # give the spread of the headline (device time per window, this rank)
    nwin = max(1, min(6, args.steps // 5)) if args.steps >= 10 else 1
    bounds = [round(k * args.steps / nwin) for k in range(nwin + 1)]
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(nwin + 1)]
    e0, e1 = marks[0], marks[-1]
    t0 = time.perf_counter()
    marks[0].record()
    nb = 1
    for i in range(args.steps):
        losses = eng.step(reals[i % 4])
        if i + 1 == bounds[nb]:
            marks[nb].record()
            nb += 1
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_s = e0.elapsed_time(e1) * 1e-3
    t = torch.tensor([max(wall, dev_s)], device=dev, dtype=torch.float64)
    if dp:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    win = sorted(marks[k].elapsed_time(marks[k + 1]) / (bounds[k + 1] - bounds[k]) for k in range(nwin))
    lv = losses.cpu().tolist()
    ok = all(x == x and abs(x) < 1e4 for x in lv)

    if rank == 0:
        f_step, f_exec = step_flops(geo, G, fp8_attn)
        ips = args.steps * B * world / elapsed
        step_tf = ips * f_step / 1e12 / world
        # the roofline leg always times the C2 shape
        roof = {"skipped": "--no-roofline"} if args.no_roofline else step_roofline(torch, 256 if args.workload != "c2" else B)
        roof["step_tflops_per_gpu"] = round(step_tf, 1)
        roof["step_frac_of_peak"] = round(step_tf / PEAK_BF16_TFLOPS, 4)
        # BASELINE.md's definition (images/s x F_step of the reference's graph / peak) above; the MFMA pipe itself did less:
        roof["step_tflops_executed_per_gpu"] = round(ips * f_exec / 1e12 / world, 1)
        roof["step_frac_of_peak_executed"] = round(ips * f_exec / 1e12 / world / PEAK_BF16_TFLOPS, 4)
        out = {
            "metric": "images/sec (G+D step) ViTGAN 32x32 patch4 dim384" if args.workload == "c2" else f"images/sec (G+D step) ViTGAN {args.workload} shape", "value": round(ips, 1), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "windows_ms_per_step": {"n": nwin, "steps_each": args.steps // nwin, "median": round(win[nwin // 2], 4), "min": round(win[0], 4),
                                    "max": round(win[-1], 4), "note": "device time of equal windows inside the timed region (rank 0)"},
            "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": {"c2": "C2: CIFAR-10-shaped 3x32x32, patch 4 (65 tokens), E=384, 4 heads, 6 blocks ViT discriminator + "
                                          "SLN/SIREN generator (z=1024, 32 tokens, 4 blocks), full alternating G+D step, AdamW",
                                    "c4": "C4 shape: 3x64x64, patch 8 (65 tokens), E=512, 8 heads, 6 blocks ViT discriminator + patch-grid "
                                          "SLN/SIREN generator (64 tokens), full alternating G+D step, AdamW",
                                    "c5": "C5 shape: 3x128x128, patch 16 (65 tokens), E=768, 12 heads, 6 blocks ViT discriminator + patch-grid "
                                          "SLN/SIREN generator (64 tokens), full alternating G+D step, AdamW"}[args.workload],
                       "fp8_attention": fp8_attn,
                       "per_gpu_batch": B, "global_batch": B * world, "loss": args.loss, "gp_weight": args.gp, "dropout": {"D": eng.p_d, "G": eng.p_g},
                       "parallelism": f"dp{world}", "backend": args.backend if dp else None, "rehearsal": ("one-rank process group: exchange calls executed and captured" if (dp and world == 1) else None), "hip_graph": eng.graph_active,
                       "hip_graph_fallback": eng.graph_fallback_reason, "compress_mapping_grad": eng.compress_map and dp, "shard_mapping_update": eng.shard_map,
                       "fused_real_fake_pass": not args.no_fuse,
                       "flops_per_image_step": f_step, "flops_executed_per_image_step": f_exec,
                       "pruned": None if args.dense_top_block else "top encoder block behind its attention runs on the CLS rows only (the classifier reads "
                                 "nothing else, modules.py:195); values and gradients unchanged; --dense-top-block 1 computes every row",
                       "losses_finite": ok, "last_losses": [round(x, 4) for x in lv]},
            "roofline": roof,
        }
        # The other single-GPU configurations of BASELINE.json (configs[3], configs[4]: C4 and C5 at B = 128 per GPU, C5 with fp8 attention),
        # timed by the SAME driver-run command: behind the headline's timed region, so the C2 number is untouched.
        if world == 1 and not dp and args.workload == "c2" and not args.no_extra_workloads and args.gp == 0.0:
            eng.close()
            del eng
            extras = []
            # c4 / c5: BASELINE's larger configurations; c2-wgan-gp: the headline configuration on the reference's Wasserstein step with the
            # gradient penalty (training.py:67-125, lambda_gp = 10) - SURVEY 8f row f2
            for wl in ("c4", "c5", "c2-wgan-gp"):
                try:  # an extra measurement must never cost the headline its line
                    if wl == "c2-wgan-gp":
                        g2, G2, e2, f8 = make("c2", B, -1, loss="wasserstein", gp=10.0)
                        B2 = B
                    else:
                        g2, G2, e2, f8 = make(wl, GEO[wl]["batch"], -1)
                        B2 = GEO[wl]["batch"]
                    gen2 = torch.Generator(device=dev).manual_seed(99)
                    r2 = [torch.rand(B2, 3, g2["image"], g2["image"], device=dev, generator=gen2) * 2 - 1 for _ in range(2)]
                    for i in range(4):
                        e2.step(r2[i % 2])
                    torch.cuda.synchronize()
                    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    n2 = 10
                    a0.record()
                    for i in range(n2):
                        l2 = e2.step(r2[i % 2])
                    a1.record()
                    torch.cuda.synchronize()
                    ms = a0.elapsed_time(a1) / n2
                    fs, fe = step_flops(g2, G2, f8)
                    ips2 = B2 / ms * 1e3
                    cfg_txt = (f"BASELINE.json configs[{3 if wl == 'c4' else 4}] geometry on one GPU: {g2['image']}x{g2['image']} patch {g2['patch']}, "
                               f"E={g2['embed']}, {g2['heads']} heads, 6 blocks, patch-grid SLN/SIREN generator, B={B2}") if wl != "c2-wgan-gp" else (
                               f"the headline configuration (C2, B={B2}) on the Wasserstein step with the gradient penalty, gp_weight 10 "
                               f"(penalty as one C call: {bool(e2.gp_c_call)})")
                    extras.append({"workload": wl, "config": cfg_txt,
                                   "fp8_attention": f8, "steps": n2, "warmup": 4, "ms_per_step": round(ms, 4), "images_per_sec": round(ips2, 1),
                                   "step_tflops": round(ips2 * fs / 1e12, 1), "step_frac_of_peak": round(ips2 * fs / 1e12 / PEAK_BF16_TFLOPS, 4),
                                   "step_frac_of_peak_executed": round(ips2 * fe / 1e12 / PEAK_BF16_TFLOPS, 4), "hip_graph": e2.graph_active,
                                   "losses_finite": all(x == x and abs(x) < 1e4 for x in l2.cpu().tolist())})
                    if wl == "c2-wgan-gp":  # (no flop count of the penalty's four extra passes is claimed)
                        for k in ("step_tflops", "step_frac_of_peak", "step_frac_of_peak_executed"):
                            extras[-1][k] = None
                    e2.close()
                    del e2, G2, r2
                    torch.cuda.empty_cache()
                except Exception as exc:  # noqa: BLE001
                    extras.append({"workload": wl, "error": f"{type(exc).__name__}: {exc}"})
            out["extra_workloads"] = extras
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
