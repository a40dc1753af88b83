"""GanEngine: the alternating G/D step (src/v2/training.py:170-211) as a short list of C calls.

One step on each rank (one process per GPU):
  1. D.grad = 0 ; fake = G(z)                                   (gen forward, saved for step 4)
  2. D([real ; fake.detach()]) -> loss_real + loss_fake -> backward into D.grad
     (the reference runs the two halves as separate passes, training.py:182-194; both accumulate
      into the same .grad before ONE optimizer step and the ViT has no cross-sample op, so running
      them as one 2B batch is the same computation up to fp32 summation order)
  3. all-reduce(D.grad) over ranks ; fused AdamW on D            (training.py:197)
  4. G.grad = 0 ; D(fake) with the UPDATED D -> loss(label = real) -> backward for the input
     gradient only (D's weight gradients of this pass are discarded by the next zero_grad,
     training.py:177, so they are never computed) -> gen backward
  5. all-reduce(G.grad) ; fused AdamW on G                        (training.py:211)
Nothing synchronises with the host; with ``use_graph`` the whole step is replayed as one hipGraph.
"""
from __future__ import annotations

import ctypes as C
import time
import warnings
import weakref
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib, flat
from .dist import GradSync, backward_pieces
from .generator import SirenGenerator
from .modules import ViTDiscriminator, VisionTransformer

LOSS_KINDS = {"ns": 0, "hinge": 1, "wasserstein": 2}  # "wasserstein": the critic losses of src/v2/training.py:72,97


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class GanEngine:
    def __init__(self, discriminator, generator: SirenGenerator, batch: int, loss: str = "ns",
                 lr_d: float = 5e-4, lr_g: float = 5e-4, weight_decay: float = 1e-3, betas=(0.9, 0.999),
                 eps: float = 1e-8, fuse_real_fake: bool = True, use_graph: bool = False,
                 d_dropout: Optional[float] = None, g_dropout: Optional[float] = None, seed: int = 0,
                 concurrent_wgrad: bool = False, clip_d: Optional[float] = None, clip_g: Optional[float] = None,
                 diversity_weight: float = 0.0, instance_noise: float = 0.0,
                 process_group: Optional["dist.ProcessGroup"] = None, external_noise: bool = False,
                 two_stream: bool = False, compress_mapping_grad: bool = False, shard_mapping_update: bool = False, gp_weight: float = 0.0,
                 exchange_single_rank: bool = False, dense_top_block: bool = False, gp_autograd: bool = False):
        """concurrent_wgrad: the discriminator's weight gradients on a side stream beside its input gradients.  Off by default
        since the persistent GEMMs (csrc/gemm_wr.hip, gemm_tn.hip: their workgroups hold the CUs for a whole launch) - the
        side stream measured 6.70 against 6.67 ms/step.
        clip_d / clip_g: max gradient norms of ``clip_grad_norm_`` before each optimizer step (the reference's
        Wasserstein step uses 5.0 / 0.5, src/v2/training.py:78,104); None = no clipping (its live loop).
        diversity_weight: weight of ``diversity_loss(fake)`` in the generator loss (0.1 there, training.py:73-74; computed
        over this rank's batch - under data parallelism it is NOT the global-batch quantity).
        instance_noise: sigma of the Gaussian noise added to the discriminator's real and fake inputs in its own step
        (0.1 there, training.py:83-90); the generator's pass through D sees the clean fake.
        two_stream: run the step as two concurrent chains on two HIP streams (single GPU only) - D(real) forward/backward
        beside [G forward, D(fake) forward/backward], which is also the reference's own pass structure (two separate D
        passes, training.py:182-194), then the generator's pass through D as two half-batches side by side.  Kernels of the
        two chains are in different phases (a GEMM main loop next to another GEMM's epilogue, a LayerNorm next to a
        GEMM), which the single-chain step cannot be: every launch of this model covers the chip about once.
        gp_weight: weight of the WGAN-GP gradient penalty in the discriminator loss (``c.lambda_gp`` of training.py:106; the
        field is missing from the reference's Config).  The penalty is ONE C call, ``vg_vit_penalty``: forward, input-gradient
        backward, its double backward and the second backward as kernel sequences with the engine's counter-based dropout masks
        (the discriminator is in train mode there, as in the reference; the input gradients + LayerNorm backwards are fused where
        the full-row kernels take the shape); ``gp_autograd=True`` takes the operator-set path below, the form the C call is
        tested against.  That path runs through torch autograd over the twice-
        differentiable operator set (penalty.py) on the discriminator's real / fake inputs of this step and accumulates
        into the same gradient buffer before the exchange and AdamW.  With ``use_graph`` the autograd passes are captured with
        the rest of the step (every operator is an enqueue-only kernel call; epsilon and the penalty pass's dropout masks come
        from torch's graph-safe generator), so a replay costs no Python dispatch; if the capture fails the step runs eager, loudly.
        compress_mapping_grad (data parallel only, default OFF): exchange the gradient of the generator's mapping Linear - 50 MB of
        the generator's 64 MB, final only when the step's last kernel has run - as bf16 (see GradSync.reduce_range).  The sum is
        then formed in bf16 inside the collective (8 mantissa bits, error growing with the world size), so the default step is
        the exact fp32 all-reduce and a caller that wants the halved link traffic opts in (bench.py does and says so in its line).
        shard_mapping_update (data parallel only, default OFF): the same layer's gradient is reduce-SCATTERED in fp32 (each rank receives
        the exact sum of one 1/world share), every rank runs AdamW on its share alone, and the updated bf16 shadow shares - what the
        GEMMs read - are all-gathered (GradSync.reduce_scatter_range / all_gather_range): exact sums like the default, 3/4 of its bytes
        on the links, 1/world of AdamW's traffic on the layer.  The fp32 master (and AdamW's moments) of the shares a rank does not
        own go stale on that rank: ``gather_master()`` brings the master up to date before a ``state_dict()`` is taken.  Not with
        ``clip_g`` (the clipping norm is taken over the whole gradient) nor together with ``compress_mapping_grad``.
        exchange_single_rank: run the staged backward and its all-reduces on a one-rank group as well (tests: the RCCL
        collectives inside a captured step, on a box with one GPU).
        dense_top_block: compute EVERY row of the top encoder block like the reference's operator graph does.  Default off: behind
        its attention that block runs on the B CLS rows only - the classifier reads nothing else (modules.py:195) and the gradient of the
        other rows is exactly zero - with the same logits and gradients (tests/test_engine_gpu.py compares the two); the switch exists
        for A/B measurements (``bench.py --dense-top-block 1``).
        use_graph: replay the step as one hipGraph.  On more than one rank the capture includes the RCCL all-reduces (backend
        "nccl"); when the capture is not possible (gloo process group, a torch build that cannot capture the collective) the
        engine says so loudly (warning + ``graph_fallback_reason``) and runs eager - it never falls back silently.
        external_noise: the latent batch is supplied by the caller (``step(real, z)``) instead of being drawn on the
        device inside the step - what parity tests use to give their CPU checker and the engine the same noise, also under
        hipGraph replay."""
        vit = discriminator.vit if isinstance(discriminator, ViTDiscriminator) else discriminator
        if not isinstance(vit, VisionTransformer) or not isinstance(generator, SirenGenerator):
            raise TypeError("GanEngine needs a ViTDiscriminator/VisionTransformer and a SirenGenerator")
        self.vit, self.gen = vit, generator
        self.dev = vit._flat.flat.device
        if self.dev.type != "cuda" or generator._flat.flat.device != self.dev:
            raise RuntimeError("GanEngine: both networks must be on the same cuda device (no CPU fallback)")
        if loss not in LOSS_KINDS:
            raise ValueError(f"loss must be one of {sorted(LOSS_KINDS)}")
        self.B, self.kind = int(batch), LOSS_KINDS[loss]
        # dropout probabilities: default = what the modules would apply in their current train/eval mode
        self.p_d = float(vit._dropout_p if vit.training else 0.0) if d_dropout is None else float(d_dropout)
        self.p_g = float(generator.dropout_p if generator.training else 0.0) if g_dropout is None else float(g_dropout)
        self.seed = int(seed)
        self.fuse = bool(fuse_real_fake)
        self.hyp = dict(lr_d=lr_d, lr_g=lr_g, wd=weight_decay, b1=betas[0], b2=betas[1], eps=eps)
        self.clip_d, self.clip_g = clip_d, clip_g
        self.dp_chunks = 3  # pieces of the D / G backward whose gradient exchange overlaps the remaining backward
        self.compress_map = bool(compress_mapping_grad)
        self._want_shard_map = bool(shard_mapping_update)
        if self._want_shard_map and (self.compress_map or clip_g is not None):
            raise ValueError("shard_mapping_update excludes compress_mapping_grad and clip_g")
        self.dense_top = int(bool(dense_top_block))
        self.gp_w = float(gp_weight)
        self.gp_loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.gp_epsilon: Optional[torch.Tensor] = None  # tests: a fixed epsilon [B,1,1,1] instead of torch.rand
        if self.gp_w != 0.0 and two_stream:
            raise ValueError("gp_weight: the gradient penalty runs through torch autograd on one stream and cannot be forked")
        if self.gp_w != 0.0 and bool(getattr(vit, "attention_fp8", False)):
            # the penalty path (ops2.py) differentiates the bf16 attention kernels: with fp8 operands in the trained network
            # it would penalise a slightly different function than the one being trained
            raise ValueError("gp_weight: the gradient penalty is built on the bf16 attention kernels; switch attention_fp8 off")
        self.gp_c_call = self.gp_w != 0.0 and not gp_autograd and not bool(getattr(vit, "attention_fp8", False))
        self.div_w = float(diversity_weight)
        self.inst_sigma = float(instance_noise)
        self.external_noise = bool(external_noise)
        self.two_stream = bool(two_stream)
        self.div_loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.clip_scratch = torch.zeros(2, 1 + 1024, dtype=torch.float32, device=self.dev)  # [net][norm, partials]
        self.pg = process_group
        self.sync = GradSync(process_group, self.dev, overlap=True, single_rank=exchange_single_rank)
        self.world = self.sync.world
        g_ = generator._dims
        # (a layer that does not divide over the ranks keeps the all-reduce; a one-rank group - `exchange_single_rank` - runs the same calls)
        self.shard_map = self._want_shard_map and self.sync.active and (g_.T * g_.E * g_.Z) % (4 * self.world) == 0
        # latent noise drawn on the device (vg_step_inputs): one stream per (seed, rank)
        self._noise_seed = (self.seed * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03 * (self.sync.rank + 1)) & 0xFFFFFFFFFFFFFFFF
        d, g = vit._dims, generator._dims
        if g.T * g.CW != d.C * d.IH * d.IH:
            raise ValueError("generator output does not match the discriminator's image shape")
        L = _lib.lib()
        B, dev = self.B, self.dev
        if self.two_stream:
            if self.world > 1:
                raise ValueError("two_stream is a single-GPU schedule (the data-parallel path overlaps the exchange instead)")
            if B % 2:
                raise ValueError("two_stream needs an even batch")
            self.fuse = False
        nD = 2 * B if self.fuse else B
        self.Kc = d.Kc
        self.ws_d = torch.empty(L.vg_vit_ws_bytes(C.byref(d), nD), dtype=torch.uint8, device=dev)
        if self.two_stream:  # second chain: its own workspace, gradient buffer and stream
            self.ws_d2 = torch.empty(L.vg_vit_ws_bytes(C.byref(d), B), dtype=torch.uint8, device=dev)
            self.grad2 = torch.zeros_like(vit._flat.grad)
            self.side = torch.cuda.Stream(device=dev)
        if self.gp_c_call:  # the penalty's own passes: its forward runs in ws_d (the step's passes come after it), the rest here
            self.ws_gp = torch.empty(L.vg_vit_penalty_ws_bytes(C.byref(d), B), dtype=torch.uint8, device=dev)
            self.gp_eps = torch.empty(B, dtype=torch.float32, device=dev)
        self.ws_g = torch.empty(L.vg_gen_ws_bytes(C.byref(g), B), dtype=torch.uint8, device=dev)
        self.imgs = torch.empty(2 * B, d.C, d.IH, d.IH, dtype=torch.bfloat16, device=dev)  # [real ; fake]
        self.dfake = torch.empty(B, d.C, d.IH, d.IH, dtype=torch.bfloat16, device=dev)
        if self.inst_sigma > 0.0:  # noisy copy of [real ; fake] for the D step, and the noise itself (kept for inspection / tests)
            self.inoise = torch.empty(2 * B, d.C, d.IH, d.IH, dtype=torch.float32, device=dev)
            self.imgs_noisy = torch.empty_like(self.imgs)
        self.div_scratch = torch.zeros((d.C * d.IH * d.IH + 15) // 16, dtype=torch.float32, device=dev)
        self.logits = torch.empty(2 * B, d.Kc, dtype=torch.float32, device=dev)
        self.dlogits = torch.empty(2 * B, d.Kc, dtype=torch.float32, device=dev)
        self.z = torch.empty(B, g.Z, dtype=torch.float32, device=dev)
        self.losses = torch.zeros(3, dtype=torch.float32, device=dev)  # d_real, d_fake, g
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        fd, fg = vit._flat, generator._flat
        self.m_d, self.v_d = torch.zeros_like(fd.flat), torch.zeros_like(fd.flat)
        self.m_g, self.v_g = torch.zeros_like(fg.flat), torch.zeros_like(fg.flat)
        fd.refresh_shadow()
        fg.refresh_shadow()
        self.ctx = _lib.context() if concurrent_wgrad else None
        # a load_state_dict into either network (directly or through a container such as ViTGAN) copies into the flat
        # master buffers in place: refresh the bf16 shadows the GEMMs read, or the next step runs on stale weights
        # (the hook holds the engine weakly: a strong reference from the module would keep every engine ever built on it -
        # workspaces, optimizer moments - alive, and re-run the refresh of stale engines on every later load_state_dict)
        me = weakref.ref(self)

        def _hook(_mod, _keys):
            eng = me()
            if eng is not None:
                eng.sync_from_modules()
        self._hooks = [m.register_load_state_dict_post_hook(_hook) for m in (vit, generator)]
        self.steps = 0
        self._graph = None
        self._use_graph = bool(use_graph)
        self.graph_fallback_reason: Optional[str] = None
        if self._use_graph and self.sync.active:
            backend = dist.get_backend(process_group)
            if backend != "nccl":
                self._graph_fallback(f"process-group backend '{backend}' cannot be captured in a hipGraph (only nccl = RCCL can)")

    def _graph_fallback(self, reason: str) -> None:
        self._use_graph = False
        self._graph = None
        self.graph_fallback_reason = reason
        warnings.warn(f"GanEngine: hipGraph replay was requested but the step runs EAGER: {reason}", RuntimeWarning, stacklevel=3)

    @property
    def graph_active(self) -> bool:
        """True when step() replays a captured hipGraph (after the first call), False in eager mode."""
        return self._use_graph

    def close(self) -> None:
        """Detach from the modules (load_state_dict hooks) and drop the captured graph and workspaces."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def _nets(self):
        fd, fg = self.vit._flat, self.gen._flat
        # masks: host seed (fixed per pass) mixed on the device with the step counter, so a replayed hipGraph
        # still draws fresh masks; pass A = [real;fake] (or real), B = fake, C = generator pass through D
        step_ptr = self.step_t.data_ptr()
        mk = lambda i, g=None, ctx=True: _lib.VgVitNet(self.vit._dims, fd.flat.data_ptr(), fd.shadow.data_ptr(),  # noqa: E731
                                                       (fd.grad if g is None else g).data_ptr(), self.p_d, self.seed * 8 + i, step_ptr,
                                                       self.ctx if ctx else None, int(self.vit.attention_fp8), self.dense_top)
        if self.two_stream:  # chains run side by side: no third stream inside a pass; the fake chain accumulates into grad2
            return (mk(0, ctx=False), mk(1, self.grad2, ctx=False), mk(2, ctx=False), mk(3, ctx=False)), self._gen_net(step_ptr)
        return (mk(0), mk(1), mk(2)), self._gen_net(step_ptr)

    def _gen_net(self, step_ptr):
        fg, tab = self.gen._flat, self.gen.fourier_table
        return _lib.VgGenNet(self.gen._dims, fg.flat.data_ptr(), fg.shadow.data_ptr(), fg.grad.data_ptr(), self.p_g, self.seed * 8 + 7, step_ptr,
                             None if tab is None else tab.data_ptr())

    def _d_backward(self, nd, n_img: int, dl, want_w: int, dimg, st) -> None:
        """D backward; under data parallelism in ``dp_chunks`` pieces (head + upper blocks first) so that the all-reduce
        of each finished piece - a contiguous tail of the flat gradient buffer - overlaps the backward of the blocks
        below it; only the last piece's exchange is exposed."""
        L = _lib.lib()
        nL = self.vit._dims.L
        if not self.sync.active or not want_w:
            _lib.check(L.vg_vit_backward(C.byref(nd), n_img, _p(self.ws_d), dl, dimg, want_w, st), "vg_vit_backward")
            return
        fd = self.vit._flat
        lay = flat.vit_layout(self.vit._dims)
        for s0, s1, lo, hi in backward_pieces(nL, self.dp_chunks, lay.layer0, lay.layer_stride, fd.total):
            _lib.check(L.vg_vit_backward_stages(C.byref(nd), n_img, _p(self.ws_d), dl, dimg, want_w, s0, s1, st), "vg_vit_backward_stages")
            self.sync.reduce_range(fd.grad, lo, hi)

    def _g_backward(self, ng, st) -> None:
        """G backward; under data parallelism in ``dp_chunks`` pieces like D's: SIREN head + upper blocks first, their
        gradients (a contiguous tail of the flat buffer) are exchanged while the lower blocks still run.  What is left
        exposed is the front of the buffer - learned embedding, mapping Linear, lowest blocks - which only completes
        with the last kernel; its 50 MB mapping-weight part goes over the links as bf16."""
        L = _lib.lib()
        fg = self.gen._flat
        if not self.sync.active:
            _lib.check(L.vg_gen_backward(C.byref(ng), self.B, _p(self.ws_g), _p(self.dfake), st), "vg_gen_backward")
            return
        lay = flat.gen_layout(self.gen._dims)
        d = self.gen._dims
        for s0, s1, lo, hi in backward_pieces(d.L, self.dp_chunks, lay.layer0, lay.layer_stride, fg.total):
            _lib.check(L.vg_gen_backward_stages(C.byref(ng), self.B, _p(self.ws_g), _p(self.dfake), s0, s1, st), "vg_gen_backward_stages")
            if lo == 0 and (self.compress_map or self.shard_map):  # [embedding | mapping weight | mapping bias, lowest blocks]
                w0, w1 = lay.map_w, lay.map_w + d.T * d.E * d.Z
                self.sync.reduce_range(fg.grad, 0, w0)
                if self.shard_map:
                    self.sync.reduce_scatter_range(fg.grad, w0, w1)   # this rank keeps the sum of its share only
                else:
                    self.sync.reduce_range(fg.grad, w0, w1, compress=True)
                self.sync.reduce_range(fg.grad, w1, hi)
            else:
                self.sync.reduce_range(fg.grad, lo, hi)

    def _map_range(self):
        lay, d = flat.gen_layout(self.gen._dims), self.gen._dims
        return lay.map_w, lay.map_w + d.T * d.E * d.Z

    def _adamw_g_sharded(self, st) -> None:
        """The generator's AdamW with the mapping Linear sharded: the whole buffer but that layer as usual, of the layer this rank's
        share only; then the updated bf16 shadow shares to every rank (the GEMMs of every rank read the whole shadow)."""
        fg, h, L = self.gen._flat, self.hyp, _lib.lib()
        w0, w1 = self._map_range()
        a, b = self.sync.share(w0, w1)

        def upd(lo, hi):
            if hi <= lo:
                return
            off = lambda t, es: C.c_void_p(t.data_ptr() + es * lo)  # noqa: E731
            _lib.check(L.vg_adamw_step(off(fg.flat, 4), off(fg.grad, 4), off(self.m_g, 4), off(self.v_g, 4), off(fg.shadow, 2), hi - lo,
                                       self.hyp["lr_g"], h["b1"], h["b2"], h["eps"], h["wd"], 0, _p(self.step_t), 1.0 / self.world, st), "vg_adamw_step")
        upd(0, w0)
        upd(a, b)
        upd(w1, fg.total)
        self.sync.all_gather_range(fg.shadow, w0, w1)
        self.sync.wait()

    def gather_master(self) -> None:
        """shard_mapping_update: bring the fp32 master of the mapping Linear up to date on every rank (each rank updates its share
        only); call it before ``state_dict()`` / a checkpoint.  A no-op otherwise."""
        if not self.shard_map:
            return
        w0, w1 = self._map_range()
        self.sync.all_gather_range(self.gen._flat.flat, w0, w1)
        self.sync.wait()
        torch.cuda.current_stream().synchronize()

    def _adamw(self, fp, m, v, lr, st, clip=None, slot=0):
        h = self.hyp
        if clip is not None:  # on the exchanged (global) gradient, like clip_grad_norm_ before optimizer.step()
            _lib.check(_lib.lib().vg_grad_clip(_p(fp.grad), fp.total, 1.0 / self.world, float(clip), _p(self.clip_scratch[slot]), st),
                       "vg_grad_clip")
        _lib.check(_lib.lib().vg_adamw_step(_p(fp.flat), _p(fp.grad), _p(m), _p(v), _p(fp.shadow), fp.total, lr, h["b1"], h["b2"],
                                            h["eps"], h["wd"], 0, _p(self.step_t), 1.0 / self.world, st), "vg_adamw_step")

    def _loss(self, lo, n, role, slot, st):
        L = _lib.lib()
        off = 4 * lo * self.Kc
        _lib.check(L.vg_gan_loss(C.c_void_p(self.logits.data_ptr() + off), C.c_void_p(self.dlogits.data_ptr() + off),
                                 C.c_void_p(self.losses.data_ptr() + 4 * slot), n * self.Kc, self.kind, role, 1.0, st), "vg_gan_loss")

    def _enqueue_two_stream(self) -> None:
        """The step as two concurrent chains (see ``two_stream``).  Everything is enqueued from this thread; the second chain
        forks from and joins the current stream through events, so the whole step is still one capturable graph."""
        L, B = _lib.lib(), self.B
        s0, s1 = torch.cuda.current_stream(), self.side
        st0, st1 = C.c_void_p(s0.cuda_stream), C.c_void_p(s1.cuda_stream)
        (nd_a, nd_b, nd_c, nd_d), ng = self._nets()
        fd, fg = self.vit._flat, self.gen._flat
        img_bytes = self.imgs[0].numel() * 2
        Kc4 = 4 * self.Kc
        off_img = lambda t, n: C.c_void_p(t.data_ptr() + n * img_bytes)  # noqa: E731
        off_log = lambda t, n: C.c_void_p(t.data_ptr() + n * Kc4)       # noqa: E731
        fake_ptr = off_img(self.imgs, B)
        _lib.check(L.vg_zero_tick(_p(fd.grad), fd.total, _p(self.step_t), st0), "vg_zero_tick")
        self.grad2.zero_()
        d_in = self.imgs
        s1.wait_stream(s0)
        # chain 1 (side stream): G forward, then D on the fake batch (weight gradients into grad2)
        with torch.cuda.stream(s1):
            _lib.check(L.vg_gen_forward(C.byref(ng), B, _p(self.z), _p(self.ws_g), fake_ptr, st1), "vg_gen_forward")
            if self.inst_sigma > 0.0:
                self.inoise[B:].normal_()
                torch.add(self.imgs[B:].float(), self.inoise[B:], alpha=self.inst_sigma, out=self.inoise[B:])
                self.imgs_noisy[B:].copy_(self.inoise[B:])
            src = off_img(self.imgs_noisy if self.inst_sigma > 0.0 else self.imgs, B)
            _lib.check(L.vg_vit_forward(C.byref(nd_b), B, src, 1, _p(self.ws_d2), off_log(self.logits, B), st1), "vg_vit_forward")
            _lib.check(L.vg_gan_loss(off_log(self.logits, B), off_log(self.dlogits, B), C.c_void_p(self.losses.data_ptr() + 4), B * self.Kc,
                                     self.kind, 1, 1.0, st1), "vg_gan_loss")
            _lib.check(L.vg_vit_backward(C.byref(nd_b), B, _p(self.ws_d2), off_log(self.dlogits, B), None, 1, st1), "vg_vit_backward")
        # chain 0 (this stream): D on the real batch
        if self.inst_sigma > 0.0:
            self.inoise[:B].normal_()
            torch.add(self.imgs[:B].float(), self.inoise[:B], alpha=self.inst_sigma, out=self.inoise[:B])
            self.imgs_noisy[:B].copy_(self.inoise[:B])
            d_in = self.imgs_noisy
        _lib.check(L.vg_vit_forward(C.byref(nd_a), B, _p(d_in), 1, _p(self.ws_d), _p(self.logits), st0), "vg_vit_forward")
        self._loss(0, B, 0, 0, st0)
        _lib.check(L.vg_vit_backward(C.byref(nd_a), B, _p(self.ws_d), _p(self.dlogits), None, 1, st0), "vg_vit_backward")
        s0.wait_stream(s1)
        fd.grad.add_(self.grad2)  # the two passes accumulate into one .grad in the reference (training.py:184,194)
        self._adamw(fd, self.m_d, self.v_d, self.hyp["lr_d"], st0, self.clip_d, 0)
        fg.grad.zero_()
        # generator's pass through the updated D: two half-batches side by side (no weight gradients, nothing shared)
        h = B // 2
        s1.wait_stream(s0)
        with torch.cuda.stream(s1):
            _lib.check(L.vg_vit_forward(C.byref(nd_d), h, off_img(self.imgs, B + h), 1, _p(self.ws_d2), off_log(self.logits, h), st1), "vg_vit_forward")
        _lib.check(L.vg_vit_forward(C.byref(nd_c), h, fake_ptr, 1, _p(self.ws_d), _p(self.logits), st0), "vg_vit_forward")
        s0.wait_stream(s1)
        self._loss(0, B, 2, 2, st0)  # one mean over the whole batch
        s1.wait_stream(s0)
        with torch.cuda.stream(s1):
            _lib.check(L.vg_vit_backward(C.byref(nd_d), h, _p(self.ws_d2), off_log(self.dlogits, h), off_img(self.dfake, h), 0, st1), "vg_vit_backward")
        _lib.check(L.vg_vit_backward(C.byref(nd_c), h, _p(self.ws_d), _p(self.dlogits), _p(self.dfake), 0, st0), "vg_vit_backward")
        s0.wait_stream(s1)
        if self.div_w != 0.0:
            Dn = self.dfake[0].numel()
            _lib.check(L.vg_diversity_loss(fake_ptr, _p(self.dfake), _p(self.div_loss), _p(self.div_scratch), B, Dn, self.div_w, st0),
                       "vg_diversity_loss")
        _lib.check(L.vg_gen_backward(C.byref(ng), B, _p(self.ws_g), _p(self.dfake), st0), "vg_gen_backward")
        self._adamw(fg, self.m_g, self.v_g, self.hyp["lr_g"], st0, self.clip_g, 1)

    def _inputs(self, real: torch.Tensor) -> None:
        """The step's inputs, ONE launch in front of the step proper (and outside its hipGraph, so it reads the caller's tensor
        directly - no staging copy): imgs[:B] = bf16(real), and unless the caller supplies it, the latent batch z ~ N(0, 1)
        (construct_noise(), training.py:35-42 / gan.py:231-232), counter-based on (seed, rank, steps done so far)."""
        B = self.B
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # the kernel dereferences the caller's pointer with 16-byte loads on THIS engine's device: anything else (another GPU's
        # tensor, an unaligned view) goes through torch's copy, which handles it
        direct = (real.dtype == torch.float32 and real.is_contiguous() and real[0].numel() == self.imgs[0].numel() and real.numel() % 4 == 0
                  and real.device == self.imgs.device and real.data_ptr() % 16 == 0)
        if not direct:
            self.imgs[:B].copy_(real)
        want_z = not self.external_noise
        if direct or want_z:
            _lib.check(_lib.lib().vg_step_inputs(_p(real) if direct else None, _p(self.imgs), real.numel() if direct else 0,
                                                 _p(self.z) if want_z else None, self.z.numel() if want_z else 0, self._noise_seed,
                                                 _p(self.step_t), st), "vg_step_inputs")

    def _enqueue(self, real: torch.Tensor) -> None:
        """Enqueue one full step on the current stream (no host sync)."""
        self._inputs(real)
        self._enqueue_body()

    def _enqueue_body(self) -> None:
        """Everything of a step behind its inputs (``_inputs``): what the hipGraph captures."""
        if self.two_stream:
            return self._enqueue_two_stream()
        L, B = _lib.lib(), self.B
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        (nd, nd_b, nd_c), ng = self._nets()
        fd, fg = self.vit._flat, self.gen._flat
        img_bytes = self.imgs[0].numel() * 2
        fake_ptr = C.c_void_p(self.imgs.data_ptr() + B * img_bytes)
        # gan.discriminator.zero_grad() (training.py:177) and the device step counter += 1, one launch
        _lib.check(L.vg_zero_tick(_p(fd.grad), fd.total, _p(self.step_t), st), "vg_zero_tick")
        _lib.check(L.vg_gen_forward(C.byref(ng), B, _p(self.z), _p(self.ws_g), fake_ptr, st), "vg_gen_forward")
        d_in = self.imgs
        if self.inst_sigma > 0.0:  # noisy_real / noisy_fake of training.py:83-90 (the clean fake stays in self.imgs for pass C)
            self.inoise.normal_()
            torch.add(self.imgs.float(), self.inoise, alpha=self.inst_sigma, out=self.inoise)
            self.imgs_noisy.copy_(self.inoise)
            d_in = self.imgs_noisy
        if self.gp_c_call:  # gradient_penalty(D, noisy_real, noisy_fake) joins the D loss (training.py:101-106): one C call
            if self.gp_epsilon is not None:
                torch.add(self.gp_epsilon.reshape(-1).float(), 0.0, out=self.gp_eps)  # (an elementwise kernel, not a D2D copy: no memcpy / memset nodes in the captured step)
            else:
                self.gp_eps.uniform_()  # epsilon = torch.rand(B, 1, 1, 1), utils.py:129
            pnet = _lib.VgVitNet(self.vit._dims, fd.flat.data_ptr(), fd.shadow.data_ptr(), fd.grad.data_ptr(), self.p_d, self.seed * 8 + 3,
                                 self.step_t.data_ptr(), None, 0, 1)
            _lib.check(L.vg_vit_penalty(C.byref(pnet), B, _p(d_in), C.c_void_p(d_in.data_ptr() + B * img_bytes), _p(self.gp_eps), self.gp_w,
                                        _p(self.ws_d), _p(self.ws_gp), _p(self.gp_loss), st), "vg_vit_penalty")
        elif self.gp_w != 0.0:
            from .penalty import gradient_penalty
            fd.attach_grads()
            disc = self.vit
            from . import ops2
            pen = gradient_penalty(disc, d_in[:B], d_in[B:], epsilon=self.gp_epsilon)
            with ops2.deferred_weight_grads(fd.grad):  # the block Linears' weight gradients: grouped per block, straight into the flat buffer
                (self.gp_w * pen).backward()   # the rest accumulates into the same buffer through the parameters' .grad (views of it)
            self.gp_loss.copy_(pen.detach().reshape(1))
        if self.fuse:
            _lib.check(L.vg_vit_forward(C.byref(nd), 2 * B, _p(d_in), 1, _p(self.ws_d), _p(self.logits), st), "vg_vit_forward")
            # D(real) -> slot 0, D(fake) -> slot 1: both halves of the fused pass in one launch
            _lib.check(_lib.lib().vg_gan_loss_pair(_p(self.logits), _p(self.dlogits), _p(self.losses), B * self.Kc, 0, B * self.Kc, 1, self.kind,
                                                   1.0, st), "vg_gan_loss_pair")
            self._d_backward(nd, 2 * B, _p(self.dlogits), 1, None, st)
        else:
            for half, role in ((0, 0), (1, 1)):
                src = C.c_void_p(d_in.data_ptr() + half * B * img_bytes)
                lg = C.c_void_p(self.logits.data_ptr() + 4 * half * B * self.Kc)
                dl = C.c_void_p(self.dlogits.data_ptr() + 4 * half * B * self.Kc)
                net = nd if half == 0 else nd_b
                _lib.check(L.vg_vit_forward(C.byref(net), B, src, 1, _p(self.ws_d), lg, st), "vg_vit_forward")
                self._loss(half * B, B, role, role, st)
                if half == 0:
                    _lib.check(L.vg_vit_backward(C.byref(net), B, _p(self.ws_d), dl, None, 1, st), "vg_vit_backward")
                else:  # second pass finishes D.grad: exchange it as it completes
                    self._d_backward(net, B, dl, 1, None, st)
        self.sync.wait()
        self._adamw(fd, self.m_d, self.v_d, self.hyp["lr_d"], st, self.clip_d, 0)
        fg.grad.zero_()            # gan.generator.zero_grad(), training.py:199
        _lib.check(L.vg_vit_forward(C.byref(nd_c), B, fake_ptr, 1, _p(self.ws_d), _p(self.logits), st), "vg_vit_forward")
        self._loss(0, B, 2, 2, st)
        _lib.check(L.vg_vit_backward(C.byref(nd_c), B, _p(self.ws_d), _p(self.dlogits), _p(self.dfake), 0, st), "vg_vit_backward")
        if self.div_w != 0.0:  # total_gen_loss = loss + w * diversity_loss(fake_images): its gradient joins dL/d fake
            Dn = self.dfake[0].numel()
            _lib.check(L.vg_diversity_loss(fake_ptr, _p(self.dfake), _p(self.div_loss), _p(self.div_scratch), B, Dn, self.div_w, st),
                       "vg_diversity_loss")
        self._g_backward(ng, st)
        self.sync.wait()
        if self.shard_map:
            self._adamw_g_sharded(st)
        else:
            self._adamw(fg, self.m_g, self.v_g, self.hyp["lr_g"], st, self.clip_g, 1)

    # ------------------------------------------------------------------------------------------
    def _state_tensors(self):
        """Everything a step changes that the next step reads (the training state held on the device)."""
        fd, fg = self.vit._flat, self.gen._flat
        return [fd.flat, fd.shadow, fg.flat, fg.shadow, self.m_d, self.v_d, self.m_g, self.v_g, self.step_t]

    def sync_from_modules(self, reset_optimizer: bool = False) -> None:
        """Call after the modules' parameters were changed behind the engine's back (``load_state_dict``, an in-place
        edit): refreshes the bf16 shadows the GEMMs read; ``reset_optimizer`` also clears AdamW's moments and step count
        (a fresh optimizer, which is what the reference has after a restart: it saves no optimizer state,
        training.py:218-226,262-263)."""
        self.vit._flat.refresh_shadow()
        self.gen._flat.refresh_shadow()
        if reset_optimizer:
            for t in (self.m_d, self.v_d, self.m_g, self.v_g, self.step_t):
                t.zero_()
            # the latent noise is keyed on the device step counter just cleared: move to a fresh stream, keyed on the steps this
            # engine has really done, so a restarted run does not replay the first run's latent sequence
            self._noise_seed = (self._noise_seed * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03 * (self.steps + 1)) & 0xFFFFFFFFFFFFFFFF

    def step(self, real: torch.Tensor, z: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Run one G/D step on ``real`` [B,C,IH,IW] (cuda).  Returns the device tensor
        [loss_d_real, loss_d_fake, loss_g] of this step without synchronising.  ``z`` [B, Z]: the latent batch, required
        iff the engine was built with ``external_noise=True``."""
        if real.shape[0] != self.B or not real.is_cuda:
            raise ValueError("real must be a cuda tensor with the engine's batch size")
        if (z is not None) != self.external_noise:
            raise ValueError("pass z exactly when the engine was built with external_noise=True")
        if not (self.vit._flat.aliased() and self.gen._flat.aliased()):
            raise RuntimeError("module parameters were re-allocated; rebuild the GanEngine")
        if z is not None:
            self.z.copy_(z)
        self.steps += 1
        if not self._use_graph:
            self._enqueue(real)
            return self.losses
        if self._graph is None:
            # Warm-up on a side stream (allocator, lazily loaded code objects), then capture.  The warm-up is a real step:
            # the training state is saved before it and restored after it, so N calls of step() are N steps in graph
            # mode exactly as in eager mode (tests compare the two bit for bit).
            saved = [t.clone() for t in self._state_tensors()]
            s = torch.cuda.Stream()
            self._inputs(real)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._enqueue_body()
            torch.cuda.current_stream().wait_stream(s)
            for t, keep in zip(self._state_tensors(), saved):
                t.copy_(keep)
            graph = torch.cuda.CUDAGraph()
            # With a process group the RCCL watchdog THREAD polls the events of the collectives the warm-up step enqueued: under the default
            # ("global") capture mode such a hipEventQuery from another thread while this one captures is an error that invalidates the
            # capture and, raised inside the watchdog, ends the process (seen once in ~10 runs of the one-rank RCCL test).  "thread_local"
            # confines the restriction to the capturing thread, which is what a captured step with collectives needs.
            mode = "thread_local" if self.sync.active else "global"
            if self.sync.active:
                # ... and the watchdog gets the time to retire the warm-up's (finished) collectives from its list - it polls every 100 ms,
                # and collectives enqueued DURING a capture are never put on that list - so that it has nothing to query while we capture
                torch.cuda.synchronize()
                time.sleep(0.5)
            try:
                with torch.cuda.graph(graph, capture_error_mode=mode):
                    self._enqueue_body()
            except Exception as exc:  # only reachable with collectives or the autograd-driven penalty in the step: otherwise it is all our own enqueue-only calls
                if not self.sync.active and self.gp_w == 0.0:
                    raise
                torch.cuda.synchronize()
                for t, keep in zip(self._state_tensors(), saved):  # a broken capture must not have advanced the state
                    t.copy_(keep)
                self.sync._pending.clear()
                self._graph_fallback(f"capturing the step ({'collectives' if self.sync.active else 'gradient penalty through torch autograd'}) failed: "
                                     f"{type(exc).__name__}: {exc}")
                self._enqueue(real)
                return self.losses
            self._graph = graph
        self._inputs(real)
        self._graph.replay()
        return self.losses
