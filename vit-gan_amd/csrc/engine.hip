// Whole-network passes: sequences the gfx950 kernels for the v2 VisionTransformer (discriminator)
// and the v1 SLN/SIREN generator.  Host code only enqueues work on one stream: no allocation, no
// synchronisation, so a pass (or a whole G/D step) can be captured into a hipGraph by the caller.
//
// Data layout (all row-major, bf16 unless noted):
//   tokens      X[l]   [B*S, E]      residual stream entering block l (X[L] = trunk output)
//   qkv[l]             [B*S, 3E]     Q | K | V thirds, head h at columns h*HE.. of each third
//   flat params        fp32 master P, bf16 shadow Pb, fp32 grads G share ONE offset table
//                      (VgVitLayout / VgGenLayout): per block the four GEMM weights are contiguous so
//                      their split-K wgrad slabs fold into G with a single streaming kernel.
#include "../../include/vitgan_hip.h"
#include "vg_kernels.h"
#include "vg_row.h"

static inline long long al64(long long x) { return (x + 63) & ~63LL; }

// Optional execution context: a second stream + events so that the weight-gradient side of a backward
// (wgrad GEMMs, bias / LayerNorm-affine reductions - everything that only writes the gradient buffer) runs
// concurrently with the input-gradient chain of the next block and fills the CUs its short tails leave idle.
#define VG_CTX_EVENTS 72
struct VgCtx {
  hipStream_t side;
  hipEvent_t ev_main[VG_CTX_EVENTS], ev_side[VG_CTX_EVENTS];
};
extern "C" int vg_ctx_create(void** out) {
  if (!out) return -1;
  VgCtx* c = new VgCtx();
  hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return (int)e; }
  for (int i = 0; i < VG_CTX_EVENTS; ++i) {
    if ((e = hipEventCreateWithFlags(&c->ev_main[i], hipEventDisableTiming)) != hipSuccess) return (int)e;
    if ((e = hipEventCreateWithFlags(&c->ev_side[i], hipEventDisableTiming)) != hipSuccess) return (int)e;
  }
  *out = c;
  return 0;
}
extern "C" int vg_ctx_destroy(void* ctx) {
  if (!ctx) return 0;
  VgCtx* c = (VgCtx*)ctx;
  for (int i = 0; i < VG_CTX_EVENTS; ++i) { (void)hipEventDestroy(c->ev_main[i]); (void)hipEventDestroy(c->ev_side[i]); }
  (void)hipStreamDestroy(c->side);
  delete c;
  return 0;
}

// =============================================================================================
//                                       layouts
// =============================================================================================
extern "C" int vg_vit_layout(const VgVitDims* d, VgVitLayout* o) {
  if (!d || !o) return -1;
  const long long E = d->E, K = (long long)d->C * d->P * d->P, NP = (long long)(d->IH / d->P) * (d->IH / d->P);
  if (d->E % 128 || d->E % d->H || d->IH % d->P || (K & 7) || d->L < 1 || d->Kc < 1 || d->R < 1) return -3;
  const int HE = d->E / d->H;
  if (HE != 32 && HE != 64 && HE != 96) return -3;
  if (NP + 1 > 80) return -3;
  // depth: the backward queues 3 deferred folds per block + 3 more (classifier head, final LayerNorm, the pruned top block's CLS-summed
  // bias row) and uses one event pair per block (+1 for the join)
  if (3 * d->L + 3 > VG_MAX_FOLD_JOBS || d->L >= VG_CTX_EVENTS - 1) return -3;
  long long p = 0;
  o->conv_w = p; p = al64(p + E * K);
  o->conv_b = p; p = al64(p + E);
  o->pos = p; p = al64(p + NP * E);
  o->cls = p; p = al64(p + E);
  // per block (relative offsets)
  long long q = 0;
  o->wqkv = q; q += 3 * E * E;
  o->wo = q; q += E * E;
  o->w1 = q; q += (long long)d->R * E * E;
  o->w2 = q; q += (long long)d->R * E * E;
  o->layer_weights = q;
  o->ln1_w = q; q = al64(q + E);
  o->ln1_b = q; q = al64(q + E);
  o->bqkv = q; q = al64(q + 3 * E);
  o->bo = q; q = al64(q + E);
  o->ln2_w = q; q = al64(q + E);
  o->ln2_b = q; q = al64(q + E);
  o->b1 = q; q = al64(q + (long long)d->R * E);
  o->b2 = q; q = al64(q + E);
  o->layer0 = p; o->layer_stride = q;
  p += q * d->L;
  o->lnf_w = p; p = al64(p + E);
  o->lnf_b = p; p = al64(p + E);
  o->hw1 = p; p = al64(p + E * E);
  o->hb1 = p; p = al64(p + E);
  o->hw2 = p; p = al64(p + (long long)d->Kc * E);
  o->hb2 = p; p = al64(p + d->Kc);
  o->total = p;
  return 0;
}

extern "C" int vg_gen_layout(const VgGenDims* d, VgGenLayout* o) {
  if (!d || !o) return -1;
  const long long E = d->E, T = d->T;
  if (d->E % 128 || d->E % d->H || (d->Z & 7) || (d->O & 7) || (d->CW & 7) || d->T > 80 || d->L < 1) return -3;
  const int HE = d->E / d->H;
  if (HE != 32 && HE != 64 && HE != 96) return -3;
  if (d->patch < 0) return -3;
  if (2 * d->L + 3 > VG_MAX_FOLD_JOBS) return -3;  // the backward queues 2L+1 deferred SLN folds + the two SIREN bias gradients
  if (d->patch > 0) {  // tokens on the patch grid: T and CW are determined by the image geometry
    if (d->C < 1 || d->IH < d->patch || d->IH % d->patch) return -3;
    const int gh = d->IH / d->patch;
    if (d->T != gh * gh || d->CW != d->C * d->patch * d->patch) return -3;
  }
  long long p = 0;
  o->emb = p; p = al64(p + T * E);
  o->map_w = p; p = al64(p + T * E * d->Z);
  o->map_b = p; p = al64(p + T * E);
  long long q = 0;
  o->wqkv = q; q += 3 * E * E;
  o->wo = q; q += E * E;
  o->wm = q; q += E * E;
  o->layer_weights = q;
  o->sln1_w = q; q = al64(q + E);
  o->sln1_b = q; q = al64(q + E);
  o->sln1_s = q; q = al64(q + 2);
  o->sln2_w = q; q = al64(q + E);
  o->sln2_b = q; q = al64(q + E);
  o->sln2_s = q; q = al64(q + 2);
  o->bo = q; q = al64(q + E);
  o->bm = q; q = al64(q + E);
  o->layer0 = p; o->layer_stride = q;
  p += q * d->L;
  o->slnf_w = p; p = al64(p + E);
  o->slnf_b = p; p = al64(p + E);
  o->slnf_s = p; p = al64(p + 2);
  o->s1_w = p; p = al64(p + (long long)d->O * E);
  o->s1_b = p; p = al64(p + d->O);
  o->s2_w = p; p = al64(p + (long long)d->CW * d->O);
  o->s2_b = p; p = al64(p + d->CW);
  o->total = p;
  return 0;
}

// =============================================================================================
//                                     small helpers
// =============================================================================================
struct Carver {
  unsigned char* base; long long off;
  template <typename T> T* take(long long n) {
    T* p = base ? (T*)(base + off) : nullptr;
    off += ((long long)n * (long long)sizeof(T) + 255) & ~255LL;
    return p;
  }
};

static VgGemmProb mk(const bf16* A, int lda, const bf16* Bm, int ldb, int M, int N, int K) {
  VgGemmProb p = vg_gemm_prob();
  p.A = A; p.lda = lda; p.B = Bm; p.ldb = ldb; p.M = M; p.N = N; p.K = K;
  return p;
}
struct Drop { unsigned thr; float scale; unsigned long long seed; const unsigned* step; };
static Drop mk_drop(float p, unsigned long long seed, const unsigned* step) {
  Drop d; int t = (int)lrintf(p * 256.f); if (t < 0) t = 0; if (t > 255) t = 255;
  d.thr = (unsigned)t; d.scale = t ? 256.f / (256.f - (float)t) : 1.f; d.seed = seed; d.step = step; return d;
}
static unsigned site_key(const Drop& d, int site) { return vg_site_key(d.seed, site); }
static void set_drop(VgGemmProb& p, const Drop& d, int site, int post) {
  if (!d.thr) return;
  p.drop_thresh = d.thr; p.drop_key = site_key(d, site); p.drop_scale = d.scale; p.drop_post = post; p.drop_step = d.step;
}
// forward Linear: C = act(A W^T + b) (+res)
static int lin_fwd(const bf16* A, int K, const bf16* W, const float* bias, bf16* C, int M, int N, int act, float ascale,
                   const bf16* res, bf16* pre_bf16, float* pre_f32, hipStream_t st, const Drop* drop = nullptr, int site = 0,
                   int c2_gelu_grad = 0) {
  VgGemmProb p = mk(A, K, W, K, M, N, K);
  p.c2_gelu_grad = c2_gelu_grad;
  if (drop) set_drop(p, *drop, site, 0);
  p.C = C; p.ldc = N; p.bias = bias; p.act = act; p.act_scale = ascale;
  p.res = res; p.ldr = N; p.C2 = pre_bf16; p.ldc2 = N;
  if (pre_f32) { p.pre_f32 = 1; p.Cf = pre_f32; p.ldcf = N; }
  return vg_gemm_launch(&p, 1, VG_NT, st);
}
// dgrad: dX[M,K] = dY[M,N] W[N,K]
static int lin_dgrad(const bf16* dY, const bf16* W, bf16* dX, int M, int N, int K, int mul, const bf16* Z, const float* Zf,
                     float ascale, hipStream_t st) {
  VgGemmProb p = mk(dY, N, W, K, M, K, N);  // GEMM (M x K_out=K) with reduction N
  p.C = dX; p.ldc = K; p.act = mul; p.act_scale = ascale; p.Z = Z; p.ldz = K; p.Zf = Zf; p.ldzf = K;
  return vg_gemm_launch(&p, 1, VG_NN, st);
}
// wgrad problem: dW[N,K] = dY[M,N]^T X[M,K] -> slab (fp32), k-dimension = M rows
static VgGemmProb wg(const bf16* dY, int N, const bf16* X, int K, int M, float* slab, long long split_stride, int splits) {
  VgGemmProb p = mk(dY, N, X, K, N, K, M);
  p.Cf = slab; p.ldcf = K; p.cf_split_stride = split_stride; p.splits = splits;
  return p;
}
static int pick_splits(long long tiles, int Krows, int cap) {
  const int ksteps = (Krows + 63) / 64;
  long long s = (640 + tiles - 1) / tiles;
  if (s > cap) s = cap;
  if (s > ksteps / 4) s = ksteps / 4;
  if (s < 1) s = 1;
  return (int)s;
}
static inline long long tiles128(long long m, long long n) { return ((m + 127) / 128) * ((n + 127) / 128); }
// Blocks whose widths are multiples of 384 run their weight gradients on 128 x 384 tiles, one 8-wave workgroup per CU
// (gemm_tn.hip): as many K slices as keep the grouped launch within one workgroup per CU.
// (tile width 384 when both widths are multiples of 384, else 512 when both are multiples of 512, else 0: tiled kernel)
static inline int wide_bn(int E, int hidden) {
  if (E % 384 == 0 && hidden % 384 == 0) return 384;
  if (E % 512 == 0 && hidden % 512 == 0) return 512;
  return 0;
}
static inline long long tiles_wide(long long m, long long n, int bn) { return ((m + 127) / 128) * (n / bn); }
static int pick_splits384(long long tiles, int Krows, int cap) {
  long long s = 256 / tiles;
  if (s > cap) s = cap;
  if (s > Krows / 256) s = Krows / 256;
  if (s < 1) s = 1;
  return (int)s;
}

// =============================================================================================
//                                   ViT (discriminator)
// =============================================================================================
#define VIT_SPLIT_CAP 16
#define EMB_SPLIT_CAP 32
struct VitWs {
  bf16 *Apatch, *X, *xn1, *qkv, *ao, *xmid, *xn2, *a1, *xcls, *hcls, *th;
  unsigned char* z1;  // gelu'(fc1 pre-activation), one byte per element (vg_common.h vg_g8_pack4)
  float *lse, *mean1, *rstd1, *mean2, *rstd2, *meanf, *rstdf;
  // backward scratch, one set per block parity (block l uses set l&1; its LN1 backward writes gin/gm2 of set (l-1)&1):
  // the weight-gradient side of block l still reads set l&1 while the main stream works on block l-1 in the other set
  struct Set { bf16 *gin, *gm2, *gmid, *gm1, *dz1, *dqkv; } set[2];
  float* lnpart;  // [2L] LayerNorm-backward partial-sum blocks, folded by one launch at the end of a backward call
  float* bslab;   // [L][VIT_SPLIT_CAP][3E + rE + E] bias-gradient rows written by the weight-gradient GEMM, one per K slice
  bf16 *dxn, *dao, *gp, *dA, *dzh, *dhcls, *dxcls;
  float *part, *part_cs, *hpart, *tok_sum, *slab;
  bf16* wpack;  // E = 384: stage images of Wo | W2 | Wqkv^T | W1^T per block for the full-row GEMMs (gemm_row.hip)
  // Pruned tail of the TOP block.  The classifier reads the CLS row only (src/v2/modules.py:195), so behind the top block's attention
  // every row-local operator - out-projection, residual, norm2, fc1, GELU, fc2, residual - matters for the B CLS rows alone, and in the
  // backward dL/dX[L] is exactly zero on the other 64/65 of the rows: those operators (and their weight gradients) run on compact
  // [B, .] tensors, forward and backward; the values and gradients the reference defines are unchanged.
  bf16 *t_xmid, *t_xn2, *t_a1, *t_xtop; unsigned char* t_z1; float *t_mean2, *t_rstd2;
  bf16 *t_gb2, *t_dz1, *t_dxn2, *t_dxmid, *t_gb1, *t_dao;
  bf16* t_ao; float* t_lse;  // and its attention for the CLS query only (dot-product scores; the fp8 mode keeps the full kernels)
};
// The full-row GEMMs (LayerNorm in the epilogue) take the block Linears whose output is the embedding when E = 384 and the
// rows come in whole units of 16; their workgroup count is also the number of LayerNorm-backward partial rows.
static inline int vit_row_nwg(const VgVitDims& d, int M) { return vg_row_width_ok(d.E) ? vg_row_nwg(M) : 0; }  // (E = 384, and 512 since round 4)
static long long carve_vit(const VgVitDims& d, int B, void* base, VitWs& w) {
  const long long E = d.E, NP = (long long)(d.IH / d.P) * (d.IH / d.P), S = NP + 1, M = (long long)B * S;
  const long long Kp = (long long)d.C * d.P * d.P, L = d.L, rE = (long long)d.R * E;
  VgVitLayout lay; vg_vit_layout(&d, &lay);
  Carver c{(unsigned char*)base, 0};
  w.Apatch = c.take<bf16>(B * NP * Kp);
  w.X = c.take<bf16>((L + 1) * M * E);
  w.xn1 = c.take<bf16>(L * M * E);
  w.qkv = c.take<bf16>(L * M * 3 * E);
  w.ao = c.take<bf16>(L * M * E);
  w.xmid = c.take<bf16>(L * M * E);
  w.xn2 = c.take<bf16>(L * M * E);
  w.z1 = c.take<unsigned char>(L * M * rE);
  w.a1 = c.take<bf16>(L * M * rE);
  w.xcls = c.take<bf16>(B * E); w.hcls = c.take<bf16>(B * E); w.th = c.take<bf16>(B * E);
  w.lse = c.take<float>(L * (long long)B * d.H * S);
  w.mean1 = c.take<float>(L * M); w.rstd1 = c.take<float>(L * M);
  w.mean2 = c.take<float>(L * M); w.rstd2 = c.take<float>(L * M);
  w.meanf = c.take<float>(B); w.rstdf = c.take<float>(B);
  for (int i = 0; i < 2; ++i) {
    VitWs::Set& t = w.set[i];
    t.gin = c.take<bf16>(M * E); t.gm2 = c.take<bf16>(M * E);   // dL/dX entering the block, and its dropout-masked copy
    t.gmid = c.take<bf16>(M * E); t.gm1 = c.take<bf16>(M * E);  // same after the MLP half of the block
    t.dz1 = c.take<bf16>(M * rE);
    t.dqkv = c.take<bf16>(M * 3 * E);
  }
  w.lnpart = c.take<float>(2 * L * (long long)vg_ln_bwd_nparts((int)M) * 3 * E);
  w.bslab = c.take<float>(L * (long long)VIT_SPLIT_CAP * (3 * E + (long long)d.R * E + E));
  w.dxn = c.take<bf16>(M * E);
  w.dao = c.take<bf16>(M * E);
  w.gp = c.take<bf16>(B * NP * E);
  w.dA = c.take<bf16>(B * NP * Kp);
  w.dzh = c.take<bf16>(B * E); w.dhcls = c.take<bf16>(B * E); w.dxcls = c.take<bf16>(B * E);
  w.part = c.take<float>((long long)vg_ln_bwd_nparts((int)M) * 3 * E);
  w.part_cs = c.take<float>((long long)vg_colsum_bf16_nparts((int)M) * 3 * E);
  w.hpart = c.take<float>(d.Kc <= 16 ? (long long)vg_head_bwd_parts((int)B) * vg_head_bwd_part_width((int)E, d.Kc) : 0);  // classifier head: partial gradient rows
  w.tok_sum = c.take<float>(S * E);
  long long slab = VIT_SPLIT_CAP * lay.layer_weights;
  if (EMB_SPLIT_CAP * E * Kp > slab) slab = EMB_SPLIT_CAP * E * Kp;
  w.slab = c.take<float>(slab);
  w.wpack = c.take<bf16>(vit_row_nwg(d, (int)M) ? L * lay.layer_weights : 0);
  w.t_xmid = c.take<bf16>(B * E); w.t_xn2 = c.take<bf16>(B * E); w.t_a1 = c.take<bf16>(B * rE); w.t_xtop = c.take<bf16>(B * E);
  w.t_z1 = c.take<unsigned char>(B * rE); w.t_mean2 = c.take<float>(B); w.t_rstd2 = c.take<float>(B);
  w.t_gb2 = c.take<bf16>(B * E); w.t_dz1 = c.take<bf16>(B * rE); w.t_dxn2 = c.take<bf16>(B * E); w.t_dxmid = c.take<bf16>(B * E);
  w.t_gb1 = c.take<bf16>(B * E); w.t_dao = c.take<bf16>(B * E);
  w.t_ao = c.take<bf16>(B * E); w.t_lse = c.take<float>((long long)B * d.H);
  return c.off;
}
extern "C" long long vg_vit_ws_bytes(const VgVitDims* d, int B) {
  VgVitLayout lay;
  if (!d || B < 1 || vg_vit_layout(d, &lay)) return -1;
  VitWs w;
  return carve_vit(*d, B, nullptr, w);
}

// Byte offsets of the saved activations / gradient scratch inside the workspace (introspection for the parity tests:
// they teacher-force each encoder block with the tensors the kernels really produced).
extern "C" int vg_vit_ws_map(const VgVitDims* d, int B, VgVitWsMap* o) {
  VgVitLayout lay;
  if (!d || !o || B < 1 || vg_vit_layout(d, &lay)) return -1;
  unsigned char* const fake = (unsigned char*)(uintptr_t)(1u << 20);  // never dereferenced
  VitWs w;
  o->total = carve_vit(*d, B, fake, w);
  auto off = [&](const void* p) { return (long long)((const unsigned char*)p - fake); };
  o->X = off(w.X); o->xn1 = off(w.xn1); o->qkv = off(w.qkv); o->ao = off(w.ao); o->xmid = off(w.xmid); o->xn2 = off(w.xn2);
  o->z1 = off(w.z1); o->a1 = off(w.a1); o->lse = off(w.lse);
  o->mean1 = off(w.mean1); o->rstd1 = off(w.rstd1); o->mean2 = off(w.mean2); o->rstd2 = off(w.rstd2);
  for (int i = 0; i < 2; ++i) { o->gin[i] = off(w.set[i].gin); o->gmid[i] = off(w.set[i].gmid); o->dqkv[i] = off(w.set[i].dqkv); o->dz1[i] = off(w.set[i].dz1); }
  o->xtop = off(w.t_xtop); o->dxtop = off(w.dxcls);
  return 0;
}

// preact (nullable): [L][M][rE] - the gradient penalty's forward keeps fc1's pre-activation (its double backward needs gelu''), not the
// one-byte gelu' code; only with dense_top (the penalty runs every row of the top block)
static int vit_forward_impl(const VgVitNet* net, int B, const void* img, int img_is_bf16, void* ws, float* logits, void* stream, bf16* preact) {
  if (!net || !img || !ws || !logits || B < 1) return -1;
  if (preact && !net->dense_top) return -3;
  const VgVitDims& d = net->d;
  VgVitLayout lay;
  VG_TRY(vg_vit_layout(&d, &lay));
  hipStream_t st = (hipStream_t)stream;
  const int E = d.E, NP = (d.IH / d.P) * (d.IH / d.P), S = NP + 1, M = B * S, Kp = d.C * d.P * d.P, rE = d.R * E;
  const int HE = E / d.H;
  VitWs w; carve_vit(d, B, ws, w);
  const float* P = net->P; const bf16* Pb = (const bf16*)net->Pb;
  const size_t ME = (size_t)M * E;
  const Drop dr = mk_drop(net->dropout_p, net->dropout_seed, net->dropout_step);  // sites: 0 embedding, 1+2l attention branch, 2+2l MLP branch

  // patch embedding (src/v2/modules.py:82-98): gather -> GEMM(+bias +pos, rows remapped past CLS) ; CLS row
  VG_TRY(vg_patchify_launch(img, img_is_bf16, w.Apatch, B, d.C, d.IH, d.P, st));
  {
    VgGemmProb p = mk(w.Apatch, Kp, Pb + lay.conv_w, Kp, B * NP, E, Kp);
    p.C = w.X; p.ldc = E; p.bias = P + lay.conv_b; p.resf = P + lay.pos; p.res_period = NP;
    p.row_in_per = NP; p.row_out_per = S; p.row_out_off = 1;
    set_drop(p, dr, 0, 1);
    VG_TRY(vg_gemm_launch(&p, 1, VG_NT, st));
  }
  VG_TRY(vg_fill_cls_launch(w.X, P + lay.cls, B, S, E, dr.thr, site_key(dr, 0), dr.scale, dr.step, st));

  // full-row path: pack this call's weights (the backward of this workspace reads the transposed images)
  const int rown = vit_row_nwg(d, M);
  const long long po_wo = 0, po_w2 = po_wo + (long long)E * E, po_wqkvT = po_w2 + (long long)E * rE, po_w1T = po_wqkvT + 3LL * E * E;
  if (rown) {
    VgPackJobs pj;
    pj.N = E;
    pj.src = Pb + lay.layer0; pj.dst = w.wpack; pj.src_stride = lay.layer_stride; pj.dst_stride = lay.layer_weights; pj.nblocks = d.L; pj.n = 4;
    pj.d[0] = {lay.wo, po_wo, E, E, 0};          // out-projection forward: W [E, E], contraction E
    pj.d[1] = {lay.w2, po_w2, rE, rE, 0};        // fc2 forward: W [E, rE], contraction rE
    pj.d[2] = {lay.wqkv, po_wqkvT, 3 * E, E, 1}; // QKV input gradient: W [3E, E] read transposed, contraction 3E
    pj.d[3] = {lay.w1, po_w1T, rE, E, 1};        // fc1 input gradient: W [rE, E] read transposed, contraction rE
    VG_TRY(vg_pack_rows_launch(pj, st));
  }
  auto row_fwd = [&](const bf16* A, int K, const bf16* Wp, const float* bias, const bf16* res, bf16* Y, bf16* Yn, float* mean,
                     float* rstd, const float* gamma, const float* beta, int site, int rows = 0, int drm = 1, long long ldr = 0) -> int {
    VgRowArgs ra = {};
    ra.N = E;
    ra.A = A; ra.lda = K; ra.Wp = Wp; ra.M = rows ? rows : M; ra.K = K; ra.bias = bias; ra.res = res; ra.ldr = ldr; ra.Y = Y; ra.Yn = Yn;
    ra.mean_out = mean; ra.rstd_out = rstd; ra.gamma = gamma; ra.beta = beta; ra.eps = 1e-5f; ra.drop_row_mul = drm;
    if (dr.thr) { ra.drop_thresh = dr.thr; ra.drop_key = site_key(dr, site); ra.drop_scale = dr.scale; ra.drop_step = dr.step; }
    const int r = vg_gemm_row_launch(ra, VG_ROW_LNFWD, st);
    return r > 0 ? 0 : (r < 0 ? -r : -3);
  };

  bool tail_norm_done = false;  // the final LayerNorm already sits in the epilogue of the top block's fc2 (full-row tail)
  for (int l = 0; l < d.L; ++l) {
    const long long lo = lay.layer0 + (long long)l * lay.layer_stride;
    const bf16* x = w.X + (size_t)l * ME;
    bf16* xn1 = w.xn1 + (size_t)l * ME;
    bf16* qkv = w.qkv + (size_t)l * ME * 3;
    bf16* ao = w.ao + (size_t)l * ME;
    bf16* xmid = w.xmid + (size_t)l * ME;
    bf16* xn2 = w.xn2 + (size_t)l * ME;
    unsigned char* z1 = w.z1 + (size_t)l * M * rE;
    bf16* a1 = w.a1 + (size_t)l * M * rE;
    const bf16* wp = w.wpack + (size_t)l * lay.layer_weights;
    // norm1: standalone for block 0 (and on the tiled path); on the full-row path the fc2 epilogue of block l-1 wrote it
    if (!rown || l == 0)
      VG_TRY(vg_ln_fwd_launch(x, E, P + lo + lay.ln1_w, P + lo + lay.ln1_b, xn1, E, w.mean1 + (size_t)l * M,
                              w.rstd1 + (size_t)l * M, M, E, 1e-5f, st));
    VG_TRY(lin_fwd(xn1, E, Pb + lo + lay.wqkv, P + lo + lay.bqkv, qkv, M, 3 * E, VG_ACT_NONE, 0.f, nullptr, nullptr, nullptr, st));
    const bool tail = (l == d.L - 1) && !net->dense_top;       // top block: behind its attention only the CLS rows matter
    const bool cls_attn = tail && !net->attn_fp8;              // ... and the CLS query is the only one the classifier sees
    if (cls_attn) VG_TRY(vg_attn_cls_fwd_launch(qkv, w.t_ao, w.t_lse, B, d.H, S, HE, 1.0f / sqrtf((float)HE), st));
    else VG_TRY(vg_attn_fwd_launch(qkv, ao, w.lse + (size_t)l * B * d.H * S, B, d.H, S, HE, 1.0f / sqrtf((float)HE), net->attn_fp8 ? 2 : 0, st));
    if (tail) {
      // Top block: only its CLS rows reach the classifier, so everything behind the attention runs on those B rows (compact tensors;
      // A = rows b S of `ao`, residual = rows b S of x by their leading dimension; dropout bits = those of rows b S of the full tensor)
      if (rown && cls_attn && vg_row_nwg(B)) {
        // the same full-row kernels as the blocks below, M = B: out-projection + residual + norm2, then fc2 + residual + the FINAL
        // LayerNorm (its rows are exactly the CLS rows) - the tiled kernel covers so small a problem with 6-12 workgroups whose 12-24
        // k-steps each wait out a full memory latency (22 us for the fc2 launch), the full-row kernel keeps two stages in flight
        VG_TRY(row_fwd(w.t_ao, E, wp + po_wo, P + lo + lay.bo, x, w.t_xmid, w.t_xn2, w.t_mean2, w.t_rstd2, P + lo + lay.ln2_w, P + lo + lay.ln2_b,
                       1 + 2 * l, B, S, (long long)S * E));  // residual = rows b S of x through its row stride
        VG_TRY(lin_fwd(w.t_xn2, E, Pb + lo + lay.w1, P + lo + lay.b1, w.t_a1, B, rE, VG_ACT_GELU, 0.f, nullptr, (bf16*)w.t_z1, nullptr, st, nullptr, 0, 2));
        VG_TRY(row_fwd(w.t_a1, rE, wp + po_w2, P + lo + lay.b2, w.t_xmid, w.t_xtop, w.hcls, w.meanf, w.rstdf, P + lay.lnf_w, P + lay.lnf_b, 2 + 2 * l, B, S));
        tail_norm_done = true;
        continue;
      }
      VgGemmProb po = cls_attn ? mk(w.t_ao, E, Pb + lo + lay.wo, E, B, E, E) : mk(ao, S * E, Pb + lo + lay.wo, E, B, E, E);
      po.C = w.t_xmid; po.ldc = E; po.bias = P + lo + lay.bo; po.res = x; po.ldr = S * E;
      set_drop(po, dr, 1 + 2 * l, 0); po.drop_row_mul = S;
      VG_TRY(vg_gemm_launch(&po, 1, VG_NT, st));
      VG_TRY(vg_ln_fwd_launch(w.t_xmid, E, P + lo + lay.ln2_w, P + lo + lay.ln2_b, w.t_xn2, E, w.t_mean2, w.t_rstd2, B, E, 1e-5f, st));
      VG_TRY(lin_fwd(w.t_xn2, E, Pb + lo + lay.w1, P + lo + lay.b1, w.t_a1, B, rE, VG_ACT_GELU, 0.f, nullptr, (bf16*)w.t_z1, nullptr, st, nullptr, 0, 2));
      VgGemmProb p2 = mk(w.t_a1, rE, Pb + lo + lay.w2, rE, B, E, rE);
      p2.C = w.t_xtop; p2.ldc = E; p2.bias = P + lo + lay.b2; p2.res = w.t_xmid; p2.ldr = E;
      set_drop(p2, dr, 2 + 2 * l, 0); p2.drop_row_mul = S;
      VG_TRY(vg_gemm_launch(&p2, 1, VG_NT, st));
      continue;
    }
    if (rown) {  // x_mid = x + drop(out_projection(ao)) and norm2(x_mid) in one kernel
      VG_TRY(row_fwd(ao, E, wp + po_wo, P + lo + lay.bo, x, xmid, xn2, w.mean2 + (size_t)l * M, w.rstd2 + (size_t)l * M,
                     P + lo + lay.ln2_w, P + lo + lay.ln2_b, 1 + 2 * l));
    } else {
      VG_TRY(lin_fwd(ao, E, Pb + lo + lay.wo, P + lo + lay.bo, xmid, M, E, VG_ACT_NONE, 0.f, x, nullptr, nullptr, st, &dr, 1 + 2 * l));
      VG_TRY(vg_ln_fwd_launch(xmid, E, P + lo + lay.ln2_w, P + lo + lay.ln2_b, xn2, E, w.mean2 + (size_t)l * M,
                              w.rstd2 + (size_t)l * M, M, E, 1e-5f, st));
    }
    // z1 keeps gelu'(pre-activation), the only thing the backward needs of it, as one byte per element
    if (preact) VG_TRY(lin_fwd(xn2, E, Pb + lo + lay.w1, P + lo + lay.b1, a1, M, rE, VG_ACT_GELU, 0.f, nullptr, preact + (size_t)l * M * rE, nullptr, st));
    else VG_TRY(lin_fwd(xn2, E, Pb + lo + lay.w1, P + lo + lay.b1, a1, M, rE, VG_ACT_GELU, 0.f, nullptr, (bf16*)z1, nullptr, st, nullptr, 0, 2));
    if (rown) {  // X[l+1] = x_mid + drop(fc2(a1)) and the NEXT block's norm1 of it (the last block's output only feeds the CLS rows)
      const bool nx = l + 1 < d.L;
      const long long ln = lo + lay.layer_stride;
      VG_TRY(row_fwd(a1, rE, wp + po_w2, P + lo + lay.b2, xmid, w.X + (size_t)(l + 1) * ME, nx ? w.xn1 + (size_t)(l + 1) * ME : nullptr,
                     nx ? w.mean1 + (size_t)(l + 1) * M : nullptr, nx ? w.rstd1 + (size_t)(l + 1) * M : nullptr,
                     nx ? P + ln + lay.ln1_w : nullptr, nx ? P + ln + lay.ln1_b : nullptr, 2 + 2 * l));
    } else {
      VG_TRY(lin_fwd(a1, rE, Pb + lo + lay.w2, P + lo + lay.b2, w.X + (size_t)(l + 1) * ME, M, E, VG_ACT_NONE, 0.f, xmid,
                     nullptr, nullptr, st, &dr, 2 + 2 * l));
    }
  }
  // final LayerNorm acts on every row in the reference (:236) but only the CLS row feeds the
  // classifier (:195): normalise the B CLS rows only.
  // final LayerNorm: acts on every row in the reference (:236), only the CLS rows feed the classifier (:195)
  if (net->dense_top) VG_TRY(vg_ln_fwd_launch(w.X + (size_t)d.L * ME, (long long)S * E, P + lay.lnf_w, P + lay.lnf_b, w.hcls, E, w.meanf, w.rstdf, B, E, 1e-5f, st));
  else if (!tail_norm_done) VG_TRY(vg_ln_fwd_launch(w.t_xtop, E, P + lay.lnf_w, P + lay.lnf_b, w.hcls, E, w.meanf, w.rstdf, B, E, 1e-5f, st));
  VG_TRY(lin_fwd(w.hcls, E, Pb + lay.hw1, P + lay.hb1, w.th, B, E, VG_ACT_TANH, 0.f, nullptr, nullptr, nullptr, st));
  VG_TRY(vg_head_fc2_launch(w.th, P + lay.hw2, P + lay.hb2, logits, B, E, d.Kc, st));
  return 0;
}
extern "C" int vg_vit_forward(const VgVitNet* net, int B, const void* img, int img_is_bf16, void* ws, float* logits,
                              void* stream) {
  return vit_forward_impl(net, B, img, img_is_bf16, ws, logits, stream, nullptr);
}

// Backward stages: 0 = classifier head + final LN, 1..L = encoder blocks L-1 .. 0, L+1 = patch embedding.
// Running [stage_begin, stage_end) lets the caller all-reduce the gradients of finished blocks (a contiguous
// range of the flat buffer) on another stream while the remaining stages still compute.
// The gradient penalty's SECOND backward (vg_vit_penalty below) is this backward with gradients injected at the activations the
// double backward reaches: dL/d(fc1 pre-activation), dL/d(qkv), dL/d(x_mid), dL/d(X[l]) per block, dL/dX[L] on the CLS rows, and
// dL/d(classifier fc1 pre-activation) in place of the logits' gradient.
struct VitInject {
  const bf16* preact;   // [L][M][rE]: the forward kept fc1's pre-activation (vit_forward_impl) - gelu' is computed from it
  int head_given;       // w.dzh already holds dL/d(classifier fc1 pre-activation): no logits' backward, no fc2 gradients
  const bf16* s_xcls;   // [B, E]
  const bf16 *s_h, *s_xmid, *s_qkv, *s_x;  // block l at + l M {rE, E, 3E, E}
  bf16* tmp;            // [M, E]: gres + s where the residual-stream gradient is an operand of other launches too
};
static int vit_backward_impl(const VgVitNet* net, int B, void* ws, const float* dlogits, void* d_img, int want_wgrad,
                             int stage_begin, int stage_end, void* stream, const VitInject* inj) {
  if (!net || !ws || !dlogits || B < 1) return -1;
  if (inj && !net->dense_top) return -3;
  if (stage_begin < 0 || stage_end > net->d.L + 2 || stage_begin >= stage_end) return -2;
  if (want_wgrad && !net->G) return -1;
  const VgVitDims& d = net->d;
  VgVitLayout lay;
  VG_TRY(vg_vit_layout(&d, &lay));
  hipStream_t st = (hipStream_t)stream;
  const int E = d.E, NP = (d.IH / d.P) * (d.IH / d.P), S = NP + 1, M = B * S, Kp = d.C * d.P * d.P, rE = d.R * E;
  const int HE = E / d.H;
  VitWs w; carve_vit(d, B, ws, w);
  const float* P = net->P; const bf16* Pb = (const bf16*)net->Pb; float* G = net->G;
  const size_t ME = (size_t)M * E;
  const int rown = vit_row_nwg(d, M);  // full-row path: the forward packed the weights into this workspace
  const int lnparts = rown ? rown : vg_ln_bwd_nparts(M);
  const long long po_wqkvT = (long long)E * E + (long long)E * rE, po_w1T = po_wqkvT + 3LL * E * E;
  const Drop dr = mk_drop(net->dropout_p, net->dropout_seed, net->dropout_step);
  const bool drop = dr.thr != 0;
  // dx = gres + LayerNorm'(A W) in one kernel (gemm_row.hip)
  auto row_bwd = [&](const bf16* A, int K, const bf16* Wp, const bf16* x, const float* mean, const float* rstd, const float* gamma,
                     const bf16* gres, bf16* dx, bf16* dxm, float* part, int site, int rows = 0, int drm = 1, const bf16* gres2 = nullptr) -> int {
    VgRowArgs ra = {};
    ra.N = E;
    ra.A = A; ra.lda = K; ra.Wp = Wp; ra.M = rows ? rows : M; ra.K = K; ra.x = x; ra.mean = mean; ra.rstd = rstd; ra.gamma = gamma;
    ra.gres = gres; ra.dx = dx; ra.dxm = dxm; ra.part = want_wgrad ? part : nullptr; ra.drop_row_mul = drm;  // (no parameter gradients wanted: no column sums)
    ra.gres2 = gres2;  // the penalty's second backward: the gradient its double backward injected at this LayerNorm's input
    if (dxm) { ra.drop_thresh = dr.thr; ra.drop_key = site_key(dr, site); ra.drop_scale = dr.scale; ra.drop_step = dr.step; }
    const int r = vg_gemm_row_launch(ra, gres2 ? VG_ROW_LNBWD_PEN : VG_ROW_LNBWD, st);
    return r > 0 ? 0 : (r < 0 ? -r : -3);
  };
  VgCtx* ctx = (VgCtx*)net->ctx;
  hipStream_t sd = ctx ? ctx->side : st;  // stream of the weight-gradient side work
  const int top = d.L - 1;
  const bool tail = !net->dense_top;                                          // the top block's pruned tail (as the forward chose)
  const bool tail_row = tail && rown && !net->attn_fp8 && vg_row_nwg(B) > 0;  // ... on the full-row kernels
  VgFoldJobs folds; folds.n = 0;  // partial-sum folds queued by this call: one launch at its end
  if (stage_begin == 0) {
  // ---- classifier head + final LN (CLS rows only) ----
  // dz, and the gradients of fc2 and of fc1's bias as partial rows for the fold at the end of this call: one launch
  const int head1 = (inj && inj->head_given) ? 0 :
                    vg_head_bwd_launch(dlogits, P + lay.hw2, w.th, w.dzh, want_wgrad ? G + lay.hw2 : nullptr, want_wgrad ? G + lay.hb2 : nullptr,
                                       B, E, d.Kc, want_wgrad, st, (want_wgrad && d.Kc <= 16) ? w.hpart : nullptr);
  if (head1 < 0) return -head1;
  if (want_wgrad) {
    if (head1) VG_TRY(vg_fold_push(folds, w.hpart, vg_head_bwd_parts(B), vg_head_bwd_part_width(E, d.Kc), G + lay.hw2, d.Kc * E, G + lay.hb1, E,
                                   G + lay.hb2, d.Kc, nullptr, 0));
    else VG_TRY(vg_colsum_bf16_launch(w.dzh, E, B, E, w.part_cs, G + lay.hb1, 1, st));
    // (a few K slices: with one, the three 128 x 384 tiles of this 384 x 384 x B problem are three workgroups walking 16 stages - 18 us at B = 512)
    VgGemmProb p = wg(w.dzh, E, w.hcls, E, B, w.slab, (long long)E * E, B >= 512 ? 4 : (B >= 256 ? 2 : 1));
    VG_TRY(vg_gemm_launch(&p, 1, VG_TN, st));
    VG_TRY(vg_slab_reduce_launch(w.slab, (long long)E * E, p.splits, G + lay.hw1, (long long)E * E, 1, st));
  }
  VG_TRY(lin_dgrad(w.dzh, Pb + lay.hw1, w.dhcls, B, E, E, 0, nullptr, nullptr, 0.f, st));
  if (tail) {
    // dL/dX[L] on the CLS rows (it is zero elsewhere) and its masked copy for the top block's MLP dropout - the bits of rows b S of the full tensor
    VG_TRY(vg_ln_bwd_launch(w.dhcls, w.t_xtop, w.meanf, w.rstdf, P + lay.lnf_w, nullptr, w.dxcls, w.part, B, E, drop ? w.t_gb2 : nullptr, dr.thr,
                            site_key(dr, 2 + 2 * top), dr.scale, dr.step, st, 1, S));
  } else {
    VG_TRY(vg_ln_bwd_launch(w.dhcls, w.X + (size_t)d.L * ME, w.meanf, w.rstdf, P + lay.lnf_w, nullptr, w.dxcls, w.part, B, E, nullptr, 0, 0, 1.f, nullptr, st, S));
    if (inj && inj->s_xcls) VG_TRY(vg_add_bf16_launch(w.dxcls, inj->s_xcls, w.dxcls, (long long)B * E, st));
    // dL/dX[L]: the CLS rows, zero elsewhere - and its masked copy for the last block's MLP dropout, in the same launch
    VG_TRY(vg_scatter_cls_launch(w.dxcls, w.set[top & 1].gin, B, S, E, st, drop ? w.set[top & 1].gm2 : nullptr, dr.thr, site_key(dr, 2 + 2 * top), dr.scale, dr.step));
  }
  if (want_wgrad)  // (the final LayerNorm's own partial count: B rows, standalone kernel; w.part is nobody else's)
    VG_TRY(vg_fold_push(folds, w.part, vg_ln_bwd_nparts(B), 3 * E, G + lay.lnf_w, E, G + lay.lnf_b, E, nullptr, E, nullptr, 0));
  }

  int last_side = -1;  // highest-index side event recorded by this call (for the join)
  const size_t part_sz = (size_t)lnparts * 3 * E;
  // Weight gradients (single-stream schedule): the blocks of this call are taken in PAIRS - the eight problems of two blocks
  // as ONE grouped split-K launch with half the K slices (same number of workgroups: 42 tiles x 6 instead of 21 x 12), which
  // halves the fp32 slab traffic (85 -> 42 MB written and folded per block) and the prologues / epilogues per unit of work.
  // The launch sits in the SECOND block of the pair, in front of its last kernel (QKV input gradient + norm1 backward): that
  // kernel writes dL/dX into the other scratch set, where the first block's fc2-gradient operand still lives.
  const int l_hi = d.L - (stage_begin > 1 ? stage_begin : 1), l_lo = d.L - ((stage_end < d.L + 1 ? stage_end : d.L + 1) - 1);
  const bool pairing = !ctx && want_wgrad;
  const long long BW = 3 * E + rE + E;
  auto wgrad_blocks = [&](int la, int nb) -> int {  // blocks la, la-1, .. (nb = 1 or 2): grouped launch + slab folds + bias partials
    long long tiles = tiles128(3 * E, E) + tiles128(E, E) + tiles128(rE, E) + tiles128(E, rE);
    // the slab and bslab carves hold VIT_SPLIT_CAP slices in all: never more (round 1 overran them from an environment knob)
    // The K partition is that of a PAIR also for a block that goes alone (the odd one out, the side-stream schedule): every
    // schedule then adds the same slices in the same order, and staged, one-shot and side-stream backward agree bit for bit.
    int splits = pick_splits(tiles * 2, M, VIT_SPLIT_CAP / 2);
    if (const int bn = wide_bn(E, rE); bn && M % 32 == 0 && E % 128 == 0 && rE % 128 == 0)
      splits = pick_splits384(2 * (tiles_wide(3 * E, E, bn) + tiles_wide(E, E, bn) + tiles_wide(rE, E, bn) + tiles_wide(E, rE, bn)), M, VIT_SPLIT_CAP / 2);
#ifdef VG_TUNING  // experimental builds only (make var): the product library reads no environment
    static const int split_env = getenv("VG_VIT_SPLITS") ? atoi(getenv("VG_VIT_SPLITS")) : 0;
    if (split_env > 0 && split_env <= VIT_SPLIT_CAP / 2) splits = split_env;
#endif
    // K slices per BLOCK, the same in every schedule (so every schedule adds the same slices in the same order).  With the pruned tail the
    // top block contributes its QKV problem only: the launch that holds it and the block below is 33 tiles instead of 48, and at 5 slices
    // 165 workgroups on 256 CUs - those two blocks therefore take 7 slices (231 workgroups; 149 -> ~110 us), wherever they are launched.
    int splits_top = splits;
    if (tail && top >= 1) {
      if (const int bn = wide_bn(E, rE); bn && M % 32 == 0 && E % 128 == 0 && rE % 128 == 0)
        splits_top = pick_splits384(tiles_wide(3 * E, E, bn) + (tiles_wide(3 * E, E, bn) + tiles_wide(E, E, bn) + tiles_wide(rE, E, bn) + tiles_wide(E, rE, bn)), M,
                                    VIT_SPLIT_CAP / 2);
      else
        splits_top = pick_splits(tiles128(3 * E, E) + tiles, M, VIT_SPLIT_CAP / 2);
    }
    auto splits_of = [&](int lb) { return (tail && top >= 1 && (lb == top || lb == top - 1)) ? splits_top : splits; };
    VgGemmProb pr[8];
    int np = 0, first[2] = {0, 0};
    long long slab_off[2] = {0, 0};
    for (int j = 0; j < nb; ++j) {
      const int lb = la - j;
      const int sp = splits_of(lb);
      if (j + 1 < nb) slab_off[j + 1] = slab_off[j] + (long long)sp * lay.layer_weights;
      VitWs::Set& sb = w.set[lb & 1];
      float* slab = w.slab + slab_off[j];
      const bf16* gb1b = drop ? sb.gm1 : sb.gmid;
      const bf16* gb2b = drop ? sb.gm2 : sb.gin;
      // bias gradients = column sums of the same dY operands: they ride along in the GEMM (ones x A on the MFMA pipe),
      // one row per K slice, folded with the LayerNorm partials at the end.  fc2's bias: only the top block needs it
      // here (lower blocks get it from the LN1 partials of the block above).
      float* bs = w.bslab + (size_t)lb * VIT_SPLIT_CAP * BW;
      VgGemmProb* q = pr + np;
      first[j] = np;
      q[0] = wg(sb.dqkv, 3 * E, w.xn1 + (size_t)lb * ME, E, M, slab + lay.wqkv, lay.layer_weights, sp);
      q[0].colsum = bs; q[0].colsum_split_stride = BW;
      if (tail && lb == top) { np += 1; continue; }  // top block: the other three are sums over its B CLS rows (below)
      q[1] = wg(gb1b, E, w.ao + (size_t)lb * ME, E, M, slab + lay.wo, lay.layer_weights, sp);
      q[2] = wg(sb.dz1, rE, w.xn2 + (size_t)lb * ME, E, M, slab + lay.w1, lay.layer_weights, sp);
      q[3] = wg(gb2b, E, w.a1 + (size_t)lb * M * rE, rE, M, slab + lay.w2, lay.layer_weights, sp);
      q[2].colsum = bs + 3 * E; q[2].colsum_split_stride = BW;
      if (lb == top) { q[3].colsum = bs + 3 * E + rE; q[3].colsum_split_stride = BW; }  // (dense top block: fc2's bias rides along here)
      np += 4;
    }
    VG_TRY(vg_gemm_launch(pr, np, VG_TN, sd));
    const bool ptop = tail && la == top;  // this launch holds the pruned top block: its slab has the QKV part only
    if (nb == 2 && !ptop && pr[first[0]].splits == pr[first[1]].splits) {  // both blocks' K slices in one launch
      const long long lo0 = lay.layer0 + (long long)la * lay.layer_stride, lo1 = lo0 - lay.layer_stride;
      VG_TRY(vg_slab_reduce2_launch(w.slab, w.slab + slab_off[1], lay.layer_weights, pr[first[0]].splits, G + lo0, G + lo1,
                                    lay.layer_weights, 1, sd));
    }
    for (int j = 0; j < nb; ++j) {
      const int lb = la - j;
      const long long lob = lay.layer0 + (long long)lb * lay.layer_stride;
      float* bs = w.bslab + (size_t)lb * VIT_SPLIT_CAP * BW;
      const int ns = pr[first[j]].splits;  // (the launcher drops empty slices; every problem of a block has the same M rows)
      const bool pt = tail && lb == top;
      if (nb == 1 || ptop || pr[first[0]].splits != pr[first[nb - 1]].splits)  // (the pruned top block's slab holds its QKV part only: wqkv is the first region of a layer)
        VG_TRY(vg_slab_reduce_launch(w.slab + slab_off[j], lay.layer_weights, ns, G + lob, pt ? 3LL * E * E : lay.layer_weights, 1, sd));
      VG_TRY(vg_fold_push(folds, bs, ns, (int)BW, G + lob + lay.bqkv, 3 * E, pt ? nullptr : G + lob + lay.b1, rE, (lb == top && !pt) ? G + lob + lay.b2 : nullptr, E,
                          nullptr, 0));
      if (!pt) continue;
      // ---- top block: out-projection / fc1 / fc2 weight gradients as sums over the B CLS rows (every other row of their dY is exactly
      // zero), ONE K slice accumulated straight into the gradient buffer; b1 / b2 ride along as one partial row ----
      const bf16* gb1c = drop ? w.t_gb1 : w.t_dxmid;
      const bf16* gb2c = drop ? w.t_gb2 : w.dxcls;
      float* bsc = bs + (size_t)(VIT_SPLIT_CAP - 1) * BW;  // the last row of the block's carve: a block never has more than VIT_SPLIT_CAP / 2 slices
      VgGemmProb c[3];
      if (!net->attn_fp8) c[0] = wg(gb1c, E, w.t_ao, E, B, G + lob + lay.wo, 0, 1);  // the CLS query's attention output
      else { c[0] = wg(gb1c, E, w.ao + (size_t)lb * ME, E, B, G + lob + lay.wo, 0, 1); c[0].ldb = S * E; }  // rows b S of the full one
      c[1] = wg(w.t_dz1, rE, w.t_xn2, E, B, G + lob + lay.w1, 0, 1);
      c[2] = wg(gb2c, E, w.t_a1, rE, B, G + lob + lay.w2, 0, 1);
      for (int i = 0; i < 3; ++i) c[i].cf_accumulate = 1;
      c[1].colsum = bsc + 3 * E; c[1].colsum_split_stride = BW;
      c[2].colsum = bsc + 3 * E + rE; c[2].colsum_split_stride = BW;
      VG_TRY(vg_gemm_launch(c, 3, VG_TN, sd));
      VG_TRY(vg_fold_push(folds, bsc, 1, (int)BW, nullptr, 3 * E, G + lob + lay.b1, rE, G + lob + lay.b2, E, nullptr, 0));
    }
    return 0;
  };
  for (int l = d.L - 1; l >= 0; --l) {
    const int stage = d.L - l;
    if (stage < stage_begin) continue;
    if (stage >= stage_end) break;
    const long long lo = lay.layer0 + (long long)l * lay.layer_stride;
    const bf16* x = w.X + (size_t)l * ME;
    const bf16* xn1 = w.xn1 + (size_t)l * ME;
    const bf16* qkv = w.qkv + (size_t)l * ME * 3;
    const bf16* ao = w.ao + (size_t)l * ME;
    const bf16* xmid = w.xmid + (size_t)l * ME;
    const bf16* xn2 = w.xn2 + (size_t)l * ME;
    const unsigned char* z1 = w.z1 + (size_t)l * M * rE;
    const bf16* a1 = w.a1 + (size_t)l * M * rE;
    VitWs::Set& cur = w.set[l & 1];
    VitWs::Set& nxt = w.set[(l & 1) ^ 1];  // receives dL/dX[l] for block l-1
    float* part2 = w.lnpart + (size_t)(2 * l) * part_sz;
    float* part1 = w.lnpart + (size_t)(2 * l + 1) * part_sz;
    const bf16* g = cur.gin;
    const bf16* gb2 = drop ? cur.gm2 : cur.gin;   // gradient w.r.t. the fc2 output (before dropout2)
    // ---------------- input-gradient chain (main stream) ----------------
    const bf16* wp = w.wpack + (size_t)l * lay.layer_weights;
    if (l == top && tail) {
      // Top block, pruned tail: dL/dX[L] lives on the B CLS rows only (w.dxcls; masked copy w.t_gb2), so the MLP half and the
      // out-projection run on compact [B, .] tensors; their results go back into zero-filled full-size tensors where the
      // attention backward (d ao) and the QKV input gradient's residual operand (d x_mid) need every row.
      const bf16* gb2c = drop ? w.t_gb2 : w.dxcls;
      VG_TRY(lin_dgrad(gb2c, Pb + lo + lay.w2, w.t_dz1, B, E, rE, VG_ACT_MUL_Z8, (const bf16*)w.t_z1, nullptr, 0.f, st));
      if (tail_row) {  // fc1 input gradient + norm2 backward in the full-row kernel, M = B (the forward's twin: see there)
        VG_TRY(row_bwd(w.t_dz1, rE, wp + po_w1T, w.t_xmid, w.t_mean2, w.t_rstd2, P + lo + lay.ln2_w, w.dxcls, w.t_dxmid, drop ? w.t_gb1 : nullptr, part2,
                       1 + 2 * l, B, S));
      } else {
        VG_TRY(lin_dgrad(w.t_dz1, Pb + lo + lay.w1, w.t_dxn2, B, rE, E, 0, nullptr, nullptr, 0.f, st));
        VG_TRY(vg_ln_bwd_launch(w.t_dxn2, w.t_xmid, w.t_mean2, w.t_rstd2, P + lo + lay.ln2_w, w.dxcls, w.t_dxmid, part2, B, E, drop ? w.t_gb1 : nullptr,
                                dr.thr, site_key(dr, 1 + 2 * l), dr.scale, dr.step, st, 1, S));
      }
      const bf16* gb1c = drop ? w.t_gb1 : w.t_dxmid;
      VG_TRY(lin_dgrad(gb1c, Pb + lo + lay.wo, w.t_dao, B, E, E, 0, nullptr, nullptr, 0.f, st));
      if (!net->attn_fp8) VG_TRY(vg_scatter_cls_launch(w.t_dxmid, cur.gmid, B, S, E, st));  // (d ao stays compact: the CLS-query attention backward below)
      else VG_TRY(vg_scatter_cls2_launch(w.t_dao, w.dao, w.t_dxmid, cur.gmid, B, S, E, st));
    } else {
    // d a1 = gb2 W2 ; dz1 = d a1 * gelu'(pre-activation), stored by the forward   (fused epilogue)
    if (inj && inj->preact) VG_TRY(lin_dgrad(gb2, Pb + lo + lay.w2, cur.dz1, M, E, rE, VG_ACT_MUL_GELU_GRAD, inj->preact + (size_t)l * M * rE, nullptr, 0.f, st));
    else VG_TRY(lin_dgrad(gb2, Pb + lo + lay.w2, cur.dz1, M, E, rE, VG_ACT_MUL_Z8, (const bf16*)z1, nullptr, 0.f, st));
    if (inj && inj->s_h) VG_TRY(vg_add_bf16_launch(cur.dz1, inj->s_h + (size_t)l * M * rE, cur.dz1, (long long)M * rE, st));
    const bf16* gres2 = g;  // the residual-stream gradient the norm2 backward adds
    const bf16* inj_mid = (inj && inj->s_xmid) ? inj->s_xmid + (size_t)l * ME : nullptr;
    if (inj_mid && !rown) { VG_TRY(vg_add_bf16_launch(g, inj_mid, inj->tmp, (long long)ME, st)); gres2 = inj->tmp; }  // (the full-row kernel takes it as an operand)
    if (rown) {  // fc1 input gradient + norm2 backward + the residual-stream gradient
      VG_TRY(row_bwd(cur.dz1, rE, wp + po_w1T, xmid, w.mean2 + (size_t)l * M, w.rstd2 + (size_t)l * M, P + lo + lay.ln2_w, gres2, cur.gmid,
                     drop ? cur.gm1 : nullptr, part2, 1 + 2 * l, 0, 1, inj_mid));
    } else {
      VG_TRY(lin_dgrad(cur.dz1, Pb + lo + lay.w1, w.dxn, M, rE, E, 0, nullptr, nullptr, 0.f, st));
      VG_TRY(vg_ln_bwd_launch(w.dxn, xmid, w.mean2 + (size_t)l * M, w.rstd2 + (size_t)l * M, P + lo + lay.ln2_w, gres2, cur.gmid, part2, M, E,
                              drop ? cur.gm1 : nullptr, dr.thr, site_key(dr, 1 + 2 * l), dr.scale, dr.step, st));
    }
    const bf16* gb1 = drop ? cur.gm1 : cur.gmid;  // gradient w.r.t. the out-projection output (before dropout1)
    VG_TRY(lin_dgrad(gb1, Pb + lo + lay.wo, w.dao, M, E, E, 0, nullptr, nullptr, 0.f, st));
    }
    if (l == top && tail && !net->attn_fp8)
      VG_TRY(vg_attn_cls_bwd_launch(qkv, w.t_ao, w.t_dao, w.t_lse, cur.dqkv, B, d.H, S, HE, 1.0f / sqrtf((float)HE), st));
    else
      VG_TRY(vg_attn_bwd_launch(qkv, ao, w.dao, w.lse + (size_t)l * B * d.H * S, cur.dqkv, B, d.H, S, HE, 1.0f / sqrtf((float)HE), net->attn_fp8 ? 2 : 0, st));
    if (inj && inj->s_qkv) VG_TRY(vg_add_bf16_launch(cur.dqkv, inj->s_qkv + (size_t)l * ME * 3, cur.dqkv, (long long)ME * 3, st));
    if (!rown) VG_TRY(lin_dgrad(cur.dqkv, Pb + lo + lay.wqkv, w.dxn, M, 3 * E, E, 0, nullptr, nullptr, 0.f, st));
    if (pairing) {  // second block of a pair (or the odd one out at the end of this call): its and its partner's weight gradients
      const int idx = l_hi - l;
      if (idx & 1) VG_TRY(wgrad_blocks(l + 1, 2));
      else if (l == l_lo) VG_TRY(wgrad_blocks(l, 1));
    }
    // LN1 backward writes dL/dX[l] (and its masked copy for the dropout it meets next) into the OTHER set, which the
    // weight-gradient side of block l+1 may still be reading: wait for it first
    if (ctx && want_wgrad && l + 1 <= top && l + 1 >= 0 && (d.L - (l + 1)) >= stage_begin)
      VG_CHECK_HIP(hipStreamWaitEvent(st, ctx->ev_side[l + 1], 0));
    const bf16* gres1 = cur.gmid;
    const bf16* inj_x = (inj && inj->s_x) ? inj->s_x + (size_t)l * ME : nullptr;
    if (inj_x && !rown) { VG_TRY(vg_add_bf16_launch(cur.gmid, inj_x, inj->tmp, (long long)ME, st)); gres1 = inj->tmp; }
    if (rown) {  // QKV input gradient + norm1 backward + the residual-stream gradient
      VG_TRY(row_bwd(cur.dqkv, 3 * E, wp + po_wqkvT, x, w.mean1 + (size_t)l * M, w.rstd1 + (size_t)l * M, P + lo + lay.ln1_w, gres1, nxt.gin,
                     drop ? nxt.gm2 : nullptr, part1, l > 0 ? 2 + 2 * (l - 1) : 0, 0, 1, inj_x));
    } else {
      VG_TRY(vg_ln_bwd_launch(w.dxn, x, w.mean1 + (size_t)l * M, w.rstd1 + (size_t)l * M, P + lo + lay.ln1_w, gres1, nxt.gin, part1, M, E,
                              drop ? nxt.gm2 : nullptr, dr.thr, site_key(dr, l > 0 ? 2 + 2 * (l - 1) : 0), dr.scale, dr.step, st));
    }
    if (!want_wgrad) continue;
    // ---------------- weight-gradient side (second stream when a context is given) ----------------
    if (ctx) {
      VG_CHECK_HIP(hipEventRecord(ctx->ev_main[l], st));
      VG_CHECK_HIP(hipStreamWaitEvent(sd, ctx->ev_main[l], 0));
    }
    VG_TRY(vg_fold_push(folds, part2, (l == top && tail) ? (tail_row ? vg_row_nwg(B) : vg_ln_bwd_nparts(B)) : lnparts, 3 * E, G + lo + lay.ln2_w, E, G + lo + lay.ln2_b, E, G + lo + lay.bo, E, nullptr, 0));
    if (!pairing) VG_TRY(wgrad_blocks(l, 1));  // side-stream schedule: block by block, behind the block's input-gradient chain
    {
      float* b2_prev = (l > 0) ? G + (lo - lay.layer_stride) + lay.b2 : nullptr;
      VG_TRY(vg_fold_push(folds, part1, lnparts, 3 * E, G + lo + lay.ln1_w, E, G + lo + lay.ln1_b, E, b2_prev, E, nullptr, 0));
    }
    if (ctx) { VG_CHECK_HIP(hipEventRecord(ctx->ev_side[l], sd)); last_side = l; }
  }
  // all LayerNorm partial sums of this call in one launch (behind the last block's side work)
  if (folds.n > 0) {
    // the partial rows come from kernels of the MAIN stream (head, final LayerNorm, the LayerNorm backwards): a call that runs no
    // encoder block (stage range [0, 1)) has recorded no main-stream event the side stream waits for - without this one the fold
    // raced the final LayerNorm's backward (seen at E = 768: d gamma / d beta of vit.norm zero or partial, run to run)
    if (ctx) { VG_CHECK_HIP(hipEventRecord(ctx->ev_main[VG_CTX_EVENTS - 1], st)); VG_CHECK_HIP(hipStreamWaitEvent(sd, ctx->ev_main[VG_CTX_EVENTS - 1], 0)); }
    VG_TRY(vg_colsum_f32_multi_launch(folds, sd));
    if (ctx) { VG_CHECK_HIP(hipEventRecord(ctx->ev_side[VG_CTX_EVENTS - 1], sd)); VG_CHECK_HIP(hipStreamWaitEvent(st, ctx->ev_side[VG_CTX_EVENTS - 1], 0)); }
  }
  // join: everything this call put on the side stream is ordered before whatever follows on the main stream
  if (ctx && last_side >= 0) VG_CHECK_HIP(hipStreamWaitEvent(st, ctx->ev_side[last_side], 0));
  bf16* g = w.set[1].gin;  // dL/dX[0]: block 0 (set 0) wrote it into the other set
  const bf16* g0m = w.set[1].gm2;

  if (stage_end < d.L + 2) return 0;
  // ---- patch embedding ----
  if (drop) g = (bf16*)g0m;  // dL/dX[0] masked by the embedding dropout (second output of block 0's LN1 backward)
  if (want_wgrad) {
    VG_TRY(vg_batch_sum_launch(g, w.tok_sum, B, S, E, st));
    VG_TRY(vg_embed_small_grads_launch(w.tok_sum, G + lay.cls, G + lay.pos, G + lay.conv_b, S, E, st));
  }
  if (want_wgrad || d_img) VG_TRY(vg_take_rows_launch(g, w.gp, B, S, 1, NP, E, st));
  if (want_wgrad) {
    const int splits = pick_splits(tiles128(E, Kp), B * NP, EMB_SPLIT_CAP);
    VgGemmProb p = wg(w.gp, E, w.Apatch, Kp, B * NP, w.slab, (long long)E * Kp, splits);
    VG_TRY(vg_gemm_launch(&p, 1, VG_TN, st));
    VG_TRY(vg_slab_reduce_launch(w.slab, (long long)E * Kp, p.splits, G + lay.conv_w, (long long)E * Kp, 1, st));
  }
  if (d_img) {
    VG_TRY(lin_dgrad(w.gp, Pb + lay.conv_w, w.dA, B * NP, E, Kp, 0, nullptr, nullptr, 0.f, st));
    VG_TRY(vg_unpatchify_launch(w.dA, (bf16*)d_img, B, d.C, d.IH, d.P, st));
  }
  return 0;
}

extern "C" int vg_vit_backward_stages(const VgVitNet* net, int B, void* ws, const float* dlogits, void* d_img, int want_wgrad,
                                      int stage_begin, int stage_end, void* stream) {
  return vit_backward_impl(net, B, ws, dlogits, d_img, want_wgrad, stage_begin, stage_end, stream, nullptr);
}
extern "C" int vg_vit_backward(const VgVitNet* net, int B, void* ws, const float* dlogits, void* d_img, int want_wgrad,
                               void* stream) {
  if (!net) return -1;
  return vg_vit_backward_stages(net, B, ws, dlogits, d_img, want_wgrad, 0, net->d.L + 2, stream);
}

// =============================================================================================
//            gradient penalty (src/v2/utils.py:124-144, training.py:101-106) as ONE call
// =============================================================================================
// G += weight * d/d theta mean_b (|| d sum(D(x^)) / d x^ ||_2 - 1)^2 at x^ = eps real + (1 - eps) fake: five passes on one stream.
//   1. forward of x^ (every row of the top block; fc1's pre-activation kept - gelu'' needs it);
//   2. first backward, input gradient only, UNFUSED and with every intermediate gradient kept per block: they are the "dY operands"
//      of the second-order operators;
//   3. n_b = ||g_b||, the penalty, and u = d(weight * penalty) / d g;
//   4. the backward of pass 2 (the direction u travels UP the network: it is the forward-mode tangent of pass 1): per block the
//      LayerNorm / attention / GELU second-order kernels (second_order.hip, attention.hip), the Linear layers as forward GEMMs
//      (d(dY) = ddX W^T) and weight gradients dW += dY^T ddX (one grouped launch per block); each second-order kernel also yields a
//      gradient with respect to a forward activation (X[l], qkv, x_mid, the fc1 pre-activation) that
//   5. the ordinary fused backward of pass 1 picks up where it reaches that activation (VitInject) - with nothing arriving from the logits.
// The operator arithmetic is that of vit-gan_amd/ops2.py (the autograd form this replaces, kept as the reference the tests compare
// with); dropout draws the engine's counter-based masks of net->dropout_seed, the same in all five passes.
struct PenWs {
  float *xhat, *ones, *logits, *pen_img, *pbb;
  bf16 *h, *gin, *gm2, *da1, *dz1, *dxn2, *gmid, *gm1, *dao, *dqkv, *dxn1, *g0, *g0m, *xcls;
  bf16 *u_dA, *u_x[2], *u_dxn[2], *u_dxn2[2], *u_dqkv, *u_dao[2], *u_gmid, *u_dz1, *u_da1[2], *ucls, *u_gc, *u_gpre, *u_gt;  // [2]: operands of a PAIR of blocks' weight gradients
  bf16 *s_x, *s_qkv, *s_xmid, *s_h, *s_xcls, *tmp;
};
static long long carve_pen(const VgVitDims& d, int B, void* base, PenWs& q) {
  const long long E = d.E, NP = (long long)(d.IH / d.P) * (d.IH / d.P), S = NP + 1, M = (long long)B * S;
  const long long Kp = (long long)d.C * d.P * d.P, L = d.L, rE = (long long)d.R * E;
  Carver c{(unsigned char*)base, 0};
  q.xhat = c.take<float>((long long)B * d.C * d.IH * d.IH);
  q.ones = c.take<float>((long long)B * d.Kc); q.logits = c.take<float>((long long)B * d.Kc); q.pen_img = c.take<float>(B);
  q.pbb = c.take<float>((2 * L + 1) * (long long)vg_ln_bwd_bwd_nparts((int)M) * E);
  q.h = c.take<bf16>(L * M * rE);
  q.gin = c.take<bf16>(L * M * E); q.gm2 = c.take<bf16>(L * M * E);
  q.da1 = c.take<bf16>(L * M * rE); q.dz1 = c.take<bf16>(L * M * rE);
  q.dxn2 = c.take<bf16>(L * M * E); q.gmid = c.take<bf16>(L * M * E); q.gm1 = c.take<bf16>(L * M * E);
  q.dao = c.take<bf16>(L * M * E); q.dqkv = c.take<bf16>(L * M * 3 * E); q.dxn1 = c.take<bf16>(L * M * E);
  q.g0 = c.take<bf16>(M * E); q.g0m = c.take<bf16>(M * E); q.xcls = c.take<bf16>(B * E);
  q.u_dA = c.take<bf16>(B * NP * Kp);
  q.u_x[0] = c.take<bf16>(M * E); q.u_x[1] = c.take<bf16>(M * E);
  for (int i = 0; i < 2; ++i) {
    q.u_dxn[i] = c.take<bf16>(M * E); q.u_dxn2[i] = c.take<bf16>(M * E); q.u_dao[i] = c.take<bf16>(M * E); q.u_da1[i] = c.take<bf16>(M * rE);
  }
  q.u_dqkv = c.take<bf16>(M * 3 * E); q.u_gmid = c.take<bf16>(M * E); q.u_dz1 = c.take<bf16>(M * rE);
  q.ucls = c.take<bf16>(B * E); q.u_gc = c.take<bf16>(B * E); q.u_gpre = c.take<bf16>(B * E); q.u_gt = c.take<bf16>(B * E);
  q.s_x = c.take<bf16>(L * M * E); q.s_qkv = c.take<bf16>(L * M * 3 * E); q.s_xmid = c.take<bf16>(L * M * E); q.s_h = c.take<bf16>(L * M * rE);
  q.s_xcls = c.take<bf16>(B * E); q.tmp = c.take<bf16>(M * E);
  return c.off;
}
// Every network the plain step trains in bf16 (E a multiple of 128: the alignment of the elementwise kernels follows); where the full-row
// kernels take the shape (E = 384 / 512, rows in whole units of 16) the input gradients and LayerNorm backwards of passes 2 and 5 are fused,
// elsewhere they are the GEMM + LayerNorm pairs.  -3: fp8 attention (the second-order attention kernel differentiates the bf16 one).
static int pen_shape_ok(const VgVitNet* net, int B) {
  const VgVitDims& d = net->d;
  return !net->attn_fp8 && ((long long)B * d.E) % 8 == 0 && ((long long)d.C * d.IH * d.IH) % 4 == 0 && ((long long)d.C * d.P * d.P) % 4 == 0;
}
extern "C" long long vg_vit_penalty_ws_bytes(const VgVitDims* d, int B) {
  VgVitLayout lay;
  if (!d || B < 1 || vg_vit_layout(d, &lay)) return -1;
  PenWs q;
  return carve_pen(*d, B, nullptr, q);
}
extern "C" int vg_vit_penalty(const VgVitNet* net0, int B, const void* real, const void* fake, const float* eps, float weight, void* ws,
                              void* ws_pen, float* penalty_out, void* stream) {
  if (!net0 || !real || !fake || !eps || !ws || !ws_pen || !penalty_out || B < 1 || !net0->G) return -1;
  if (!pen_shape_ok(net0, B)) return -3;
  VgVitNet net_ = *net0;
  net_.dense_top = 1; net_.ctx = nullptr;
  const VgVitNet* net = &net_;
  const VgVitDims& d = net->d;
  VgVitLayout lay;
  VG_TRY(vg_vit_layout(&d, &lay));
  hipStream_t st = (hipStream_t)stream;
  const int E = d.E, NP = (d.IH / d.P) * (d.IH / d.P), S = NP + 1, M = B * S, Kp = d.C * d.P * d.P, rE = d.R * E, HE = E / d.H, L = d.L;
  const float scale = 1.0f / sqrtf((float)HE);
  VitWs w; carve_vit(d, B, ws, w);
  PenWs q; carve_pen(d, B, ws_pen, q);
  const float* P = net->P; const bf16* Pb = (const bf16*)net->Pb; float* G = net->G;
  const size_t ME = (size_t)M * E, MR = (size_t)M * rE;
  const Drop dr = mk_drop(net->dropout_p, net->dropout_seed, net->dropout_step);
  const bool drop = dr.thr != 0;
  const int top = L - 1;

  // ---- 1. forward of the interpolated images ----
  VG_TRY(vg_pen_interp_launch((const bf16*)real, (const bf16*)fake, eps, q.xhat, B, (long long)d.C * d.IH * d.IH, st));
  VG_TRY(vit_forward_impl(net, B, q.xhat, 0, ws, q.logits, stream, q.h));

  // ---- 2. first backward: d sum(logits) / d x^, every intermediate kept ----
  VG_TRY(vg_fill_f32_launch(q.ones, (long long)B * d.Kc, 1.0f, st));
  { const int r = vg_head_bwd_launch(q.ones, P + lay.hw2, w.th, w.dzh, nullptr, nullptr, B, E, d.Kc, 0, st); if (r < 0) return -r; }  // g_pre
  VG_TRY(lin_dgrad(w.dzh, Pb + lay.hw1, w.dhcls, B, E, E, 0, nullptr, nullptr, 0.f, st));                                                  // g_c
  VG_TRY(vg_ln_bwd_launch(w.dhcls, w.X + (size_t)L * ME, w.meanf, w.rstdf, P + lay.lnf_w, nullptr, w.dxcls, w.part, B, E, nullptr, 0, 0, 1.f, nullptr, st, S));
  VG_TRY(vg_scatter_cls_launch(w.dxcls, q.gin + (size_t)top * ME, B, S, E, st, drop ? q.gm2 + (size_t)top * ME : nullptr, dr.thr, site_key(dr, 2 + 2 * top),
                               dr.scale, dr.step));
  const long long po_wqkvT = (long long)E * E + (long long)E * rE, po_w1T = po_wqkvT + 3LL * E * E;  // the forward packed these images (carve_vit: wpack)
  const int rown = vit_row_nwg(d, M);  // 0: no full-row kernel for this shape - the unfused pairs
  auto pen_row = [&](const bf16* A, int K, const bf16* Wp, const bf16* x, const float* mean, const float* rstd, const float* gamma, const bf16* gres,
                     bf16* dx, bf16* dxm, bf16* dy_out, int site) -> int {
    VgRowArgs ra = {};
    ra.N = E; ra.A = A; ra.lda = K; ra.Wp = Wp; ra.M = M; ra.K = K; ra.x = x; ra.mean = mean; ra.rstd = rstd; ra.gamma = gamma;
    ra.gres = gres; ra.dx = dx; ra.dxm = dxm; ra.dy_out = dy_out; ra.drop_row_mul = 1;  // (part = nullptr: the input gradient only)
    if (dxm) { ra.drop_thresh = dr.thr; ra.drop_key = site_key(dr, site); ra.drop_scale = dr.scale; ra.drop_step = dr.step; }
    const int r = vg_gemm_row_launch(ra, VG_ROW_LNBWD_PEN, st);
    return r > 0 ? 0 : (r < 0 ? -r : -3);
  };
  for (int l = top; l >= 0; --l) {
    const long long lo = lay.layer0 + (long long)l * lay.layer_stride;
    const bf16* wp = w.wpack + (size_t)l * lay.layer_weights;
    const bf16* gin = q.gin + (size_t)l * ME;
    const bf16* gb2 = drop ? q.gm2 + (size_t)l * ME : gin;
    bf16 *da1 = q.da1 + (size_t)l * MR, *dz1 = q.dz1 + (size_t)l * MR, *dxn2 = q.dxn2 + (size_t)l * ME, *gmid = q.gmid + (size_t)l * ME;
    bf16 *gm1 = q.gm1 + (size_t)l * ME, *dao = q.dao + (size_t)l * ME, *dqkv = q.dqkv + (size_t)l * ME * 3, *dxn1 = q.dxn1 + (size_t)l * ME;
    VG_TRY(lin_dgrad(gb2, Pb + lo + lay.w2, da1, M, E, rE, 0, nullptr, nullptr, 0.f, st));
    VG_TRY(vg_act2_launch(q.h + (size_t)l * MR, da1, nullptr, dz1, nullptr, (long long)MR, 1, 1, st));
    // fc1 input gradient + norm2 backward in the full-row kernel, which here also WRITES the GEMM result (the double backward's d xn2)
    if (rown) {
      VG_TRY(pen_row(dz1, rE, wp + po_w1T, w.xmid + (size_t)l * ME, w.mean2 + (size_t)l * M, w.rstd2 + (size_t)l * M, P + lo + lay.ln2_w, gin, gmid,
                     drop ? gm1 : nullptr, dxn2, 1 + 2 * l));
    } else {
      VG_TRY(lin_dgrad(dz1, Pb + lo + lay.w1, dxn2, M, rE, E, 0, nullptr, nullptr, 0.f, st));
      VG_TRY(vg_ln_bwd_launch(dxn2, w.xmid + (size_t)l * ME, w.mean2 + (size_t)l * M, w.rstd2 + (size_t)l * M, P + lo + lay.ln2_w, gin, gmid, w.part, M, E,
                              drop ? gm1 : nullptr, dr.thr, site_key(dr, 1 + 2 * l), dr.scale, dr.step, st));
    }
    VG_TRY(lin_dgrad(drop ? gm1 : gmid, Pb + lo + lay.wo, dao, M, E, E, 0, nullptr, nullptr, 0.f, st));
    VG_TRY(vg_attn_bwd_launch(w.qkv + (size_t)l * ME * 3, w.ao + (size_t)l * ME, dao, w.lse + (size_t)l * B * d.H * S, dqkv, B, d.H, S, HE, scale, 0, st));
    bf16* gx = l > 0 ? q.gin + (size_t)(l - 1) * ME : q.g0;
    bf16* gxm = l > 0 ? q.gm2 + (size_t)(l - 1) * ME : q.g0m;
    if (rown) {
      VG_TRY(pen_row(dqkv, 3 * E, wp + po_wqkvT, w.X + (size_t)l * ME, w.mean1 + (size_t)l * M, w.rstd1 + (size_t)l * M, P + lo + lay.ln1_w, gmid, gx,
                     drop ? gxm : nullptr, dxn1, l > 0 ? 2 + 2 * (l - 1) : 0));
    } else {
      VG_TRY(lin_dgrad(dqkv, Pb + lo + lay.wqkv, dxn1, M, 3 * E, E, 0, nullptr, nullptr, 0.f, st));
      VG_TRY(vg_ln_bwd_launch(dxn1, w.X + (size_t)l * ME, w.mean1 + (size_t)l * M, w.rstd1 + (size_t)l * M, P + lo + lay.ln1_w, gmid, gx, w.part, M, E,
                              drop ? gxm : nullptr, dr.thr, site_key(dr, l > 0 ? 2 + 2 * (l - 1) : 0), dr.scale, dr.step, st));
    }
  }
  VG_TRY(vg_take_rows_launch(drop ? q.g0m : q.g0, w.gp, B, S, 1, NP, E, st));
  VG_TRY(lin_dgrad(w.gp, Pb + lay.conv_w, w.dA, B * NP, E, Kp, 0, nullptr, nullptr, 0.f, st));  // = the image gradient, patch by patch

  // ---- 3. the penalty and the direction of the second backward ----
  VG_TRY(vg_pen_norm_launch(w.dA, q.u_dA, q.pen_img, penalty_out, B, (long long)NP * Kp, weight, st));

  // ---- 4. backward of pass 2, bottom to top ----
  VgFoldJobs folds; folds.n = 0;
  const int bbparts = vg_ln_bwd_bwd_nparts(M);
  {  // patch embedding: d A = gp Wc  ->  u_gp = u_dA Wc^T (rows back behind the CLS rows, embedding dropout's mask), dWc += gp^T u_dA
    VG_TRY(vg_fill_f32_launch((float*)q.u_x[0], (long long)(ME / 2), 0.0f, st));  // (a kernel, not hipMemsetAsync: see DESIGN 7 - the memset node of a captured graph was not ordered with its neighbours)
    VgGemmProb p = mk(q.u_dA, Kp, Pb + lay.conv_w, Kp, B * NP, E, Kp);
    p.C = q.u_x[0]; p.ldc = E; p.row_in_per = NP; p.row_out_per = S; p.row_out_off = 1;
    set_drop(p, dr, 0, 1);
    VG_TRY(vg_gemm_launch(&p, 1, VG_NT, st));
    const int splits = pick_splits(tiles128(E, Kp), B * NP, EMB_SPLIT_CAP);
    VgGemmProb pw = wg(w.gp, E, q.u_dA, Kp, B * NP, w.slab, (long long)E * Kp, splits);
    VG_TRY(vg_gemm_launch(&pw, 1, VG_TN, st));
    VG_TRY(vg_slab_reduce_launch(w.slab, (long long)E * Kp, pw.splits, G + lay.conv_w, (long long)E * Kp, 1, st));
  }
  int cur = 0;
  // the weight gradients dW += dY^T ddX of TWO blocks go out as one grouped split-K launch + one fold (half the K slices each: half the slab
  // traffic per unit of work, like the engine's own backward), so the tangent operands of a block live in one of two buffer sets
  const long long wtiles = tiles128(3 * E, E) + tiles128(E, E) + tiles128(rE, E) + tiles128(E, rE);
  int sp = pick_splits(wtiles * 2, M, VIT_SPLIT_CAP / 2);
  if (const int bn = wide_bn(E, rE); bn && M % 32 == 0 && E % 128 == 0 && rE % 128 == 0)
    sp = pick_splits384(2 * (tiles_wide(3 * E, E, bn) + tiles_wide(E, E, bn) + tiles_wide(rE, E, bn) + tiles_wide(E, rE, bn)), M, VIT_SPLIT_CAP / 2);
  VgGemmProb pr[8];
  int npr = 0;
  for (int l = 0; l < L; ++l) {
    const long long lo = lay.layer0 + (long long)l * lay.layer_stride;
    const bf16* gb2 = drop ? q.gm2 + (size_t)l * ME : q.gin + (size_t)l * ME;
    const bf16* gb1 = drop ? q.gm1 + (size_t)l * ME : q.gmid + (size_t)l * ME;
    const bf16 *da1 = q.da1 + (size_t)l * MR, *dz1 = q.dz1 + (size_t)l * MR, *dxn2 = q.dxn2 + (size_t)l * ME;
    const bf16 *dao = q.dao + (size_t)l * ME, *dqkv = q.dqkv + (size_t)l * ME * 3, *dxn1 = q.dxn1 + (size_t)l * ME;
    const bf16* u_gx = q.u_x[cur];
    bf16* u_up = q.u_x[cur ^ 1];
    const int ps = l & 1;  // operand set, and this block's half of the slab
    bf16 *u_dxn = q.u_dxn[ps], *u_dxn2 = q.u_dxn2[ps], *u_dao = q.u_dao[ps], *u_da1 = q.u_da1[ps];
    float* slab = w.slab + (size_t)ps * sp * lay.layer_weights;
    float* pb1 = q.pbb + (size_t)(2 * l) * bbparts * E;
    float* pb2 = q.pbb + (size_t)(2 * l + 1) * bbparts * E;
    // gX = gmid + LN1'(dxn1; X): the norm's double backward; u reaches gmid unchanged (added below)
    VG_TRY(vg_ln_bwd_bwd_launch(u_gx, dxn1, w.X + (size_t)l * ME, w.mean1 + (size_t)l * M, w.rstd1 + (size_t)l * M, P + lo + lay.ln1_w, u_dxn,
                                q.s_x + (size_t)l * ME, pb1, M, E, st));
    VG_TRY(vg_fold_push(folds, pb1, bbparts, E, G + lo + lay.ln1_w, E, nullptr, 0, nullptr, 0, nullptr, 0));
    // dxn1 = dqkv Wqkv
    VG_TRY(lin_fwd(u_dxn, E, Pb + lo + lay.wqkv, nullptr, q.u_dqkv, M, 3 * E, VG_ACT_NONE, 0.f, nullptr, nullptr, nullptr, st));
    pr[npr++] = wg(dqkv, 3 * E, u_dxn, E, M, slab + lay.wqkv, lay.layer_weights, sp);
    // dqkv = attention'(dao; qkv)
    VG_TRY(vg_attn_bwd_bwd_launch(w.qkv + (size_t)l * ME * 3, dao, w.lse + (size_t)l * B * d.H * S, q.u_dqkv, u_dao, q.s_qkv + (size_t)l * ME * 3, B, d.H, S, HE,
                                  scale, st));
    // dao = gb1 Wo ; gb1 = mask1 gmid  ->  u_gmid = u_gX + mask1 (u_dao Wo^T)
    VG_TRY(lin_fwd(u_dao, E, Pb + lo + lay.wo, nullptr, q.u_gmid, M, E, VG_ACT_NONE, 0.f, u_gx, nullptr, nullptr, st, &dr, 1 + 2 * l));
    pr[npr++] = wg(gb1, E, u_dao, E, M, slab + lay.wo, lay.layer_weights, sp);
    // gmid = gin + LN2'(dxn2; x_mid)
    VG_TRY(vg_ln_bwd_bwd_launch(q.u_gmid, dxn2, w.xmid + (size_t)l * ME, w.mean2 + (size_t)l * M, w.rstd2 + (size_t)l * M, P + lo + lay.ln2_w, u_dxn2,
                                q.s_xmid + (size_t)l * ME, pb2, M, E, st));
    VG_TRY(vg_fold_push(folds, pb2, bbparts, E, G + lo + lay.ln2_w, E, nullptr, 0, nullptr, 0, nullptr, 0));
    // dxn2 = dz1 W1
    VG_TRY(lin_fwd(u_dxn2, E, Pb + lo + lay.w1, nullptr, q.u_dz1, M, rE, VG_ACT_NONE, 0.f, nullptr, nullptr, nullptr, st));
    pr[npr++] = wg(dz1, rE, u_dxn2, E, M, slab + lay.w1, lay.layer_weights, sp);
    // dz1 = da1 gelu'(h)
    VG_TRY(vg_act2_launch(q.h + (size_t)l * MR, da1, q.u_dz1, u_da1, q.s_h + (size_t)l * MR, (long long)MR, 1, 2, st));
    // da1 = gb2 W2 ; gb2 = mask2 gin  ->  u_gin = u_gmid + mask2 (u_da1 W2^T)
    VG_TRY(lin_fwd(u_da1, rE, Pb + lo + lay.w2, nullptr, u_up, M, E, VG_ACT_NONE, 0.f, q.u_gmid, nullptr, nullptr, st, &dr, 2 + 2 * l));
    pr[npr++] = wg(gb2, E, u_da1, rE, M, slab + lay.w2, lay.layer_weights, sp);
    if (ps == 1 || l == L - 1) {  // the pair (or the odd block out) is complete
      VG_TRY(vg_gemm_launch(pr, npr, VG_TN, st));
      const int ns = pr[0].splits;  // (the launcher may lower the slice count; every problem has the same M rows)
      if (npr == 8)
        VG_TRY(vg_slab_reduce2_launch(w.slab, w.slab + (size_t)sp * lay.layer_weights, lay.layer_weights, ns, G + lo - lay.layer_stride, G + lo, lay.layer_weights, 1, st));
      else
        VG_TRY(vg_slab_reduce_launch(w.slab, lay.layer_weights, ns, G + lo, lay.layer_weights, 1, st));
      npr = 0;
    }
    cur ^= 1;
  }
  {  // final LayerNorm on the CLS rows and the classifier head
    float* pbf = q.pbb + (size_t)(2 * L) * bbparts * E;
    VG_TRY(vg_take_rows_launch(q.u_x[cur], q.ucls, B, S, 0, 1, E, st));
    VG_TRY(vg_take_rows_launch(w.X + (size_t)L * ME, q.xcls, B, S, 0, 1, E, st));
    VG_TRY(vg_ln_bwd_bwd_launch(q.ucls, w.dhcls, q.xcls, w.meanf, w.rstdf, P + lay.lnf_w, q.u_gc, q.s_xcls, pbf, B, E, st));
    VG_TRY(vg_fold_push(folds, pbf, vg_ln_bwd_bwd_nparts(B), E, G + lay.lnf_w, E, nullptr, 0, nullptr, 0, nullptr, 0));
    VG_TRY(lin_fwd(q.u_gc, E, Pb + lay.hw1, nullptr, q.u_gpre, B, E, VG_ACT_NONE, 0.f, nullptr, nullptr, nullptr, st));  // g_c = g_pre Wh1
    VgGemmProb p = wg(w.dzh, E, q.u_gc, E, B, w.slab, (long long)E * E, B >= 512 ? 4 : (B >= 256 ? 2 : 1));
    VG_TRY(vg_gemm_launch(&p, 1, VG_TN, st));
    VG_TRY(vg_slab_reduce_launch(w.slab, (long long)E * E, p.splits, G + lay.hw1, (long long)E * E, 1, st));
    // g_pre = g_t tanh'(p): u_gt (its batch sum is every row of dWh2) and dL/dp, which the second backward starts from (in w.dzh)
    VG_TRY(vg_pen_head2_launch(q.u_gpre, w.th, P + lay.hw2, q.u_gt, w.dzh, B, E, d.Kc, st));
    for (int k = 0; k < d.Kc; ++k) VG_TRY(vg_colsum_bf16_launch(q.u_gt, E, B, E, w.part_cs, G + lay.hw2 + (long long)k * E, 1, st));
  }
  VG_TRY(vg_colsum_f32_multi_launch(folds, st));

  // ---- 5. the ordinary backward of pass 1 under the injected gradients ----
  VitInject inj = {};
  inj.preact = q.h; inj.head_given = 1; inj.s_xcls = q.s_xcls; inj.s_h = q.s_h; inj.s_xmid = q.s_xmid; inj.s_qkv = q.s_qkv; inj.s_x = q.s_x; inj.tmp = q.tmp;
  return vit_backward_impl(net, B, ws, q.ones, nullptr, 1, 0, L + 2, stream, &inj);
}

// =============================================================================================
//                                   generator (v1 SLN / SIREN)
// =============================================================================================
#define GEN_SPLIT_CAP 16
struct GenWs {
  bf16 *zb, *wmod, *s1, *qkv, *cat, *htmp, *s2, *hout, *sf, *y1;
  float *lse, *mean1, *rstd1, *mean2, *rstd2, *meanf, *rstdf, *zf1, *zf2;
  bf16 *g[3], *gm[2], *dz2, *dz1, *ds, *dcat, *dqkv, *dwb;
  bf16 *y2, *dy2;  // patch-grid variant only: token rows [R, CW] before the un-patchify / after the patchify of d_img
  float *dw_acc, *part, *part_cs, *part_cs2, *emb_sum, *slab;
  bf16* wpack;  // E = 384: stage images of Wo | Wm | Wqkv^T | Wm^T per block, then s1_w^T, for the full-row GEMMs (gemm_row.hip)
  bf16 *pdqkv, *pgm1, *pgm2;  // per-block copies of the weight-gradient dY operands (dropout on): two blocks' weight gradients go out as one launch
};
// generator rows R = B*T: the full-row kernels (SLN in the epilogue) take the Linears whose output is the embedding when E = 384
static inline int gen_row_nwg(const VgGenDims& d, int R) { return (vg_row_width_ok(d.E) && d.O % 64 == 0 && d.O >= 128) ? vg_row_nwg(R) : 0; }
static inline long long gen_pack_block(const VgGenDims& d) { return 6LL * d.E * d.E; }  // E*E + E*E + 3E*E + E*E
static long long carve_gen(const VgGenDims& d, int B, void* base, GenWs& w) {
  const long long E = d.E, T = d.T, R = (long long)B * T, L = d.L;
  VgGenLayout lay; vg_gen_layout(&d, &lay);
  Carver c{(unsigned char*)base, 0};
  w.zb = c.take<bf16>((long long)B * d.Z);
  w.wmod = c.take<bf16>(R * E);
  w.s1 = c.take<bf16>(L * R * E);
  w.qkv = c.take<bf16>(L * R * 3 * E);
  w.cat = c.take<bf16>(L * R * E);
  w.htmp = c.take<bf16>(L * R * E);
  w.s2 = c.take<bf16>(L * R * E);
  w.hout = c.take<bf16>(L * R * E);
  w.sf = c.take<bf16>(R * E);
  w.y1 = c.take<bf16>(R * d.O);
  w.lse = c.take<float>(L * (long long)B * d.H * T);
  w.mean1 = c.take<float>(L * R); w.rstd1 = c.take<float>(L * R);
  w.mean2 = c.take<float>(L * R); w.rstd2 = c.take<float>(L * R);
  w.meanf = c.take<float>(R); w.rstdf = c.take<float>(R);
  w.zf1 = c.take<float>(R * d.O);
  w.zf2 = c.take<float>(R * d.CW);
  for (int i = 0; i < 3; ++i) w.g[i] = c.take<bf16>(R * E);
  for (int i = 0; i < 2; ++i) w.gm[i] = c.take<bf16>(R * E);
  w.dz2 = c.take<bf16>(R * d.CW);
  w.dz1 = c.take<bf16>(R * d.O);
  w.ds = c.take<bf16>(R * E);
  w.dcat = c.take<bf16>(R * E);
  w.dqkv = c.take<bf16>(R * 3 * E);
  w.dwb = c.take<bf16>(R * E);
  w.y2 = c.take<bf16>(d.patch > 0 ? R * d.CW : 0);
  w.dy2 = c.take<bf16>(d.patch > 0 ? R * d.CW : 0);
  w.dw_acc = c.take<float>(R * E);
  w.pdqkv = c.take<bf16>(L * R * 3 * E); w.pgm1 = c.take<bf16>(L * R * E); w.pgm2 = c.take<bf16>(L * R * E);
  w.part = c.take<float>((2 * L + 1) * (long long)vg_ln_bwd_nparts((int)R) * (3 * E + 64));  // one block per SLN backward
  w.part_cs = c.take<float>((long long)vg_colsum_bf16_nparts((int)R) * (d.O > 3 * E ? d.O : 3 * E));
  w.part_cs2 = c.take<float>((long long)vg_colsum_bf16_nparts((int)R) * d.CW);
  w.emb_sum = c.take<float>(T * E);
  long long slab = GEN_SPLIT_CAP * lay.layer_weights;
  if (GEN_SPLIT_CAP * (long long)d.O * E > slab) slab = GEN_SPLIT_CAP * (long long)d.O * E;
  w.slab = c.take<float>(slab);
  w.wpack = c.take<bf16>(gen_row_nwg(d, (int)R) ? L * gen_pack_block(d) + (long long)d.O * E : 0);
  return c.off;
}
extern "C" long long vg_gen_ws_bytes(const VgGenDims* d, int B) {
  VgGenLayout lay;
  if (!d || B < 1 || vg_gen_layout(d, &lay)) return -1;
  GenWs w;
  return carve_gen(*d, B, nullptr, w);
}

extern "C" int vg_gen_ws_map(const VgGenDims* d, int B, VgGenWsMap* o) {
  VgGenLayout lay;
  if (!d || !o || B < 1 || vg_gen_layout(d, &lay)) return -1;
  unsigned char* const fake = (unsigned char*)(uintptr_t)(1u << 20);  // never dereferenced
  GenWs w;
  o->total = carve_gen(*d, B, fake, w);
  auto off = [&](const void* p) { return (long long)((const unsigned char*)p - fake); };
  o->wmod = off(w.wmod); o->s1 = off(w.s1); o->qkv = off(w.qkv); o->cat = off(w.cat); o->htmp = off(w.htmp); o->s2 = off(w.s2);
  o->hout = off(w.hout); o->sf = off(w.sf); o->y1 = off(w.y1); o->zf1 = off(w.zf1); o->zf2 = off(w.zf2);
  for (int i = 0; i < 3; ++i) o->g[i] = off(w.g[i]);
  o->dw_acc = off(w.dw_acc);
  return 0;
}

extern "C" int vg_gen_forward(const VgGenNet* net, int B, const float* z, void* ws, void* img, void* stream) {
  if (!net || !z || !ws || !img || B < 1) return -1;
  const VgGenDims& d = net->d;
  VgGenLayout lay;
  VG_TRY(vg_gen_layout(&d, &lay));
  hipStream_t st = (hipStream_t)stream;
  const int E = d.E, T = d.T, R = B * T, HE = E / d.H;
  GenWs w; carve_gen(d, B, ws, w);
  const float* P = net->P; const bf16* Pb = (const bf16*)net->Pb;
  const size_t RE = (size_t)R * E;
  const float scale = 1.0f / sqrtf((float)E);  // softmax(q.k / sqrt(H*hd)), src/v1/attention.py:51,90
  const Drop dr = mk_drop(net->dropout_p, net->dropout_seed, net->dropout_step);  // sites: 100+2l after output_linear, 101+2l inside the MLP

  // mapping network (generator.py:59-61): w = Linear(z) viewed [B*T, E]
  VG_TRY(vg_cast_f32_bf16_launch(z, w.zb, (long long)B * d.Z, st));
  VG_TRY(lin_fwd(w.zb, d.Z, Pb + lay.map_w, P + lay.map_b, w.wmod, B, T * E, VG_ACT_NONE, 0.f, nullptr, nullptr, nullptr, st));

  // full-row path: every Linear whose output is the embedding carries the SLN behind it in its epilogue (gemm_row.hip)
  const int rown = gen_row_nwg(d, R);
  const long long pb = gen_pack_block(d), po_wo = 0, po_wm = (long long)E * E, po_wqkvT = 2LL * E * E, po_wmT = 5LL * E * E;
  if (rown) {
    VgPackJobs pj;
    pj.N = E;
    pj.src = Pb + lay.layer0; pj.dst = w.wpack; pj.src_stride = lay.layer_stride; pj.dst_stride = pb; pj.nblocks = d.L; pj.n = 4;
    pj.d[0] = {lay.wo, po_wo, E, E, 0};           // output_linear forward
    pj.d[1] = {lay.wm, po_wm, E, E, 0};           // block MLP forward
    pj.d[2] = {lay.wqkv, po_wqkvT, 3 * E, E, 1};  // q|k|v input gradient
    pj.d[3] = {lay.wm, po_wmT, E, E, 1};          // block MLP input gradient
    VG_TRY(vg_pack_rows_launch(pj, st));
    VgPackJobs ph;                                // first SIREN layer's input gradient: s1_w [O, E] read transposed
    ph.N = E;
    ph.src = Pb + lay.s1_w; ph.dst = w.wpack + (long long)d.L * pb; ph.src_stride = 0; ph.dst_stride = 0; ph.nblocks = 1; ph.n = 1;
    ph.d[0] = {0, 0, d.O, E, 1};
    VG_TRY(vg_pack_rows_launch(ph, st));
  }
  // y = (res | emb table) + drop(A W^T + b);  yn = SLN(y, w)
  auto row_fwd = [&](const bf16* A, const bf16* Wp, const float* bias, const bf16* res, const float* resf, bf16* Y, bf16* Yn, float* mean,
                     float* rstd, const float* lw, const float* lb, const float* sc, int site) -> int {
    VgRowArgs ra = {};
    ra.N = E;
    ra.A = A; ra.lda = E; ra.Wp = Wp; ra.M = R; ra.K = E; ra.bias = bias; ra.res = res; ra.resf = resf; ra.res_period = T; ra.Y = Y; ra.Yn = Yn;
    ra.mean_out = mean; ra.rstd_out = rstd; ra.gamma = lw; ra.beta = lb; ra.eps = 1e-5f; ra.wmod = w.wmod; ra.gs = sc; ra.bs = sc + 1;
    if (dr.thr) { ra.drop_thresh = dr.thr; ra.drop_key = site_key(dr, site); ra.drop_scale = dr.scale; ra.drop_step = dr.step; }
    const int r = vg_gemm_row_launch(ra, VG_ROW_LNFWD, st);
    return r > 0 ? 0 : (r < 0 ? -r : -3);
  };

  for (int l = 0; l < d.L; ++l) {
    const long long lo = lay.layer0 + (long long)l * lay.layer_stride;
    const bf16* h = (l == 0) ? Pb + lay.emb : w.hout + (size_t)(l - 1) * RE;
    const int hb = (l == 0) ? T : 0;
    bf16* s1 = w.s1 + (size_t)l * RE;
    bf16* qkv = w.qkv + (size_t)l * RE * 3;
    bf16* cat = w.cat + (size_t)l * RE;
    bf16* htmp = w.htmp + (size_t)l * RE;
    bf16* s2 = w.s2 + (size_t)l * RE;
    const bf16* wp = w.wpack + (size_t)l * pb;
    if (!rown || l == 0)  // block 0 normalises the broadcast embedding; later blocks got s1 from the MLP epilogue of the block below
      VG_TRY(vg_sln_fwd_launch(h, hb, w.wmod, P + lo + lay.sln1_w, P + lo + lay.sln1_b, P + lo + lay.sln1_s, P + lo + lay.sln1_s + 1,
                               s1, w.mean1 + (size_t)l * R, w.rstd1 + (size_t)l * R, R, E, 1e-5f, st));
    VG_TRY(lin_fwd(s1, E, Pb + lo + lay.wqkv, nullptr, qkv, R, 3 * E, VG_ACT_NONE, 0.f, nullptr, nullptr, nullptr, st));
    VG_TRY(vg_attn_fwd_launch(qkv, cat, w.lse + (size_t)l * B * d.H * T, B, d.H, T, HE, scale, 0, st));
    if (rown) {  // htmp = output_linear(cat) + h (block 0: + the broadcast embedding) and SLN2(htmp) in one kernel
      VG_TRY(row_fwd(cat, wp + po_wo, P + lo + lay.bo, l == 0 ? nullptr : h, l == 0 ? P + lay.emb : nullptr, htmp, s2,
                     w.mean2 + (size_t)l * R, w.rstd2 + (size_t)l * R, P + lo + lay.sln2_w, P + lo + lay.sln2_b, P + lo + lay.sln2_s, 100 + 2 * l));
      // hout = drop(mlp(s2)) + htmp and the NEXT SLN of it: the block above's SLN1, or the final SLN in front of the SIREN
      const bool nx = l + 1 < d.L;
      const long long ln = lo + lay.layer_stride;
      VG_TRY(row_fwd(s2, wp + po_wm, P + lo + lay.bm, htmp, nullptr, w.hout + (size_t)l * RE, nx ? w.s1 + (size_t)(l + 1) * RE : w.sf,
                     nx ? w.mean1 + (size_t)(l + 1) * R : w.meanf, nx ? w.rstd1 + (size_t)(l + 1) * R : w.rstdf,
                     nx ? P + ln + lay.sln1_w : P + lay.slnf_w, nx ? P + ln + lay.sln1_b : P + lay.slnf_b, nx ? P + ln + lay.sln1_s : P + lay.slnf_s,
                     101 + 2 * l));
    } else {
      {  // htmp = output_linear(cat) + h   (transformer.py:86); block 0 adds the broadcast embedding
        VgGemmProb p = mk(cat, E, Pb + lo + lay.wo, E, R, E, E);
        p.C = htmp; p.ldc = E; p.bias = P + lo + lay.bo;
        if (l == 0) { p.resf = P + lay.emb; p.res_period = T; } else { p.res = h; p.ldr = E; }
        set_drop(p, dr, 100 + 2 * l, 0);  // attention_dropout(msha(...)) + h, transformer.py:86
        VG_TRY(vg_gemm_launch(&p, 1, VG_NT, st));
      }
      VG_TRY(vg_sln_fwd_launch(htmp, 0, w.wmod, P + lo + lay.sln2_w, P + lo + lay.sln2_b, P + lo + lay.sln2_s, P + lo + lay.sln2_s + 1,
                               s2, w.mean2 + (size_t)l * R, w.rstd2 + (size_t)l * R, R, E, 1e-5f, st));
      VG_TRY(lin_fwd(s2, E, Pb + lo + lay.wm, P + lo + lay.bm, w.hout + (size_t)l * RE, R, E, VG_ACT_NONE, 0.f, htmp, nullptr, nullptr, st,
                     &dr, 101 + 2 * l));  // Sequential(Linear, Dropout) + htmp, muilti_layer_perceptron.py:26-28
    }
  }
  const bf16* hL = w.hout + (size_t)(d.L - 1) * RE;
  if (!rown)
    VG_TRY(vg_sln_fwd_launch(hL, 0, w.wmod, P + lay.slnf_w, P + lay.slnf_b, P + lay.slnf_s, P + lay.slnf_s + 1, w.sf, w.meanf, w.rstdf,
                             R, E, 1e-5f, st));
  if (net->pos_table) VG_TRY(vg_add_table_launch(w.sf, net->pos_table, R, E, T, st));  // constant: the backward is unchanged
  VG_TRY(lin_fwd(w.sf, E, Pb + lay.s1_w, P + lay.s1_b, w.y1, R, d.O, VG_ACT_SIN, d.omega0, nullptr, nullptr, w.zf1, st));
  bf16* rows = d.patch > 0 ? w.y2 : (bf16*)img;
  VG_TRY(lin_fwd(w.y1, d.O, Pb + lay.s2_w, P + lay.s2_b, rows, R, d.CW, VG_ACT_SIN, d.omega0, nullptr, nullptr, w.zf2, st));
  if (d.patch > 0) VG_TRY(vg_unpatchify_launch(rows, (bf16*)img, B, d.C, d.IH, d.patch, st));  // token rows -> NCHW
  return 0;
}

// Backward stages: 0 = SIREN output layers + final SLN, 1..L = blocks L-1 .. 0, L+1 = learned embedding + mapping Linear.
// After a call returning stages up to s (1 <= s <= L) the gradients of blocks >= L-s and of everything behind the blocks
// (final SLN, SIREN) - a contiguous tail of the flat buffer from layer0 + (L-s)*layer_stride - are final.
extern "C" int vg_gen_backward_stages(const VgGenNet* net, int B, void* ws, const void* d_img, int stage_begin, int stage_end,
                                      void* stream) {
  if (!net || !ws || !d_img || !net->G || B < 1) return -1;
  if (stage_begin < 0 || stage_end > net->d.L + 2 || stage_begin >= stage_end) return -2;
  const VgGenDims& d = net->d;
  VgGenLayout lay;
  VG_TRY(vg_gen_layout(&d, &lay));
  hipStream_t st = (hipStream_t)stream;
  const int E = d.E, T = d.T, R = B * T, HE = E / d.H, PW = 3 * E + 64;
  GenWs w; carve_gen(d, B, ws, w);
  const float* P = net->P; const bf16* Pb = (const bf16*)net->Pb; float* G = net->G;
  const size_t RE = (size_t)R * E;
  const float scale = 1.0f / sqrtf((float)E);
  const int rown = gen_row_nwg(d, R);  // full-row path: the forward packed the weights into this workspace
  const int parts = rown ? rown : vg_ln_bwd_nparts(R);
  const long long pb = gen_pack_block(d), po_wqkvT = 2LL * E * E, po_wmT = 5LL * E * E;
  const size_t part_sz = (size_t)vg_ln_bwd_nparts(R) * PW;  // slot stride in the workspace (sized for the standalone kernels)
  VgFoldJobs folds; folds.n = 0;
  const Drop dr = mk_drop(net->dropout_p, net->dropout_seed, net->dropout_step);
  const bool drop = dr.thr != 0;
  // g masked for the MLP-branch dropout it meets next / gmid masked for the attention-branch dropout: one copy PER BLOCK (with the block's
  // dqkv), so that the weight gradients of two blocks - which read them - can wait for each other and go out as ONE grouped launch
  // (half the launches, folds and slab traffic; the discriminator's pairs, second half of round 3).  Without dropout the unmasked
  // rotating buffers are the operands and every block launches its own.
  auto gm2_of = [&](int l) { return drop ? w.pgm2 + (size_t)l * R * E : w.gm[0]; };
  auto gm1_of = [&](int l) { return drop ? w.pgm1 + (size_t)l * R * E : w.gm[1]; };
  auto dqkv_of = [&](int l) { return drop ? w.pdqkv + (size_t)l * R * 3 * E : w.dqkv; };

  // dh = gres + SLN'(A W) in one kernel
  auto row_bwd = [&](const bf16* A, int K, const bf16* Wp, const bf16* hx, int hbc, const float* mean, const float* rstd, const float* lw,
                     const float* lb, const float* sc, const bf16* gres, bf16* dh, bf16* dhm, int accumulate, float* part, int site) -> int {
    VgRowArgs ra = {};
    ra.N = E;
    ra.A = A; ra.lda = K; ra.Wp = Wp; ra.M = R; ra.K = K; ra.x = hx; ra.x_period = hbc; ra.mean = mean; ra.rstd = rstd; ra.gamma = lw; ra.lbias = lb;
    ra.gs = sc; ra.bs = sc + 1; ra.wmod = w.wmod; ra.gres = gres; ra.dx = dh; ra.dxm = dhm; ra.dw_acc = w.dw_acc; ra.dw_accumulate = accumulate;
    ra.part = part;
    if (dhm) { ra.drop_thresh = dr.thr; ra.drop_key = site_key(dr, site); ra.drop_scale = dr.scale; ra.drop_step = dr.step; }
    const int r = vg_gemm_row_launch(ra, VG_ROW_LNBWD, st);
    return r > 0 ? 0 : (r < 0 ? -r : -3);
  };
  bf16 *g = w.g[0], *gmid = w.g[1], *gin = w.g[2];
  if (stage_begin == 0) {
  // SIREN output layers (siren.py:44-45): y = sin(w0 z)  ->  dz = dy * w0 cos(w0 z)
  const bf16* d_rows = (const bf16*)d_img;
  if (d.patch > 0) {  // NCHW gradient -> token rows, the adjoint of the forward scatter
    VG_TRY(vg_patchify_launch(d_img, 1, w.dy2, B, d.C, d.IH, d.patch, st));
    d_rows = w.dy2;
  }
  VG_TRY(vg_sin_grad_launch(d_rows, w.zf2, w.dz2, (long long)R * d.CW, d.omega0, st));
  // bias gradients of the two SIREN layers = column sums of the weight gradients' dY operands: they ride along in those GEMMs (ones x dY on the
  // MFMA pipe, one row per K slice - two 11 us column-sum launches less) and are folded with the SLN partials at the end of this call
  {
    int splits = pick_splits(tiles128(d.CW, d.O), R, GEN_SPLIT_CAP);
    if (splits > vg_colsum_bf16_nparts(R)) splits = vg_colsum_bf16_nparts(R);  // (part_cs2 holds that many rows)
    VgGemmProb p = wg(w.dz2, d.CW, w.y1, d.O, R, w.slab, (long long)d.CW * d.O, splits);
    p.colsum = w.part_cs2; p.colsum_split_stride = d.CW;
    VG_TRY(vg_gemm_launch(&p, 1, VG_TN, st));
    VG_TRY(vg_slab_reduce_launch(w.slab, (long long)d.CW * d.O, p.splits, G + lay.s2_w, (long long)d.CW * d.O, 1, st));
    VG_TRY(vg_fold_push(folds, w.part_cs2, p.splits, d.CW, G + lay.s2_b, d.CW, nullptr, 0, nullptr, 0, nullptr, 0));
  }
  VG_TRY(lin_dgrad(w.dz2, Pb + lay.s2_w, w.dz1, R, d.CW, d.O, VG_ACT_MUL_COS, nullptr, w.zf1, d.omega0, st));
  {
    int splits = pick_splits(tiles128(d.O, E), R, GEN_SPLIT_CAP);
    if (splits > vg_colsum_bf16_nparts(R)) splits = vg_colsum_bf16_nparts(R);
    VgGemmProb p = wg(w.dz1, d.O, w.sf, E, R, w.slab, (long long)d.O * E, splits);
    p.colsum = w.part_cs; p.colsum_split_stride = d.O;
    VG_TRY(vg_gemm_launch(&p, 1, VG_TN, st));
    VG_TRY(vg_slab_reduce_launch(w.slab, (long long)d.O * E, p.splits, G + lay.s1_w, (long long)d.O * E, 1, st));
    VG_TRY(vg_fold_push(folds, w.part_cs, p.splits, d.O, G + lay.s1_b, d.O, nullptr, 0, nullptr, 0, nullptr, 0));
  }
  const bf16* hL = w.hout + (size_t)(d.L - 1) * RE;
  if (rown) {  // first SIREN layer's input gradient + the final SLN's backward
    VG_TRY(row_bwd(w.dz1, d.O, w.wpack + (long long)d.L * pb, hL, 0, w.meanf, w.rstdf, P + lay.slnf_w, P + lay.slnf_b, P + lay.slnf_s, nullptr, g,
                   drop ? gm2_of(d.L - 1) : nullptr, 0, w.part + (size_t)(2 * d.L) * part_sz, 101 + 2 * (d.L - 1)));
  } else {
    VG_TRY(lin_dgrad(w.dz1, Pb + lay.s1_w, w.ds, R, d.O, E, 0, nullptr, nullptr, 0.f, st));
    VG_TRY(vg_sln_bwd_launch(w.ds, hL, 0, w.wmod, w.meanf, w.rstdf, P + lay.slnf_w, P + lay.slnf_b, P + lay.slnf_s, P + lay.slnf_s + 1,
                             nullptr, g, w.dw_acc, 0, w.part + (size_t)(2 * d.L) * part_sz, R, E, drop ? gm2_of(d.L - 1) : nullptr, dr.thr, site_key(dr, 101 + 2 * (d.L - 1)), dr.scale, dr.step, st));
  }
  {
    const long long lo = lay.layer0 + (long long)(d.L - 1) * lay.layer_stride;
    VG_TRY(vg_fold_push(folds, w.part + (size_t)(2 * d.L) * part_sz, parts, PW, G + lay.slnf_w, E, G + lay.slnf_b, E, G + lo + lay.bm, E, G + lay.slnf_s, 2));
  }
  }  // stage 0
  int pend[2], npend = 0;
  auto gen_wgrad = [&](int la, int nb) -> int {  // blocks la, la - 1 (nb = 2) or la alone: grouped split-K launch + fold
    int splits = pick_splits(2 * (tiles128(3 * E, E) + 2 * tiles128(E, E)), R, GEN_SPLIT_CAP / 2);
    if (const int bn = wide_bn(E, E); bn && R % 32 == 0 && E % 128 == 0)
      splits = pick_splits384(2 * (tiles_wide(3 * E, E, bn) + 2 * tiles_wide(E, E, bn)), R, GEN_SPLIT_CAP / 2);
    VgGemmProb pr[6];
    for (int j = 0; j < nb; ++j) {
      const int lb = la - j;
      float* slab = w.slab + (size_t)j * splits * lay.layer_weights;
      pr[3 * j + 0] = wg(dqkv_of(lb), 3 * E, w.s1 + (size_t)lb * RE, E, R, slab + lay.wqkv, lay.layer_weights, splits);
      pr[3 * j + 1] = wg(gm1_of(lb), E, w.cat + (size_t)lb * RE, E, R, slab + lay.wo, lay.layer_weights, splits);
      pr[3 * j + 2] = wg(gm2_of(lb), E, w.s2 + (size_t)lb * RE, E, R, slab + lay.wm, lay.layer_weights, splits);
    }
    VG_TRY(vg_gemm_launch(pr, 3 * nb, VG_TN, st));
    const long long lo0 = lay.layer0 + (long long)la * lay.layer_stride;
    if (nb == 2)
      VG_TRY(vg_slab_reduce2_launch(w.slab, w.slab + (size_t)splits * lay.layer_weights, lay.layer_weights, pr[0].splits, G + lo0, G + lo0 - lay.layer_stride,
                                    lay.layer_weights, 1, st));
    else
      VG_TRY(vg_slab_reduce_launch(w.slab, lay.layer_weights, pr[0].splits, G + lo0, lay.layer_weights, 1, st));
    return 0;
  };
  for (int l = d.L - 1; l >= 0; --l) {
    const int stage = d.L - l;
    if (stage >= stage_end) break;
    if (stage < stage_begin) { bf16* t = g; g = gin; gin = t; continue; }  // the buffers rotate once per block already done
    const long long lo = lay.layer0 + (long long)l * lay.layer_stride;
    const bf16* h = (l == 0) ? Pb + lay.emb : w.hout + (size_t)(l - 1) * RE;
    const int hb = (l == 0) ? T : 0;
    const bf16* s1 = w.s1 + (size_t)l * RE;
    const bf16* qkv = w.qkv + (size_t)l * RE * 3;
    const bf16* cat = w.cat + (size_t)l * RE;
    const bf16* htmp = w.htmp + (size_t)l * RE;
    const bf16* s2 = w.s2 + (size_t)l * RE;
    // hout = drop(mlp(s2)) + htmp  (transformer.py:87; MLP is a single Linear, muilti_layer_perceptron.py:37-42)
    bf16* const gm2buf = gm2_of(l);
    bf16* const gm1buf = gm1_of(l);
    bf16* const dqkv_l = dqkv_of(l);
    const bf16* gb2 = drop ? gm2buf : g;
    const bf16* wp = w.wpack + (size_t)l * pb;
    if (rown) {  // block MLP input gradient + SLN2 backward + the residual-stream gradient
      VG_TRY(row_bwd(gb2, E, wp + po_wmT, htmp, 0, w.mean2 + (size_t)l * R, w.rstd2 + (size_t)l * R, P + lo + lay.sln2_w, P + lo + lay.sln2_b,
                     P + lo + lay.sln2_s, g, gmid, drop ? gm1buf : nullptr, 1, w.part + (size_t)(2 * l) * part_sz, 100 + 2 * l));
    } else {
      VG_TRY(lin_dgrad(gb2, Pb + lo + lay.wm, w.ds, R, E, E, 0, nullptr, nullptr, 0.f, st));
      VG_TRY(vg_sln_bwd_launch(w.ds, htmp, 0, w.wmod, w.mean2 + (size_t)l * R, w.rstd2 + (size_t)l * R, P + lo + lay.sln2_w,
                               P + lo + lay.sln2_b, P + lo + lay.sln2_s, P + lo + lay.sln2_s + 1, g, gmid, w.dw_acc, 1, w.part + (size_t)(2 * l) * part_sz, R, E,
                               drop ? gm1buf : nullptr, dr.thr, site_key(dr, 100 + 2 * l), dr.scale, dr.step, st));
    }
    const bf16* gb1 = drop ? gm1buf : gmid;
    VG_TRY(vg_fold_push(folds, w.part + (size_t)(2 * l) * part_sz, parts, PW, G + lo + lay.sln2_w, E, G + lo + lay.sln2_b, E, G + lo + lay.bo, E,
                 G + lo + lay.sln2_s, 2));
    VG_TRY(lin_dgrad(gb1, Pb + lo + lay.wo, w.dcat, R, E, E, 0, nullptr, nullptr, 0.f, st));
    VG_TRY(vg_attn_bwd_launch(qkv, cat, w.dcat, w.lse + (size_t)l * B * d.H * T, dqkv_l, B, d.H, T, HE, scale, 0, st));
    if (!rown) VG_TRY(lin_dgrad(dqkv_l, Pb + lo + lay.wqkv, w.ds, R, 3 * E, E, 0, nullptr, nullptr, 0.f, st));
    if (!drop) {  // the operands are the rotating buffers: this block's weight gradients now
      const long long tiles = tiles128(3 * E, E) + 2 * tiles128(E, E);
      int splits = pick_splits(tiles, R, GEN_SPLIT_CAP);
      if (const int bn = wide_bn(E, E); bn && R % 32 == 0 && E % 128 == 0)
        splits = pick_splits384(tiles_wide(3 * E, E, bn) + 2 * tiles_wide(E, E, bn), R, GEN_SPLIT_CAP);
      VgGemmProb pr[3];
      pr[0] = wg(dqkv_l, 3 * E, s1, E, R, w.slab + lay.wqkv, lay.layer_weights, splits);
      pr[1] = wg(gb1, E, cat, E, R, w.slab + lay.wo, lay.layer_weights, splits);
      pr[2] = wg(gb2, E, s2, E, R, w.slab + lay.wm, lay.layer_weights, splits);
      VG_TRY(vg_gemm_launch(pr, 3, VG_TN, st));
      VG_TRY(vg_slab_reduce_launch(w.slab, lay.layer_weights, pr[0].splits, G + lo, lay.layer_weights, 1, st));
    } else {  // per-block operands: two blocks per launch (the odd one out of a call goes alone, with the SAME K partition: every schedule adds the same slices)
      pend[npend++] = l;
      if (npend == 2) { VG_TRY(gen_wgrad(pend[0], 2)); npend = 0; }
    }
    if (rown) {  // q|k|v input gradient + SLN1 backward + the residual-stream gradient
      VG_TRY(row_bwd(dqkv_l, 3 * E, wp + po_wqkvT, h, hb, w.mean1 + (size_t)l * R, w.rstd1 + (size_t)l * R, P + lo + lay.sln1_w, P + lo + lay.sln1_b,
                     P + lo + lay.sln1_s, gmid, gin, (drop && l > 0) ? gm2_of(l - 1) : nullptr, 1, w.part + (size_t)(2 * l + 1) * part_sz, 101 + 2 * (l - 1)));
    } else {
      VG_TRY(vg_sln_bwd_launch(w.ds, h, hb, w.wmod, w.mean1 + (size_t)l * R, w.rstd1 + (size_t)l * R, P + lo + lay.sln1_w,
                               P + lo + lay.sln1_b, P + lo + lay.sln1_s, P + lo + lay.sln1_s + 1, gmid, gin, w.dw_acc, 1, w.part + (size_t)(2 * l + 1) * part_sz, R, E,
                               (drop && l > 0) ? gm2_of(l - 1) : nullptr, dr.thr, site_key(dr, 101 + 2 * (l - 1)), dr.scale, dr.step, st));
    }
    float* bm_prev = (l > 0) ? G + (lo - lay.layer_stride) + lay.bm : nullptr;
    VG_TRY(vg_fold_push(folds, w.part + (size_t)(2 * l + 1) * part_sz, parts, PW, G + lo + lay.sln1_w, E, G + lo + lay.sln1_b, E, bm_prev, E, G + lo + lay.sln1_s, 2));
    bf16* t = g; g = gin; gin = t;
  }
  if (npend == 1) VG_TRY(gen_wgrad(pend[0], 1));  // the odd block out of this call
  VG_TRY(vg_colsum_f32_multi_launch(folds, st));  // all SLN partial sums queued by this call in one launch
  if (stage_end < d.L + 2) return 0;
  // learned embedding (generator.py:24-26,62) is broadcast over the batch: its gradient is the batch sum
  VG_TRY(vg_batch_sum_launch(g, w.emb_sum, B, T, E, st));
  VG_TRY(vg_slab_reduce_launch(w.emb_sum, 0, 1, G + lay.emb, (long long)T * E, 1, st));
  // mapping Linear: d W = d w^T z ; d b = colsum(d w)   (d w accumulated in fp32 over the 2L+1 SLN uses)
  VG_TRY(vg_colsum_f32_launch(w.dw_acc, B, T * E, G + lay.map_b, T * E, nullptr, 0, nullptr, 0, nullptr, 0, 1, st));
  VG_TRY(vg_cast_f32_bf16_launch(w.dw_acc, w.dwb, (long long)R * E, st));
  {
    // K = B rows only: one K slice, accumulated straight into the gradient buffer (a 50 MB slab and its fold pass saved)
    VgGemmProb p = wg(w.dwb, T * E, w.zb, d.Z, B, G + lay.map_w, 0, 1);
    p.cf_accumulate = 1;
    VG_TRY(vg_gemm_launch(&p, 1, VG_TN, st));
  }
  return 0;
}

extern "C" int vg_gen_backward(const VgGenNet* net, int B, void* ws, const void* d_img, void* stream) {
  if (!net) return -1;
  return vg_gen_backward_stages(net, B, ws, d_img, 0, net->d.L + 2, stream);
}
