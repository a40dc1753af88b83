// extern "C" single-operator entry points of include/vitgan_hip.h (thin argument adapters).
#include "../../include/vitgan_hip.h"
#include "vg_kernels.h"
#include "vg_row.h"
#include "vg_chain.h"

extern "C" int vg_abi_version(void) { return VG_ABI_VERSION; }

extern "C" int vg_linear_fwd(const void* A, const void* W, const float* bias, const void* res, void* C, void* pre_bf16,
                             float* pre_f32, int M, int N, int K, int act, float act_scale, void* stream) {
  if (!A || !W || (!C && !pre_bf16 && !pre_f32)) return -1;
  if (act < 0 || act > 3) return -4;
  VgGemmProb p = vg_gemm_prob();
  p.A = (const bf16*)A; p.lda = K; p.B = (const bf16*)W; p.ldb = K; p.M = M; p.N = N; p.K = K;
  p.C = (bf16*)C; p.ldc = N; p.bias = bias; p.res = (const bf16*)res; p.ldr = N; p.C2 = (bf16*)pre_bf16; p.ldc2 = N;
  if (pre_f32) { p.pre_f32 = 1; p.Cf = pre_f32; p.ldcf = N; }
  p.act = act; p.act_scale = act_scale;
  return vg_gemm_launch(&p, 1, VG_NT, (hipStream_t)stream);
}
extern "C" int vg_linear_gelu_fwd(const void* A, const void* W, const float* bias, void* C, void* dcode, int M, int N, int K, void* stream) {
  if (!A || !W || !C || !dcode) return -1;
  VgGemmProb p = vg_gemm_prob();
  p.A = (const bf16*)A; p.lda = K; p.B = (const bf16*)W; p.ldb = K; p.M = M; p.N = N; p.K = K;
  p.C = (bf16*)C; p.ldc = N; p.bias = bias; p.C2 = (bf16*)dcode; p.ldc2 = N; p.c2_gelu_grad = 2; p.act = VG_ACT_GELU;
  return vg_gemm_launch(&p, 1, VG_NT, (hipStream_t)stream);
}
extern "C" int vg_linear_dgrad(const void* dY, const void* W, void* dX, int M, int N, int K, int mul_mode, const void* Z,
                               const float* Zf, float act_scale, void* stream) {
  if (!dY || !W || !dX) return -1;
  if (mul_mode != 0 && mul_mode != VG_ACT_MUL_GELU_GRAD && mul_mode != VG_ACT_MUL_COS && mul_mode != VG_ACT_MUL_TANH_GRAD && mul_mode != VG_ACT_MUL_Z && mul_mode != VG_ACT_MUL_Z8) return -4;
  if (((mul_mode == VG_ACT_MUL_GELU_GRAD || mul_mode == VG_ACT_MUL_TANH_GRAD || mul_mode == VG_ACT_MUL_Z || mul_mode == VG_ACT_MUL_Z8) && !Z) || (mul_mode == VG_ACT_MUL_COS && !Zf)) return -1;
  VgGemmProb p = vg_gemm_prob();
  p.A = (const bf16*)dY; p.lda = N; p.B = (const bf16*)W; p.ldb = K; p.M = M; p.N = K; p.K = N;
  p.C = (bf16*)dX; p.ldc = K; p.act = mul_mode; p.act_scale = act_scale;
  p.Z = (const bf16*)Z; p.ldz = K; p.Zf = Zf; p.ldzf = K;
  return vg_gemm_launch(&p, 1, VG_NN, (hipStream_t)stream);
}
extern "C" long long vg_linear_wgrad_slab_floats(int N, int K, int splits) {
  if (N < 1 || K < 1 || splits < 1 || splits > VG_WGRAD_MAX_SPLITS) return -2;
  return (long long)splits * N * K;
}
extern "C" int vg_linear_wgrad(const void* dY, const void* X, float* dW, float* slab_ws, long long slab_floats, int M, int N, int K,
                               int splits, int accumulate, void* stream) {
  if (!dY || !X || !dW || !slab_ws) return -1;
  const long long need = vg_linear_wgrad_slab_floats(N, K, splits);
  if (need < 0 || slab_floats < need || M < 1) return -2;  // the K slices are written at slab_ws + s*N*K: never past the caller's buffer
  VgGemmProb p = vg_gemm_prob();
  p.A = (const bf16*)dY; p.lda = N; p.B = (const bf16*)X; p.ldb = K; p.M = N; p.N = K; p.K = M;
  p.Cf = slab_ws; p.ldcf = K; p.cf_split_stride = (long long)N * K; p.splits = splits;
  VG_TRY(vg_gemm_launch(&p, 1, VG_TN, (hipStream_t)stream));
  return vg_slab_reduce_launch(slab_ws, (long long)N * K, p.splits, dW, (long long)N * K, accumulate, (hipStream_t)stream);
}
// Several weight gradients that share their row count as ONE grouped split-K launch and ONE fold: problem j writes its K slices at
// slab_ws + off[j] (+ s * region_floats per slice), the regions [off[j], off[j] + N[j] K[j]) must tile [0, region_floats) exactly -
// the layout of a block's weights in the flat gradient buffer - and dst[0 .. region_floats) (+)= the folded slices.
extern "C" int vg_linear_wgrad_group(int n, const void* const* dY, const void* const* X, const int* N, const int* K, const long long* off,
                                     int M, int splits, float* slab_ws, long long slab_floats, float* dst, long long region_floats,
                                     int accumulate, void* stream) {
  if (n < 1 || n > 8 || !dY || !X || !N || !K || !off || !slab_ws || !dst) return -1;
  if (M < 1 || splits < 1 || splits > VG_WGRAD_MAX_SPLITS || region_floats < 1) return -2;
  if (slab_floats < (long long)splits * region_floats) return -2;
  int order[8];
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 1; i < n; ++i)  // insertion sort by offset
    for (int j = i; j > 0 && off[order[j]] < off[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  long long at = 0;
  for (int i = 0; i < n; ++i) {
    const int j = order[i];
    if (!dY[j] || !X[j]) return -1;
    if (N[j] < 1 || K[j] < 1 || off[j] != at) return -2;  // a gap would fold uninitialised slab memory into dst, an overlap would race
    at += (long long)N[j] * K[j];
  }
  if (at != region_floats) return -2;
  VgGemmProb pr[8];
  for (int j = 0; j < n; ++j) {
    VgGemmProb p = vg_gemm_prob();
    p.A = (const bf16*)dY[j]; p.lda = N[j]; p.B = (const bf16*)X[j]; p.ldb = K[j]; p.M = N[j]; p.N = K[j]; p.K = M;
    p.Cf = slab_ws + off[j]; p.ldcf = K[j]; p.cf_split_stride = region_floats; p.splits = splits;
    pr[j] = p;
  }
  VG_TRY(vg_gemm_launch(pr, n, VG_TN, (hipStream_t)stream));
  return vg_slab_reduce_launch(slab_ws, region_floats, pr[0].splits, dst, region_floats, accumulate, (hipStream_t)stream);
}
extern "C" int vg_layernorm_fwd(const void* x, long long xs, const float* gamma, const float* beta, void* y, long long ys,
                                float* mean, float* rstd, int R, int E, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd) return -1;
  return vg_ln_fwd_launch((const bf16*)x, xs, gamma, beta, (bf16*)y, ys, mean, rstd, R, E, eps, (hipStream_t)stream);
}
extern "C" int vg_layernorm_bwd_parts(int R) { return vg_ln_bwd_nparts(R); }
extern "C" int vg_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                const void* gres, void* dx, float* part, int R, int E, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !part) return -1;
  return vg_ln_bwd_launch((const bf16*)dy, (const bf16*)x, mean, rstd, gamma, (const bf16*)gres, (bf16*)dx, part, R, E,
                          nullptr, 0, 0, 1.f, nullptr, (hipStream_t)stream);
}
extern "C" int vg_sln_fwd(const void* h, int h_bcast_rows, const void* w, const float* lw, const float* lb, const float* gs,
                          const float* bs, void* y, float* mean, float* rstd, int R, int E, float eps, void* stream) {
  if (!h || !w || !lw || !lb || !gs || !bs || !y || !mean || !rstd) return -1;
  return vg_sln_fwd_launch((const bf16*)h, h_bcast_rows, (const bf16*)w, lw, lb, gs, bs, (bf16*)y, mean, rstd, R, E, eps,
                           (hipStream_t)stream);
}
extern "C" int vg_sln_bwd(const void* dy, const void* h, int h_bcast_rows, const void* w, const float* mean, const float* rstd,
                          const float* lw, const float* lb, const float* gs, const float* bs, const void* gres, void* dh,
                          float* dw_acc, int dw_accumulate, float* part, int R, int E, void* stream) {
  if (!dy || !h || !w || !mean || !rstd || !lw || !lb || !gs || !bs || !dh || !dw_acc || !part) return -1;
  return vg_sln_bwd_launch((const bf16*)dy, (const bf16*)h, h_bcast_rows, (const bf16*)w, mean, rstd, lw, lb, gs, bs,
                           (const bf16*)gres, (bf16*)dh, dw_acc, dw_accumulate, part, R, E, nullptr, 0, 0, 1.f, nullptr, (hipStream_t)stream);
}
extern "C" int vg_colsum_f32(const float* part, int rows, int width, float* d0, int n0, float* d1, int n1, float* d2, int n2,
                             float* d3, int n3, int accumulate, void* stream) {
  if (!part || rows < 1 || width < 1) return -1;
  return vg_colsum_f32_launch(part, rows, width, d0, n0, d1, n1, d2, n2, d3, n3, accumulate, (hipStream_t)stream);
}
extern "C" int vg_colsum_bf16_parts(int R) { return vg_colsum_bf16_nparts(R); }
extern "C" int vg_colsum_bf16(const void* X, long long ld, int R, int N, float* part_ws, float* dst, int accumulate, void* stream) {
  if (!X || !part_ws || !dst) return -1;
  return vg_colsum_bf16_launch((const bf16*)X, ld, R, N, part_ws, dst, accumulate, (hipStream_t)stream);
}
extern "C" int vg_dropout_apply(const void* x, void* y, long long n, float p, unsigned long long seed, int site,
                                const unsigned* step_dev, void* stream) {
  if (!x || !y || n < 1 || p < 0.f || p >= 1.f) return -1;
  int t = (int)lrintf(p * 256.f); if (t > 255) t = 255;
  return vg_dropout_apply_launch((const bf16*)x, (bf16*)y, n, (unsigned)t, vg_site_key(seed, site), t ? 256.f / (256.f - t) : 1.f,
                                 step_dev, (hipStream_t)stream);
}
// ---- full-row Linear + LayerNorm (gemm_row.hip) ----
extern "C" long long vg_row_pack_elems_e(int E, int K) { return (!vg_row_width_ok(E) || K < 32 || (K & 31)) ? -2 : (long long)E * K; }
extern "C" long long vg_row_pack_elems(int K) { return vg_row_pack_elems_e(VG_ROW_N, K); }
extern "C" int vg_row_pack_weight_e(int E, const void* W, int ld, int K, int transposed, void* Wp, void* stream) {
  if (!W || !Wp) return -1;
  if (!vg_row_width_ok(E)) return -3;
  VgPackJobs pj;
  pj.N = E;
  pj.src = (const bf16*)W; pj.dst = (bf16*)Wp; pj.src_stride = 0; pj.dst_stride = 0; pj.nblocks = 1; pj.n = 1;
  pj.d[0] = {0, 0, K, ld, transposed ? 1 : 0};
  return vg_pack_rows_launch(pj, (hipStream_t)stream);
}
extern "C" int vg_row_pack_weight(const void* W, int ld, int K, int transposed, void* Wp, void* stream) {
  return vg_row_pack_weight_e(VG_ROW_N, W, ld, K, transposed, Wp, stream);
}
extern "C" int vg_row_parts(int M) { return vg_row_nwg(M); }
static void row_drop(VgRowArgs& ra, float p, unsigned long long seed, int site, const unsigned* step_dev) {
  int t = (int)lrintf(p * 256.f); if (t < 0) t = 0; if (t > 255) t = 255;
  if (!t) return;
  ra.drop_thresh = (unsigned)t; ra.drop_key = vg_site_key(seed, site); ra.drop_scale = 256.f / (256.f - t); ra.drop_step = step_dev;
}
extern "C" int vg_linear_ln_fwd_e(int E, const void* A, const void* Wp, const float* bias, const void* res, void* Y, void* Yn, float* mean,
                                float* rstd, const float* gamma, const float* beta, int M, int K, float eps, float drop_p,
                                unsigned long long seed, int site, const unsigned* step_dev, void* stream) {
  if (!vg_row_width_ok(E)) return -3;
  if (!A || !Wp || !Y || (Yn && (!mean || !rstd || !gamma || !beta)) || drop_p < 0.f || drop_p >= 1.f) return -1;
  VgRowArgs ra = {};
  ra.N = E;
  ra.A = (const bf16*)A; ra.lda = K; ra.Wp = (const bf16*)Wp; ra.M = M; ra.K = K; ra.bias = bias; ra.res = (const bf16*)res;
  ra.Y = (bf16*)Y; ra.Yn = (bf16*)Yn; ra.mean_out = mean; ra.rstd_out = rstd; ra.gamma = gamma; ra.beta = beta; ra.eps = eps;
  row_drop(ra, drop_p, seed, site, step_dev);
  const int r = vg_gemm_row_launch(ra, VG_ROW_LNFWD, (hipStream_t)stream);
  return r > 0 ? 0 : (r < 0 ? -r : -3);
}
extern "C" int vg_linear_ln_fwd(const void* A, const void* Wp, const float* bias, const void* res, void* Y, void* Yn, float* mean,
                                float* rstd, const float* gamma, const float* beta, int M, int K, float eps, float drop_p,
                                unsigned long long seed, int site, const unsigned* step_dev, void* stream) {
  return vg_linear_ln_fwd_e(VG_ROW_N, A, Wp, bias, res, Y, Yn, mean, rstd, gamma, beta, M, K, eps, drop_p, seed, site, step_dev, stream);
}
extern "C" long long vg_encoder_mlp_image_elems(void) { return (long long)VG_CH_MLP_STAGES * VG_CH_STAGE / 2; }
extern "C" int vg_encoder_mlp_pack(const void* W1, const void* W2, void* img, void* stream) {
  return vg_chain_mlp_pack_launch((const bf16*)W1, (const bf16*)W2, (bf16*)img, VG_CH_KNAT, (hipStream_t)stream);
}
extern "C" int vg_encoder_mlp_fwd(const void* xn, const void* img, const float* b1, const float* b2, const void* res, void* a1,
                                  void* dcode, void* Y, void* Yn, float* mean, float* rstd, const float* gamma, const float* beta,
                                  int M, float eps, float drop_p, unsigned long long seed, int site, const unsigned* step_dev,
                                  void* stream) {
  if (!xn || !img || !b1 || !b2 || !a1 || !dcode || !Y || (Yn && (!mean || !rstd || !gamma || !beta)) || drop_p < 0.f || drop_p >= 1.f) return -1;
  VgChainMlpArgs ca = {};
  ca.xn = (const bf16*)xn; ca.ldx = VG_CH_E; ca.img = (const bf16*)img; ca.b1 = b1; ca.b2 = b2; ca.res = (const bf16*)res;
  ca.a1 = (bf16*)a1; ca.z8 = (unsigned char*)dcode; ca.Y = (bf16*)Y; ca.Yn = (bf16*)Yn; ca.mean_out = mean; ca.rstd_out = rstd;
  ca.gamma = gamma; ca.beta = beta; ca.eps = eps; ca.M = M;
  int t = (int)lrintf(drop_p * 256.f); if (t < 0) t = 0; if (t > 255) t = 255;
  if (t) { ca.drop_thresh = (unsigned)t; ca.drop_key = vg_site_key(seed, site); ca.drop_scale = 256.f / (256.f - t); ca.drop_step = step_dev; }
  const int r = vg_chain_mlp_fwd_launch(ca, (hipStream_t)stream);
  return r > 0 ? 0 : (r < 0 ? -r : -3);
}
extern "C" long long vg_encoder_post_attention_image_elems(void) { return (long long)(VG_CH_MLP_STAGES + VG_CH_FRONT_STAGES) * VG_CH_STAGE / 2; }
extern "C" int vg_encoder_post_attention_pack(const void* Wo, const void* W1, const void* W2, void* img, void* stream) {
  return vg_chain_block_pack_launch((const bf16*)Wo, (const bf16*)W1, (const bf16*)W2, (bf16*)img, (hipStream_t)stream);
}
extern "C" int vg_encoder_post_attention_fwd(const void* ao, const void* x, const void* img, const float* bo, const float* b1, const float* b2,
                                             const float* gamma2, const float* beta2, const float* gamma, const float* beta, void* xmid, void* xn2,
                                             float* mean2, float* rstd2, void* a1, void* dcode, void* Y, void* Yn, float* mean, float* rstd, int M,
                                             float eps, float drop_p, unsigned long long seed, int site_attn, int site_mlp,
                                             const unsigned* step_dev, void* stream) {
  if (!ao || !x || !img || !b1 || !b2 || !gamma2 || !beta2 || !xmid || !xn2 || !mean2 || !rstd2 || !a1 || !dcode || !Y ||
      (Yn && (!mean || !rstd || !gamma || !beta)) || drop_p < 0.f || drop_p >= 1.f)
    return -1;
  VgChainMlpArgs ca = {};
  ca.ao = (const bf16*)ao; ca.xin = (const bf16*)x; ca.ldx = VG_CH_E; ca.img = (const bf16*)img; ca.bo = bo; ca.b1 = b1; ca.b2 = b2;
  ca.gamma2 = gamma2; ca.beta2 = beta2; ca.xmid = (bf16*)xmid; ca.xn_out = (bf16*)xn2; ca.mean2 = mean2; ca.rstd2 = rstd2;
  ca.a1 = (bf16*)a1; ca.z8 = (unsigned char*)dcode; ca.Y = (bf16*)Y; ca.Yn = (bf16*)Yn; ca.mean_out = mean; ca.rstd_out = rstd;
  ca.gamma = gamma; ca.beta = beta; ca.eps = eps; ca.M = M;
  int t = (int)lrintf(drop_p * 256.f); if (t < 0) t = 0; if (t > 255) t = 255;
  if (t) {
    ca.drop_thresh = (unsigned)t; ca.drop_key = vg_site_key(seed, site_mlp); ca.drop_key_a = vg_site_key(seed, site_attn);
    ca.drop_scale = 256.f / (256.f - t); ca.drop_step = step_dev;
  }
  const int r = vg_chain_mlp_fwd_launch(ca, (hipStream_t)stream);
  return r > 0 ? 0 : (r < 0 ? -r : -3);
}
extern "C" int vg_linear_sln_fwd_e(int E, const void* A, const void* Wp, const float* bias, const void* res, const float* resf, int res_period,
                                 void* Y, void* Yn, float* mean, float* rstd, const void* wmod, const float* lw, const float* lb,
                                 const float* gs, const float* bs, int M, int K, float eps, float drop_p, unsigned long long seed,
                                 int site, const unsigned* step_dev, void* stream) {
  if (!vg_row_width_ok(E)) return -3;
  if (!A || !Wp || !Y || !Yn || !mean || !rstd || !wmod || !lw || !lb || !gs || !bs || drop_p < 0.f || drop_p >= 1.f) return -1;
  if (res && resf) return -1;
  VgRowArgs ra = {};
  ra.N = E;
  ra.A = (const bf16*)A; ra.lda = K; ra.Wp = (const bf16*)Wp; ra.M = M; ra.K = K; ra.bias = bias; ra.res = (const bf16*)res;
  ra.resf = resf; ra.res_period = res_period;
  ra.Y = (bf16*)Y; ra.Yn = (bf16*)Yn; ra.mean_out = mean; ra.rstd_out = rstd; ra.gamma = lw; ra.beta = lb; ra.eps = eps;
  ra.wmod = (const bf16*)wmod; ra.gs = gs; ra.bs = bs;
  row_drop(ra, drop_p, seed, site, step_dev);
  const int r = vg_gemm_row_launch(ra, VG_ROW_LNFWD, (hipStream_t)stream);
  return r > 0 ? 0 : (r < 0 ? -r : -3);
}
extern "C" int vg_linear_sln_fwd(const void* A, const void* Wp, const float* bias, const void* res, const float* resf, int res_period,
                                 void* Y, void* Yn, float* mean, float* rstd, const void* wmod, const float* lw, const float* lb,
                                 const float* gs, const float* bs, int M, int K, float eps, float drop_p, unsigned long long seed,
                                 int site, const unsigned* step_dev, void* stream) {
  return vg_linear_sln_fwd_e(VG_ROW_N, A, Wp, bias, res, resf, res_period, Y, Yn, mean, rstd, wmod, lw, lb, gs, bs, M, K, eps, drop_p, seed, site, step_dev, stream);
}
extern "C" int vg_linear_dgrad_sln_bwd_e(int E, const void* dY, const void* WpT, const void* h, int h_bcast_rows, const void* wmod,
                                       const float* mean, const float* rstd, const float* lw, const float* lb, const float* gs,
                                       const float* bs, const void* gres, void* dh, void* dhm, float* dw_acc, int dw_accumulate,
                                       float* part, int M, int K, float drop_p, unsigned long long seed, int site,
                                       const unsigned* step_dev, void* stream) {
  if (!vg_row_width_ok(E)) return -3;
  if (!dY || !WpT || !h || !wmod || !mean || !rstd || !lw || !lb || !gs || !bs || !dh || !dw_acc || !part || drop_p < 0.f || drop_p >= 1.f)
    return -1;
  VgRowArgs ra = {};
  ra.N = E;
  ra.A = (const bf16*)dY; ra.lda = K; ra.Wp = (const bf16*)WpT; ra.M = M; ra.K = K; ra.x = (const bf16*)h; ra.x_period = h_bcast_rows;
  ra.mean = mean; ra.rstd = rstd; ra.gamma = lw; ra.lbias = lb; ra.gs = gs; ra.bs = bs; ra.wmod = (const bf16*)wmod;
  ra.gres = (const bf16*)gres; ra.dx = (bf16*)dh; ra.dxm = (bf16*)dhm; ra.dw_acc = dw_acc; ra.dw_accumulate = dw_accumulate; ra.part = part;
  if (dhm) row_drop(ra, drop_p, seed, site, step_dev);
  if (dhm && !ra.drop_thresh) { ra.drop_thresh = 0; ra.drop_scale = 1.f; }
  const int r = vg_gemm_row_launch(ra, VG_ROW_LNBWD, (hipStream_t)stream);
  return r > 0 ? 0 : (r < 0 ? -r : -3);
}
extern "C" int vg_linear_dgrad_sln_bwd(const void* dY, const void* WpT, const void* h, int h_bcast_rows, const void* wmod,
                                       const float* mean, const float* rstd, const float* lw, const float* lb, const float* gs,
                                       const float* bs, const void* gres, void* dh, void* dhm, float* dw_acc, int dw_accumulate,
                                       float* part, int M, int K, float drop_p, unsigned long long seed, int site,
                                       const unsigned* step_dev, void* stream) {
  return vg_linear_dgrad_sln_bwd_e(VG_ROW_N, dY, WpT, h, h_bcast_rows, wmod, mean, rstd, lw, lb, gs, bs, gres, dh, dhm, dw_acc, dw_accumulate, part, M, K, drop_p, seed, site, step_dev, stream);
}
extern "C" int vg_linear_dgrad_ln_bwd_e(int E, const void* dY, const void* WpT, const void* x, const float* mean, const float* rstd,
                                      const float* gamma, const void* gres, void* dx, void* dxm, float* part, int M, int K,
                                      float drop_p, unsigned long long seed, int site, const unsigned* step_dev, void* stream) {
  if (!vg_row_width_ok(E)) return -3;
  if (!dY || !WpT || !x || !mean || !rstd || !gamma || !dx || !part || drop_p < 0.f || drop_p >= 1.f) return -1;
  VgRowArgs ra = {};
  ra.N = E;
  ra.A = (const bf16*)dY; ra.lda = K; ra.Wp = (const bf16*)WpT; ra.M = M; ra.K = K; ra.x = (const bf16*)x; ra.mean = mean; ra.rstd = rstd;
  ra.gamma = gamma; ra.gres = (const bf16*)gres; ra.dx = (bf16*)dx; ra.dxm = (bf16*)dxm; ra.part = part;
  if (dxm) row_drop(ra, drop_p, seed, site, step_dev);
  if (dxm && !ra.drop_thresh) { ra.drop_thresh = 0; ra.drop_scale = 1.f; }
  const int r = vg_gemm_row_launch(ra, VG_ROW_LNBWD, (hipStream_t)stream);
  return r > 0 ? 0 : (r < 0 ? -r : -3);
}
extern "C" int vg_linear_dgrad_ln_bwd(const void* dY, const void* WpT, const void* x, const float* mean, const float* rstd,
                                      const float* gamma, const void* gres, void* dx, void* dxm, float* part, int M, int K,
                                      float drop_p, unsigned long long seed, int site, const unsigned* step_dev, void* stream) {
  return vg_linear_dgrad_ln_bwd_e(VG_ROW_N, dY, WpT, x, mean, rstd, gamma, gres, dx, dxm, part, M, K, drop_p, seed, site, step_dev, stream);
}
extern "C" int vg_attention_fwd(const void* qkv, void* out, float* lse, int B, int H, int S, int HE, float scale, void* stream) {
  if (!qkv || !out || !lse) return -1;
  return vg_attn_fwd_launch((const bf16*)qkv, (bf16*)out, lse, B, H, S, HE, scale, 0, (hipStream_t)stream);
}
extern "C" int vg_attention_cls_fwd(const void* qkv, void* out_cls, float* lse_cls, int B, int H, int S, int HE, float scale, void* stream) {
  if (!qkv || !out_cls || !lse_cls) return -1;
  return vg_attn_cls_fwd_launch((const bf16*)qkv, (bf16*)out_cls, lse_cls, B, H, S, HE, scale, (hipStream_t)stream);
}
extern "C" int vg_attention_cls_bwd(const void* qkv, const void* out_cls, const void* d_out_cls, const float* lse_cls, void* d_qkv, int B, int H,
                                    int S, int HE, float scale, void* stream) {
  if (!qkv || !out_cls || !d_out_cls || !lse_cls || !d_qkv) return -1;
  return vg_attn_cls_bwd_launch((const bf16*)qkv, (const bf16*)out_cls, (const bf16*)d_out_cls, lse_cls, (bf16*)d_qkv, B, H, S, HE, scale,
                                (hipStream_t)stream);
}
extern "C" int vg_attention_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, void* d_qkv, int B, int H,
                                int S, int HE, float scale, void* stream) {
  if (!qkv || !out || !d_out || !lse || !d_qkv) return -1;
  return vg_attn_bwd_launch((const bf16*)qkv, (const bf16*)out, (const bf16*)d_out, lse, (bf16*)d_qkv, B, H, S, HE, scale, 0,
                            (hipStream_t)stream);
}
extern "C" int vg_attention_fp8_fwd(const void* qkv, void* out, float* lse, int B, int H, int S, int HE, float scale, void* stream) {
  if (!qkv || !out || !lse) return -1;
  return vg_attn_fwd_launch((const bf16*)qkv, (bf16*)out, lse, B, H, S, HE, scale, 2, (hipStream_t)stream);
}
extern "C" int vg_attention_fp8_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, void* d_qkv, int B, int H,
                                    int S, int HE, float scale, void* stream) {
  if (!qkv || !out || !d_out || !lse || !d_qkv) return -1;
  return vg_attn_bwd_launch((const bf16*)qkv, (const bf16*)out, (const bf16*)d_out, lse, (bf16*)d_qkv, B, H, S, HE, scale, 2,
                            (hipStream_t)stream);
}
extern "C" int vg_attention_l2_fwd(const void* qkv, void* out, float* lse, int B, int H, int S, int HE, float scale, void* stream) {
  if (!qkv || !out || !lse) return -1;
  return vg_attn_fwd_launch((const bf16*)qkv, (bf16*)out, lse, B, H, S, HE, scale, 1, (hipStream_t)stream);
}
extern "C" int vg_attention_l2_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, void* d_qkv, int B, int H,
                                   int S, int HE, float scale, void* stream) {
  if (!qkv || !out || !d_out || !lse || !d_qkv) return -1;
  return vg_attn_bwd_launch((const bf16*)qkv, (const bf16*)out, (const bf16*)d_out, lse, (bf16*)d_qkv, B, H, S, HE, scale, 1,
                            (hipStream_t)stream);
}
extern "C" int vg_unfold_tokens_fwd(const void* img, int img_is_bf16, void* tokens, int B, int C, int IH, int P, int overlap, void* stream) {
  if (!img || !tokens || B < 1 || C < 1) return -1;
  return vg_unfold_tokens_launch(img, img_is_bf16, (bf16*)tokens, B, C, IH, P, overlap, (hipStream_t)stream);
}
extern "C" int vg_unfold_tokens_bwd(const void* d_tokens, void* d_img, int B, int C, int IH, int P, int overlap, void* stream) {
  if (!d_tokens || !d_img || B < 1 || C < 1) return -1;
  return vg_unfold_tokens_bwd_launch((const bf16*)d_tokens, (bf16*)d_img, B, C, IH, P, overlap, (hipStream_t)stream);
}
extern "C" int vg_act_fwd(const void* h, void* y, long long n, int act, void* stream) {
  if (!h || !y || n < 1) return -1;
  return vg_act2_launch((const bf16*)h, nullptr, nullptr, (bf16*)y, nullptr, n, act, 0, (hipStream_t)stream);
}
extern "C" int vg_act_bwd(const void* dy, const void* h, void* dh, long long n, int act, void* stream) {
  if (!dy || !h || !dh || n < 1) return -1;
  return vg_act2_launch((const bf16*)h, (const bf16*)dy, nullptr, (bf16*)dh, nullptr, n, act, 1, (hipStream_t)stream);
}
extern "C" int vg_act_bwd_bwd(const void* u, const void* dy, const void* h, void* d_dy, void* d_h, long long n, int act, void* stream) {
  if (!u || !dy || !h || !d_dy || !d_h || n < 1) return -1;
  return vg_act2_launch((const bf16*)h, (const bf16*)dy, (const bf16*)u, (bf16*)d_dy, (bf16*)d_h, n, act, 2, (hipStream_t)stream);
}
extern "C" int vg_layernorm_bwd_bwd_parts(int R) { return vg_ln_bwd_bwd_nparts(R); }
extern "C" int vg_layernorm_bwd_bwd(const void* u, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                    void* d_dy, void* d_x, float* part, int R, int E, void* stream) {
  if (!u || !dy || !x || !mean || !rstd || !gamma || !d_dy || !d_x || !part) return -1;
  return vg_ln_bwd_bwd_launch((const bf16*)u, (const bf16*)dy, (const bf16*)x, mean, rstd, gamma, (bf16*)d_dy, (bf16*)d_x, part, R, E,
                              (hipStream_t)stream);
}
extern "C" int vg_attention_bwd_bwd(const void* qkv, const void* d_out, const float* lse, const void* u_qkv, void* d_d_out, void* d_qkv2,
                                    int B, int H, int S, int HE, float scale, void* stream) {
  if (!qkv || !d_out || !lse || !u_qkv || !d_d_out || !d_qkv2) return -1;
  return vg_attn_bwd_bwd_launch((const bf16*)qkv, (const bf16*)d_out, lse, (const bf16*)u_qkv, (bf16*)d_d_out, (bf16*)d_qkv2, B, H, S, HE,
                                scale, (hipStream_t)stream);
}
extern "C" int vg_gan_loss(const float* logits, float* dlogits, float* loss_out, int n, int kind, int role, float grad_scale,
                           void* stream) {
  if (!logits || !dlogits || !loss_out || n < 1) return -1;
  return vg_gan_loss_launch(logits, dlogits, loss_out, n, kind, role, grad_scale, (hipStream_t)stream);
}
extern "C" int vg_zero_tick(float* g, long long n, int* step_dev, void* stream) {
  if (!g) return -1;
  return vg_zero_tick_launch(g, n, step_dev, (hipStream_t)stream);
}
extern "C" int vg_step_inputs(const float* real, void* imgs_bf16, long long n_img, float* z, long long n_z, unsigned long long seed,
                              const int* step_dev, void* stream) {
  if ((!real && !z) || (real && !imgs_bf16)) return -1;
  return vg_step_inputs_launch(real, (bf16*)imgs_bf16, n_img, z, n_z, seed, step_dev, (hipStream_t)stream);
}
extern "C" int vg_gan_loss_pair(const float* logits, float* dlogits, float* loss_out, int n0, int role0, int n1, int role1, int kind,
                                float grad_scale, void* stream) {
  if (!logits || !dlogits || !loss_out) return -1;
  return vg_gan_loss_pair_launch(logits, dlogits, loss_out, n0, role0, n1, role1, kind, grad_scale, (hipStream_t)stream);
}
extern "C" int vg_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, long long n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, const int* step_dev, float gscale,
                             void* stream) {
  if (!p || !g || !m || !v || !shadow_bf16 || n < 1 || (step < 1 && !step_dev)) return -1;
  return vg_adamw_launch(p, g, m, v, (bf16*)shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, step, step_dev, gscale,
                         (hipStream_t)stream);
}
extern "C" int vg_diversity_loss(const void* images, void* d_images, float* loss_out, float* scratch, int B, int D, float weight,
                                void* stream) {
  if (!images || !loss_out || !scratch) return -1;
  return vg_diversity_launch((const bf16*)images, (bf16*)d_images, loss_out, scratch, B, D, weight, (hipStream_t)stream);
}
extern "C" int vg_grad_clip(float* g, long long n, float gscale, float max_norm, float* scratch, void* stream) {
  if (!g || !scratch || n < 1 || !(max_norm > 0.f)) return -1;
  return vg_grad_clip_launch(g, n, gscale, max_norm, scratch, (hipStream_t)stream);
}
extern "C" int vg_cast_f32_bf16(const float* src, void* dst_bf16, long long n, void* stream) {
  if (!src || !dst_bf16 || n < 1) return -1;
  return vg_cast_f32_bf16_launch(src, (bf16*)dst_bf16, n, (hipStream_t)stream);
}
