// Internal C++ launch API shared by the engine and the C ABI (all functions only enqueue work).
#pragma once
#include "vg_common.h"
#include "vg_gemm.h"

int vg_attn_fwd_launch(const bf16* qkv, bf16* o, float* lse, int B, int H, int S, int HE, float scale, int mode, hipStream_t st);  // mode 0 dot, 1 L2 scores, 2 fp8 operands
int vg_attn_bwd_launch(const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse, bf16* dqkv, int B, int H,
                       int S, int HE, float scale, int mode, hipStream_t st);

int vg_ln_fwd_launch(const bf16* x, long long xs, const float* gamma, const float* beta, bf16* y, long long ys,
                     float* mean, float* rstd, int R, int E, float eps, hipStream_t st);
int vg_sln_fwd_launch(const bf16* h, int h_bcast_rows, const bf16* wmod, const float* lw, const float* lb,
                      const float* gs, const float* bs, bf16* y, float* mean, float* rstd, int R, int E, float eps,
                      hipStream_t st);
int vg_ln_bwd_nparts(int R);
int vg_ln_bwd_launch(const bf16* dy, const bf16* x, const float* mean, const float* rstd, const float* gamma,
                     const bf16* gres, bf16* dx, float* part, int R, int E, bf16* dxm, unsigned dthr, unsigned dkey,
                     float dscale, const unsigned* dstep, hipStream_t st, int x_row_step = 1, int drop_row_mul = 1);  // drop_row_mul: dxm's mask is that of row r * drop_row_mul of the full tensor
int vg_sln_bwd_launch(const bf16* dy, const bf16* h, int h_bcast_rows, const bf16* wmod, const float* mean,
                      const float* rstd, const float* lw, const float* lb, const float* gs, const float* bs,
                      const bf16* gres, bf16* dh, float* dw_acc, int dw_accumulate, float* part, int R, int E,
                      bf16* dhm, unsigned dthr, unsigned dkey, float dscale, const unsigned* dstep, hipStream_t st);
int vg_colsum_f32_launch(const float* part, int rows, int width, float* d0, int n0, float* d1, int n1, float* d2, int n2,
                         float* d3, int n3, int accumulate, hipStream_t st);
#include "vg_fold.h"
int vg_colsum_f32_multi_launch(const VgFoldJobs& jobs, hipStream_t st);
int vg_colsum_bf16_nparts(int R);
int vg_colsum_bf16_part_launch(const bf16* X, long long ld, int R, int N, float* part, hipStream_t st);
int vg_colsum_bf16_launch(const bf16* X, long long ld, int R, int N, float* part, float* dst, int accumulate,
                          hipStream_t st);

int vg_patchify_launch(const void* img, int img_is_bf16, bf16* A, int B, int C, int IH, int P, hipStream_t st);
int vg_unpatchify_launch(const bf16* dA, bf16* dimg, int B, int C, int IH, int P, hipStream_t st);
int vg_unfold_tokens_launch(const void* img, int img_is_bf16, bf16* out, int B, int C, int IH, int P, int overlap, hipStream_t st);
int vg_unfold_tokens_bwd_launch(const bf16* dout, bf16* dimg, int B, int C, int IH, int P, int overlap, hipStream_t st);
int vg_fill_cls_launch(bf16* x, const float* cls, int B, int S, int E, unsigned dthr, unsigned dkey, float dscale, const unsigned* dstep,
                       hipStream_t st);
int vg_dropout_apply_launch(const bf16* x, bf16* y, long long n, unsigned dthr, unsigned dkey, float dscale, const unsigned* dstep,
                            hipStream_t st);
int vg_take_rows_launch(const bf16* in, bf16* out, int B, int S, int first, int n_take, int E, hipStream_t st);
// gm (nullable): the same rows times the dropout mask of site key dkey over the [B*S, E] buffer (what vg_dropout_apply would make of g)
int vg_scatter_cls_launch(const bf16* src, bf16* g, int B, int S, int E, hipStream_t st, bf16* gm = nullptr, unsigned dthr = 0, unsigned dkey = 0,
                          float dscale = 1.f, const unsigned* dstep = nullptr);
// two compact [B, E] sources into the CLS rows of two zero-filled [B*S, E] tensors, one launch (the pruned tail of the top encoder block)
int vg_scatter_cls2_launch(const bf16* src_a, bf16* dst_a, const bf16* src_b, bf16* dst_b, int B, int S, int E, hipStream_t st);
int vg_batch_sum_launch(const bf16* g, float* out, int B, int S, int E, hipStream_t st);
int vg_embed_small_grads_launch(const float* tok_sum, float* d_cls, float* d_pos, float* d_bias, int S, int E, hipStream_t st);
int vg_head_fc2_launch(const bf16* t, const float* W2, const float* b2, float* logits, int B, int E, int Kc, hipStream_t st);
// one launch (returns 1: dW2 / db1 / db2 left as partial rows in `part` for a deferred fold) or the separate kernels (returns 0); < 0 error
int vg_head_bwd_parts(int B);
int vg_head_bwd_part_width(int E, int Kc);
int vg_head_bwd_launch(const float* dlog, const float* W2, const bf16* t, bf16* dz, float* dW2, float* db2, int B, int E, int Kc,
                       int want_wgrad, hipStream_t st, float* part = nullptr);
// start of a step: zero_grad + step counter; input cast + latent noise (elementwise.hip)
int vg_zero_tick_launch(float* g, long long n, int* step, hipStream_t st);
int vg_step_inputs_launch(const float* real, bf16* imgs, long long n_img, float* z, long long n_z, unsigned long long seed, const int* step,
                          hipStream_t st);
// two segments of one logit vector in one launch: [0, n0) with role0 -> loss_out[0], [n0, n0 + n1) with role1 -> loss_out[1]
int vg_gan_loss_pair_launch(const float* logit, float* dlog, float* loss_out, int n0, int role0, int n1, int role1, int kind, float grad_scale,
                            hipStream_t st);
int vg_gan_loss_launch(const float* logit, float* dlog, float* loss_out, int n, int kind, int role, float grad_scale,
                       hipStream_t st);
int vg_adamw_launch(float* p, const float* g, float* m, float* v, bf16* shadow, long long n, float lr, float b1, float b2,
                    float eps, float wd, int step, const int* step_dev, float gscale, hipStream_t st);
int vg_cast_f32_bf16_launch(const float* src, bf16* dst, long long n, hipStream_t st);
int vg_slab_reduce_launch(const float* slab, long long stride, int nslab, float* dst, long long n, int accumulate, hipStream_t st);
// two folds of the same shape in one launch (the two blocks of a paired weight-gradient launch)
int vg_slab_reduce2_launch(const float* slab0, const float* slab1, long long stride, int nslab, float* dst0, float* dst1, long long n, int accumulate,
                           hipStream_t st);
int vg_sin_grad_launch(const bf16* dy, const float* z, bf16* dz, long long n, float w0, hipStream_t st);

// dropout key of (seed, site): splitmix64 folded to 32 bits (the engine's site numbering: include/vitgan_hip.h, vg_dropout_apply)
static inline unsigned vg_site_key(unsigned long long seed, int site) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(site + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
  return (unsigned)(z ^ (z >> 32));
}

#define VG_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != 0) return _rc;   \
  } while (0)
// second-order operators (second_order.hip): the backward of the backward operators, for the gradient penalty
int vg_act2_launch(const bf16* h, const bf16* dy, const bf16* u, bf16* o0, bf16* o1, long long n, int mode, int what, hipStream_t st);
int vg_ln_bwd_bwd_nparts(int R);
int vg_ln_bwd_bwd_launch(const bf16* u, const bf16* dy, const bf16* x, const float* mean, const float* rstd, const float* gamma,
                         bf16* d_dy, bf16* d_x, float* part, int R, int E, hipStream_t st);
// the top block's attention as the classifier sees it: the CLS query only (attention.hip); o_cls / do_cls [B, E], lse_cls [B, H]
int vg_attn_cls_fwd_launch(const bf16* qkv, bf16* o_cls, float* lse_cls, int B, int H, int S, int HE, float scale, hipStream_t st);
int vg_attn_cls_bwd_launch(const bf16* qkv, const bf16* o_cls, const bf16* do_cls, const float* lse_cls, bf16* dqkv, int B, int H, int S, int HE,
                           float scale, hipStream_t st);
int vg_attn_bwd_bwd_mfma_launch(const bf16* qkv, const bf16* d_o, const float* lse, const bf16* uqkv, bf16* d_do, bf16* d_qkv, int B, int H,
                                int S, int HE, float scale, hipStream_t st);  // attention.hip
int vg_attn_bwd_bwd_launch(const bf16* qkv, const bf16* d_o, const float* lse, const bf16* uqkv, bf16* d_do, bf16* d_qkv, int B, int H,
                           int S, int HE, float scale, hipStream_t st);
int vg_grad_clip_launch(float* g, long long n, float gscale, float max_norm, float* scratch, hipStream_t st);
int vg_add_table_launch(bf16* x, const float* table, long long rows, int E, int period, hipStream_t st);
int vg_diversity_launch(const bf16* x, bf16* d_img, float* loss_out, float* scratch, int B, int D, float weight, hipStream_t st);
// the gradient penalty as one C call (vg_vit_penalty): its small kernels (second_order.hip)
int vg_pen_interp_launch(const bf16* real, const bf16* fake, const float* eps, float* out, int B, long long per, hipStream_t st);
int vg_pen_norm_launch(const bf16* g, bf16* u, float* pen_img, float* pen_out, int B, long long per, float weight, hipStream_t st);
int vg_pen_head2_launch(const bf16* u, const bf16* t, const float* W2, bf16* u_gt, bf16* s_p, int B, int E, int Kc, hipStream_t st);
int vg_add_bf16_launch(const bf16* a, const bf16* b, bf16* out, long long n, hipStream_t st);
int vg_fill_f32_launch(float* p, long long n, float v, hipStream_t st);
