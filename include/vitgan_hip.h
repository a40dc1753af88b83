/*
 * vitgan_hip.h - C ABI of libvitgan_hip.so, the MI355X (gfx950) compute library under the
 * ViTGAN nn.Module surface.
 *
 * The reference (krzkro4122/vit-gan) is pure Python on PyTorch ATen ops and has no FFI / operator
 * plugin interface of its own; its boundary for the hot path is the nn.Module surface
 * of src/v2/modules.py plus the step body of src/v2/training.py:170-211.  Each entry point below
 * names the reference code whose ATen call sequence it replaces.  The Python side binds these with
 * ctypes (vit-gan_amd/_lib.py); INTEGRATION.md shows the binding a maintainer of the reference adds.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory unless it says "host";
 *   - `bf16` tensors are passed as `void*` (2-byte IEEE bfloat16, row-major);
 *   - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it (no host
 *     synchronisation, no allocation), so callers may capture a sequence of calls in a hipGraph;
 *   - return value: 0 on success, a positive hipError_t, or a negative argument-validation code
 *     (-1 bad count, -2 bad size, -3 unsupported shape/alignment, -4 bad mode).  Nothing is
 *     launched when the return value is negative.
 *   - ownership: the caller owns all memory; the library keeps no state between calls.
 */
#ifndef VITGAN_HIP_H
#define VITGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VG_ABI_VERSION 9
int vg_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Single operators (parity-tested one by one; tests/test_ops_gpu.py)
 * ---------------------------------------------------------------------------------------- */

/* nn.Linear forward (+ fused epilogue):  C[M,N] = act(A[M,K] @ W[N,K]^T + bias) + res
 * replaces F.linear / nn.GELU / residual add of src/v2/modules.py:128-139,161,179-182 and
 * sin(30*Linear) of src/v1/siren.py:44-45.
 * act: 0 none, 1 GELU(erf), 2 sin(act_scale*v), 3 tanh.  bias fp32 [N] or NULL; res bf16 [M,N]
 * or NULL; pre_bf16 (bf16 [M,N]) / pre_f32 (fp32 [M,N]) receive the pre-activation or NULL. */
int vg_linear_fwd(const void* A, const void* W, const float* bias, const void* res, void* C,
                  void* pre_bf16, float* pre_f32, int M, int N, int K, int act, float act_scale,
                  void* stream);
/* fc1 of the encoder MLP as the engine runs it (src/v2/modules.py:128-139: Linear -> nn.GELU):  C = gelu(A W^T + bias) in bf16,
 * and dcode[M,N] = gelu'(A W^T + bias) as ONE BYTE per element: code = round(200 g) + 27 (27 <-> 0 and 227 <-> 1 exactly,
 * |error| <= 0.0025 over gelu's derivative range [-0.129, 1.129]) - all the backward keeps of the pre-activation.  N % 8 == 0. */
int vg_linear_gelu_fwd(const void* A, const void* W, const float* bias, void* C, void* dcode, int M, int N, int K,
                       void* stream);
/* input gradient of nn.Linear:  dX[M,K] = dY[M,N] @ W[N,K]   (autograd of F.linear)
 * mul_mode 0: none; 4: dX *= gelu'(Z) with Z bf16 [M,K]; 5: dX *= s*cos(s*Zf), Zf fp32 [M,K];
 * 6: dX *= 1 - Z^2 (Z = tanh output, bf16 [M,K]); 7: dX *= Z (Z = a derivative the forward stored, bf16 [M,K]:
 * 8: dX *= decode(Z), Z the byte codes [M,K] written by vg_linear_gelu_fwd - what the engine's fc2 input gradient uses). */
int vg_linear_dgrad(const void* dY, const void* W, void* dX, int M, int N, int K, int mul_mode,
                    const void* Z, const float* Zf, float act_scale, void* stream);
/* weight gradient of nn.Linear:  dW[N,K] (+)= dY[M,N]^T @ X[M,K], computed as `splits` slices of M
 * into fp32 slabs folded in a fixed order (deterministic).  slab_ws holds slab_floats floats and must be at
 * least vg_linear_wgrad_slab_floats(N, K, splits) = splits*N*K: a smaller workspace, or splits outside
 * [1, VG_WGRAD_MAX_SPLITS], returns -2 and launches nothing. */
#define VG_WGRAD_MAX_SPLITS 64
long long vg_linear_wgrad_slab_floats(int N, int K, int splits); /* host only; -2 for a bad argument */
int vg_linear_wgrad(const void* dY, const void* X, float* dW, float* slab_ws, long long slab_floats, int M,
                    int N, int K, int splits, int accumulate, void* stream);
/* n <= 8 weight gradients over the same M rows as ONE grouped split-K launch and ONE fold (ABI v7; what the gradient penalty's
 * double backward uses for a block's four Linears, src/v2/utils.py:124-144 through autograd of src/v2/modules.py:128-139,173-182):
 * problem j's slices go to slab_ws + off[j] + s*region_floats, the regions [off[j], off[j] + N[j]*K[j]) must tile
 * [0, region_floats) exactly (-2 otherwise: a gap would fold uninitialised memory), dst[0..region_floats) (+)= the fold. */
int vg_linear_wgrad_group(int n, const void* const* dY, const void* const* X, const int* N, const int* K, const long long* off,
                          int M, int splits, float* slab_ws, long long slab_floats, float* dst, long long region_floats,
                          int accumulate, void* stream);

/* nn.LayerNorm forward/backward (src/v2/modules.py:168,172,225; eps 1e-5, biased variance).
 * x,y bf16 [R,E] with row strides xs/ys (elements); mean/rstd fp32 [R]. E % 128 == 0, E <= 1024. */
int vg_layernorm_fwd(const void* x, long long xs, const float* gamma, const float* beta, void* y,
                     long long ys, float* mean, float* rstd, int R, int E, float eps, void* stream);
/* dx = (gres ? gres : 0) + LN'(dy).  part: fp32 [vg_layernorm_bwd_parts(R)][3E] scratch holding per
 * workgroup column sums (d gamma | d beta | colsum(dx)); fold with vg_colsum_f32. */
int vg_layernorm_bwd_parts(int R);
int vg_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                     const float* gamma, const void* gres, void* dx, float* part, int R, int E,
                     void* stream);
/* Self-modulated LayerNorm (src/v1/spectral_layer_norm.py:19-20): y = w*(gs*(LN(h)*lw+lb)+bs).
 * h_bcast_rows > 0: h has that many rows and is broadcast over the batch (generator.py:62). */
int vg_sln_fwd(const void* h, int h_bcast_rows, const void* w, const float* lw, const float* lb,
               const float* gs, const float* bs, void* y, float* mean, float* rstd, int R, int E,
               float eps, void* stream);
/* part: fp32 [parts][3E+64]: d lw | d lb | colsum(dh) | d gs, d bs.  dw_acc fp32 [R,E]. */
int vg_sln_bwd(const void* dy, const void* h, int h_bcast_rows, const void* w, const float* mean,
               const float* rstd, const float* lw, const float* lb, const float* gs, const float* bs,
               const void* gres, void* dh, float* dw_acc, int dw_accumulate, float* part, int R,
               int E, void* stream);
/* Linear + LayerNorm in ONE kernel, for the Linears whose output is the embedding width 384 (rows owned whole by a
 * workgroup, statistics in the epilogue; csrc/gemm_row.hip).  M % 16 == 0, K % 64 == 0, K >= 128; other shapes return -3
 * (vg_row_parts(M) == 0 says so beforehand) and the caller uses vg_linear_* + vg_layernorm_*.
 * The weight operand is PACKED first: Wp = vg_row_pack_weight(W): [K/32][384][32] stage images, vg_row_pack_elems(K) bf16;
 *   transposed = 0: W is [384, K] row-major with leading dimension ld (nn.Linear weight whose out_features is 384: forward);
 *   transposed = 1: W is [K, 384] (nn.Linear weight whose in_features is 384: its input gradient dX = dY W).
 * vg_linear_ln_fwd:  Y = res + drop(A Wp^T + bias);  Yn = LayerNorm(Y) * gamma + beta, mean / rstd of Y   (Yn NULL: Y only)
 *   replaces out_projection / fc2 + dropout + residual add + the LayerNorm that reads the sum (src/v2/modules.py:168,172,179-183).
 *   drop(.) is dropout site `site` of (drop_p, seed) as in vg_dropout_apply (drop_p = 0: none).
 * vg_linear_dgrad_ln_bwd:  dx = gres + LayerNorm'(dY Wp);  dxm = dx * mask(site) (NULL: none);
 *   part: fp32 [vg_row_parts(M)][3*384] per-workgroup column sums d gamma | d beta | colsum(dxm ? dxm : dx), fold with vg_colsum_f32
 *   replaces autograd of queries|keys|values / fc1 and of the LayerNorm in front of them (modules.py:178-181). */
long long vg_row_pack_elems(int K); /* host only */
int vg_row_pack_weight(const void* W, int ld, int K, int transposed, void* Wp, void* stream);
int vg_row_parts(int M);            /* host only */
int vg_linear_ln_fwd(const void* A, const void* Wp, const float* bias, const void* res, void* Y, void* Yn, float* mean,
                     float* rstd, const float* gamma, const float* beta, int M, int K, float eps, float drop_p,
                     unsigned long long seed, int site, const unsigned* step_dev, void* stream);
int vg_linear_dgrad_ln_bwd(const void* dY, const void* WpT, const void* x, const float* mean, const float* rstd,
                           const float* gamma, const void* gres, void* dx, void* dxm, float* part, int M, int K,
                           float drop_p, unsigned long long seed, int site, const unsigned* step_dev, void* stream);

/* The same entry points for another embedding width (round 4; E = 384 or 512 - C4's width; anything else returns -3, and
 * vg_row_pack_elems_e -2): every [M, 384] operand above is [M, E], packed images are [K/32][E][32], partial rows 3 E (+ 64) wide.
 * E = 768 (C5) does not fit the kernel's LDS ring (one 32-deep stage of W alone is 48 KiB) and stays on vg_linear_* + vg_layernorm_*. */
long long vg_row_pack_elems_e(int E, int K); /* host only */
int vg_row_pack_weight_e(int E, const void* W, int ld, int K, int transposed, void* Wp, void* stream);
int vg_linear_ln_fwd_e(int E, const void* A, const void* Wp, const float* bias, const void* res, void* Y, void* Yn, float* mean,
                       float* rstd, const float* gamma, const float* beta, int M, int K, float eps, float drop_p,
                       unsigned long long seed, int site, const unsigned* step_dev, void* stream);
int vg_linear_dgrad_ln_bwd_e(int E, const void* dY, const void* WpT, const void* x, const float* mean, const float* rstd,
                             const float* gamma, const void* gres, void* dx, void* dxm, float* part, int M, int K,
                             float drop_p, unsigned long long seed, int site, const unsigned* step_dev, void* stream);
int vg_linear_sln_fwd_e(int E, const void* A, const void* Wp, const float* bias, const void* res, const float* resf, int res_period,
                        void* Y, void* Yn, float* mean, float* rstd, const void* wmod, const float* lw, const float* lb,
                        const float* gs, const float* bs, int M, int K, float eps, float drop_p, unsigned long long seed,
                        int site, const unsigned* step_dev, void* stream);
int vg_linear_dgrad_sln_bwd_e(int E, const void* dY, const void* WpT, const void* h, int h_bcast_rows, const void* wmod,
                              const float* mean, const float* rstd, const float* lw, const float* lb, const float* gs,
                              const float* bs, const void* gres, void* dh, void* dhm, float* dw_acc, int dw_accumulate,
                              float* part, int M, int K, float drop_p, unsigned long long seed, int site,
                              const unsigned* step_dev, void* stream);

/* The MLP half of an encoder block as ONE launch (csrc/chain.hip; E = 384, hidden 768):
 *   a1 = gelu(xn W1^T + b1);  Y = res + drop(a1 W2^T + b2);  Yn = LayerNorm(Y) * gamma + beta, mean / rstd of Y   (Yn NULL: Y only)
 * replaces fc1 -> nn.GELU -> fc2 -> dropout -> residual add of src/v2/modules.py:181-182 and the LayerNorm that reads the sum
 * (the next block's norm1, :168 / :178).  A wave keeps its 16 rows in registers for the whole chain: the hidden a1 [M,768] and
 * dcode [M,768] (gelu' as byte codes, as vg_linear_gelu_fwd) are written for the backward and never read back.
 * The weights come as a chain image: img = vg_encoder_mlp_pack(W1 [768,384], W2 [384,768]), vg_encoder_mlp_image_elems() bf16.
 * M % 16 == 0; other shapes return -3 and the caller uses vg_linear_gelu_fwd + vg_linear_ln_fwd.  drop(.) as in vg_linear_ln_fwd. */
long long vg_encoder_mlp_image_elems(void); /* host only */
int vg_encoder_mlp_pack(const void* W1, const void* W2, void* img, void* stream);
int vg_encoder_mlp_fwd(const void* xn, const void* img, const float* b1, const float* b2, const void* res, void* a1, void* dcode,
                       void* Y, void* Yn, float* mean, float* rstd, const float* gamma, const float* beta, int M, float eps,
                       float drop_p, unsigned long long seed, int site, const unsigned* step_dev, void* stream);

/* Everything of an encoder block BETWEEN its attention and the next block's, but for the QKV projection, as ONE launch (csrc/chain.hip, the MLP
 * chain with the out-projection in front):
 *   x_mid = x + drop_a(ao Wo^T + bo);  xn2 = LayerNorm2(x_mid);  a1 = gelu(xn2 W1^T + b1);  Y = x_mid + drop_m(a1 W2^T + b2);  Yn = LayerNorm(Y)
 * replaces out_projection + dropout1 + residual, norm2, fc1, nn.GELU, fc2 + dropout2 + residual (src/v2/modules.py:179-182) and the next block's
 * norm1 (:168).  x_mid, xn2 (+ mean2 / rstd2), a1, dcode, Y, Yn (+ mean / rstd) are written for the backward; none is read back.
 * img = vg_encoder_post_attention_pack(Wo [384,384], W1 [768,384], W2 [384,768]).  M % 16 == 0.  An operator with its tests and measurements
 * (DESIGN.md s3); the engine does not call it. */
long long vg_encoder_post_attention_image_elems(void); /* host only */
int vg_encoder_post_attention_pack(const void* Wo, const void* W1, const void* W2, void* img, void* stream);
int vg_encoder_post_attention_fwd(const void* ao, const void* x, const void* img, const float* bo, const float* b1, const float* b2,
                                  const float* gamma2, const float* beta2, const float* gamma, const float* beta, void* xmid, void* xn2,
                                  float* mean2, float* rstd2, void* a1, void* dcode, void* Y, void* Yn, float* mean, float* rstd, int M,
                                  float eps, float drop_p, unsigned long long seed, int site_attn, int site_mlp,
                                  const unsigned* step_dev, void* stream);

/* The same two kernels with the v1 generator's self-modulated LayerNorm (src/v1/spectral_layer_norm.py:19-20) in the epilogue:
 * vg_linear_sln_fwd:  Y = (res | resf[row % res_period]) + drop(A Wp^T + bias);  Yn = w * (gs * (LN(Y) * lw + lb) + bs)
 *   replaces output_linear / the block MLP + dropout + residual + the SLN that reads the sum (src/v1/transformer.py:85-88);
 *   resf: fp32 [res_period, 384] residual broadcast over the batch (block 0: the learned embedding, generator.py:62), or NULL.
 * vg_linear_dgrad_sln_bwd:  dh = gres + LN'(dy_eff), dy_eff = (dY Wp) * w * gs;  dhm = dh * mask(site) (NULL: none);
 *   dw_acc (+)= (dY Wp) * (gs * (xhat * lw + lb) + bs)  (fp32 [M,384]);  part: fp32 [vg_row_parts(M)][3*384 + 64]:
 *   d lw | d lb | colsum(dhm ? dhm : dh) | d gs, d bs;  h_bcast_rows > 0: h has that many rows, broadcast over the batch. */
int vg_linear_sln_fwd(const void* A, const void* Wp, const float* bias, const void* res, const float* resf, int res_period,
                      void* Y, void* Yn, float* mean, float* rstd, const void* wmod, const float* lw, const float* lb,
                      const float* gs, const float* bs, int M, int K, float eps, float drop_p, unsigned long long seed,
                      int site, const unsigned* step_dev, void* stream);
int vg_linear_dgrad_sln_bwd(const void* dY, const void* WpT, const void* h, int h_bcast_rows, const void* wmod,
                            const float* mean, const float* rstd, const float* lw, const float* lb, const float* gs,
                            const float* bs, const void* gres, void* dh, void* dhm, float* dw_acc, int dw_accumulate,
                            float* part, int M, int K, float drop_p, unsigned long long seed, int site,
                            const unsigned* step_dev, void* stream);

/* dst_k[c] (+)= sum_r part[r][off_k + c] for up to 4 consecutive column segments (NULL = skip). */
int vg_colsum_f32(const float* part, int rows, int width, float* d0, int n0, float* d1, int n1,
                  float* d2, int n2, float* d3, int n3, int accumulate, void* stream);

/* dst[c] (+)= sum_r X[r][c] for a bf16 matrix (bias gradients: autograd of the "+ bias" in F.linear).
 * part_ws: vg_colsum_bf16_parts(R) * N floats of scratch.  Deterministic two-stage reduction. */
int vg_colsum_bf16_parts(int R);
int vg_colsum_bf16(const void* X, long long ld, int R, int N, float* part_ws, float* dst, int accumulate,
                   void* stream);

/* y = x * mask / keep for dropout site `site` of a network run with (p, seed): exactly the mask the fused
 * passes apply (element index = position in the row-major tensor; n % 4 == 0).  D sites: 0 embedding,
 * 1+2l attention branch, 2+2l MLP branch of block l; G sites: 100+2l, 101+2l.  Used by tests and by the
 * backward where no producer kernel can fuse the mask. */
int vg_dropout_apply(const void* x, void* y, long long n, float p, unsigned long long seed, int site,
                     const unsigned* step_dev, void* stream);

/* Fused multi-head self-attention (src/v2/modules.py:128-159 after the projections; src/v1/attention.py
 * :43-52,:97-101).  qkv bf16 [B*S, 3*H*HE] (Q | K | V thirds, head-major inside each third);
 * out bf16 [B*S, H*HE]; lse fp32 [B,H,S].  softmax(scale * q.k).  HE in {32,64,96}, S <= 80. */
int vg_attention_fwd(const void* qkv, void* out, float* lse, int B, int H, int S, int HE,
                     float scale, void* stream);
int vg_attention_bwd(const void* qkv, const void* out, const void* d_out, const float* lse,
                     void* d_qkv, int B, int H, int S, int HE, float scale, void* stream);
/* The same attention for ONE query per image, row 0 (ABI v7): what the classifier sees of the top encoder block
 * (src/v2/modules.py:195 reads the CLS row only).  out_cls / d_out_cls bf16 [B, H*HE] (the CLS rows, compact), lse_cls fp32
 * [B, H]; d_qkv is the full [B*S, 3*H*HE] gradient (dK, dV of every key, dQ zero off row 0) - exactly what vg_attention_bwd
 * writes when d_out is zero on every other row.  S <= 128. */
int vg_attention_cls_fwd(const void* qkv, void* out_cls, float* lse_cls, int B, int H, int S, int HE, float scale, void* stream);
int vg_attention_cls_bwd(const void* qkv, const void* out_cls, const void* d_out_cls, const float* lse_cls, void* d_qkv, int B, int H,
                         int S, int HE, float scale, void* stream);

/* The same fused attention with the v1 discriminator's L2-distance scores (src/v1/attention.py:43-52,66-67, lp = 2):
 * out = softmax(cdist(q, k) * scale) @ v - the Euclidean distance itself, as the reference has it.  Same layouts. */
/* The same fused attention with fp8 (OCP e4m3) MFMA operands for the activation products: Q.K^T in the forward and in the
 * backward's recompute (so P agrees with the forward's lse) and P.V (numerators scaled by 256 into e4m3's normal range);
 * the products with a gradient operand (dP, dV, dQ, dK) stay bf16.  Same layouts; outputs within ~2^-4 of the fp32 math. */
int vg_attention_fp8_fwd(const void* qkv, void* out, float* lse, int B, int H, int S, int HE,
                         float scale, void* stream);
int vg_attention_fp8_bwd(const void* qkv, const void* out, const void* d_out, const float* lse,
                         void* d_qkv, int B, int H, int S, int HE, float scale, void* stream);

int vg_attention_l2_fwd(const void* qkv, void* out, float* lse, int B, int H, int S, int HE,
                        float scale, void* stream);
int vg_attention_l2_bwd(const void* qkv, const void* out, const void* d_out, const float* lse,
                        void* d_qkv, int B, int H, int S, int HE, float scale, void* stream);

/* v1 overlapping-window tokeniser (src/v1/patch_encoder.py:20-27,54-73): windows of P + 2*overlap pixels at stride
 * (IH - P - 2*overlap)/P + 1, n x n of them; tokens bf16 [B, n*n, C*W*W] is the reference's FLAT view of the
 * (b, c, ty, tx, wy, wx) unfold order (no permute).  img fp32 or bf16 [B,C,IH,IH]; _bwd is the adjoint (d_img bf16). */
int vg_unfold_tokens_fwd(const void* img, int img_is_bf16, void* tokens, int B, int C, int IH, int P,
                         int overlap, void* stream);
int vg_unfold_tokens_bwd(const void* d_tokens, void* d_img, int B, int C, int IH, int P, int overlap,
                         void* stream);

/* ---- second-order operators: the backward OF the backward operators --------------------------------------------------
 * The gradient penalty of the reference's Wasserstein step (src/v2/utils.py:124-144, called at training.py:101-106)
 * differentiates ||d D(x)/d x|| with respect to D's parameters: every backward operator on the input-gradient path needs a
 * backward of its own.  For nn.Linear that is the GEMM family again (backward of dX = dY W:  d(dY) = ddX W^T = vg_linear_fwd,
 * dW = dY^T ddX = vg_linear_wgrad); the three below are the non-linear ones.  u = dL/d(output of the backward operator). */
/* elementwise activation on a stored bf16 pre-activation h (act 1 = GELU(erf), nn.GELU of modules.py:174; 3 = tanh, :191):
 * fwd y = f(h); bwd dh = dy f'(h); bwd_bwd: d_dy = u f'(h), d_h = u dy f''(h).  n % 4 == 0. */
int vg_act_fwd(const void* h, void* y, long long n, int act, void* stream);
int vg_act_bwd(const void* dy, const void* h, void* dh, long long n, int act, void* stream);
int vg_act_bwd_bwd(const void* u, const void* dy, const void* h, void* d_dy, void* d_h, long long n, int act,
                   void* stream);
/* backward of vg_layernorm_bwd (without residual): given u = dL/d(dx), writes d_dy = dL/d(dy) and d_x = dL/d(x) (bf16
 * [R,E]) and per-workgroup partial sums of dL/d(gamma) (part: fp32 [vg_layernorm_bwd_bwd_parts(R)][E], fold with
 * vg_colsum_f32).  E % 64 == 0, E <= 1024. */
int vg_layernorm_bwd_bwd_parts(int R);
int vg_layernorm_bwd_bwd(const void* u, const void* dy, const void* x, const float* mean, const float* rstd,
                         const float* gamma, void* d_dy, void* d_x, float* part, int R, int E, void* stream);
/* backward of vg_attention_bwd: given u_qkv = dL/d(d_qkv) ([B*S, 3E], same layout as qkv), writes d_d_out = dL/d(d_out)
 * [B*S, E] and d_qkv2 = dL/d(qkv) [B*S, 3E].  lse from the forward.  S <= 80 and 14 S HE + 16 S^2 + 8 S <= 160 KiB. */
int vg_attention_bwd_bwd(const void* qkv, const void* d_out, const float* lse, const void* u_qkv, void* d_d_out,
                         void* d_qkv2, int B, int H, int S, int HE, float scale, void* stream);

/* Start of a training step, two launches instead of torch's seven:
 * vg_zero_tick: g[0, n) = 0 (discriminator.zero_grad(), src/v2/training.py:177) and step_dev[0] += 1 (the device step counter that keys
 *   the dropout masks, AdamW's bias correction and the noise below); n % 4 == 0; step_dev may be NULL.
 * vg_step_inputs: imgs_bf16[0, n_img) = bf16(real) (real NULL: skipped) and z[0, n_z) ~ N(0, 1) (z NULL: skipped) - the latent batch of
 *   construct_noise (training.py:35-42 = torch.randn), counter-based on (seed, step_dev[0], index): Box-Muller of two hashed 24-bit
 *   uniforms, |z| <= 5.77; reproducible per (seed, step), fresh on every replay of a captured graph. */
int vg_zero_tick(float* g, long long n, int* step_dev, void* stream);
int vg_step_inputs(const float* real, void* imgs_bf16, long long n_img, float* z, long long n_z,
                   unsigned long long seed, const int* step_dev, void* stream);

/* GAN losses on logits (src/v1/gan.py:16-20,227,238,250 for kind 0; hinge for kind 1;
 * kind 2 = the Wasserstein critic losses of src/v2/training.py:72,97).
 * role 0 D-real, 1 D-fake, 2 G.  loss_out[0] = mean loss, dlogits = d loss / d logits * grad_scale. */
int vg_gan_loss(const float* logits, float* dlogits, float* loss_out, int n, int kind, int role,
                float grad_scale, void* stream);
/* the same on two consecutive segments of one logit vector in ONE launch (the fused real + fake discriminator pass):
 * logits[0, n0) with role0 -> loss_out[0], logits[n0, n0 + n1) with role1 -> loss_out[1]; each mean is over its own segment. */
int vg_gan_loss_pair(const float* logits, float* dlogits, float* loss_out, int n0, int role0, int n1, int role1,
                     int kind, float grad_scale, void* stream);

/* torch.optim.AdamW step over a flat fp32 buffer (src/v2/training.py:150-157), also refreshing the
 * bf16 shadow the GEMMs read.  n % 4 == 0.  grads are multiplied by gscale first.  The step number
 * (1-based) comes from step_dev[0] (device int) when step_dev != NULL - for hipGraph replay - else
 * from `step`. */
int vg_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, long long n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  const int* step_dev, float gscale, void* stream);
/* diversity_loss of the reference's unreached generator step (src/v2/utils.py:147-152; training.py:73-74 adds 0.1 x it
 * to the generator loss): loss_out[0] = sum_{i,j} |x_i - x_j|_1 / (B (B-1)) over images bf16 [B, D]; when d_images is
 * not NULL, d_images (bf16 [B, D]) += weight * d loss / d images.  scratch: ceil(D/16) floats.  B <= 1024. */
int vg_diversity_loss(const void* images, void* d_images, float* loss_out, float* scratch, int B, int D,
                      float weight, void* stream);

/* torch.nn.utils.clip_grad_norm_ (the reference's Wasserstein step, src/v2/training.py:78,104) on a flat fp32 gradient
 * buffer: g *= min(1, max_norm / (gscale*|g|_2 + 1e-6)) in place; scratch: 1 + 1024 floats of device memory,
 * scratch[0] receives gscale*|g|_2.  Deterministic (no atomics).  n % 4 == 0. */
int vg_grad_clip(float* g, long long n, float gscale, float max_norm, float* scratch, void* stream);
int vg_cast_f32_bf16(const float* src, void* dst_bf16, long long n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole-network passes (what the nn.Modules call: one C call per forward / backward)
 * ---------------------------------------------------------------------------------------- */

/* Execution context (second HIP stream + events) for concurrent weight-gradient work; host only.  One per
 * engine / process; the calls using it must come from one host thread. */
int vg_ctx_create(void** out_ctx);
int vg_ctx_destroy(void* ctx);

/* v2 VisionTransformer (src/v2/modules.py:202-238) = ViTDiscriminator.forward (:393-395). */
typedef struct VgVitDims {
  int C, IH, P, E, H, L, R, Kc; /* channels, image size, patch, embed, heads, layers, mlp ratio, classes */
} VgVitDims;
/* Flat parameter layout (element offsets, identical for fp32 master, bf16 shadow, fp32 grads). */
typedef struct VgVitLayout {
  long long conv_w, conv_b, pos, cls;           /* embedding.conv1.weight [E,C*P*P], .bias, pos_embedding, cls_token */
  long long layer0, layer_stride;               /* first encoder block, distance between blocks */
  long long wqkv, wo, w1, w2;                   /* per block, relative: queries|keys|values [3E,E], out_projection, fc1, fc2 */
  long long ln1_w, ln1_b, bqkv, bo, ln2_w, ln2_b, b1, b2;
  long long layer_weights;                      /* elements in [wqkv, w2 end) */
  long long lnf_w, lnf_b, hw1, hb1, hw2, hb2;   /* vit.norm, classifier.fc1, classifier.fc2 */
  long long total;
} VgVitLayout;
int vg_vit_layout(const VgVitDims* d, VgVitLayout* out); /* host only */
long long vg_vit_ws_bytes(const VgVitDims* d, int B);   /* host only: activation + scratch workspace */

/* Byte offsets inside the workspace of what a forward saves and a backward leaves behind (host only; used by the parity
 * tests to teacher-force single encoder blocks with the tensors the kernels produced).  M = B*(NP+1) rows.  Per block l:
 * X[l] at X + l*M*E*2 (X[L] = trunk output), xn1 / ao / xmid / xn2 likewise, qkv at qkv + l*M*3E*2, z1 (gelu') and a1
 * (gelu) at + l*M*R*E*2, lse at lse + l*B*H*S*4, mean / rstd at + l*M*4.  Backward scratch, one set per block parity:
 * the backward of block l reads dL/dX[l+1] from gin[l&1] and writes dL/dX[l] to gin[(l&1)^1]; gmid = dL/d x_mid. */
typedef struct VgVitWsMap {
  long long X, xn1, qkv, ao, xmid, xn2, z1, a1, lse, mean1, rstd1, mean2, rstd2;
  long long gin[2], gmid[2], dqkv[2], dz1[2];
  long long total;
  /* ABI v7 - the pruned tail of the TOP block: only the CLS rows of its output reach the classifier (modules.py:195), so behind its
   * attention the block runs on compact [B, E] tensors.  xtop = X[L] on the CLS rows (bf16 [B, E]; the rows of X at L*M*E are not
   * written any more), dxtop = dL/dX[L] on the CLS rows (bf16 [B, E], exactly zero on every other row; gin[] holds dL/dX[l] for l < L). */
  long long xtop, dxtop;
} VgVitWsMap;
int vg_vit_ws_map(const VgVitDims* d, int B, VgVitWsMap* out);

typedef struct VgVitNet {
  VgVitDims d;
  const float* P;   /* fp32 master parameters */
  const void* Pb;   /* bf16 shadow of P */
  float* G;         /* fp32 gradient accumulator (same layout); may be NULL when want_wgrad == 0 */
  /* nn.Dropout(p) at the reference's three sites (modules.py:80,170,176: embedding, after the attention
   * output projection, after fc2), fused into the GEMM epilogues.  p = 0 disables it (eval mode).  p is
   * quantised to 1/256; the mask is a counter-based hash of (dropout_seed, site, element index), so the
   * backward regenerates it: pass the SAME seed to the backward of a forward. */
  float dropout_p;
  unsigned long long dropout_seed;
  const unsigned* dropout_step; /* optional device counter mixed into the mask key (hipGraph replay); NULL = none */
  void* ctx;                    /* optional vg_ctx_create() handle: the backward runs its weight-gradient side on the
                                   context's second stream, concurrently with the input-gradient chain; NULL = one stream */
  int attn_fp8;                 /* 1: fp8 (OCP e4m3) MFMA operands for the attention's activation products, Q.K^T (forward
                                   and the backward's recompute) and P.V, as BASELINE.json's 128x128 configuration asks;
                                   gradient-carrying products stay bf16.  0 = bf16 everywhere (default; parity tiers) */
  int dense_top;                /* ABI v7.  0 (default): the top encoder block computes what the classifier reads - behind its attention
                                   only the B CLS rows (modules.py:195 takes x[:, 0, :]; the other rows of its output are never read and their
                                   gradient is exactly zero), forward, backward and weight gradients.  1: every row of it, like the blocks below -
                                   the reference's operator graph row for row (for A/B measurements and the test that the two agree) */
} VgVitNet;
/* img: [B,C,IH,IH] fp32 (img_is_bf16 = 0) or bf16; logits fp32 [B,Kc].  ws keeps everything the
 * backward needs; one ws per in-flight forward. */
int vg_vit_forward(const VgVitNet* net, int B, const void* img, int img_is_bf16, void* ws,
                   float* logits, void* stream);
/* dlogits fp32 [B,Kc].  d_img bf16 [B,C,IH,IH] or NULL.  want_wgrad: accumulate into net->G
 * (G += dL/dP); 0 skips every weight-gradient kernel (generator pass through D, training.py:204-210). */
int vg_vit_backward(const VgVitNet* net, int B, void* ws, const float* dlogits, void* d_img,
                    int want_wgrad, void* stream);
/* The same backward in pieces, for overlapping the data-parallel gradient all-reduce with compute:
 * stage 0 = classifier head + final LayerNorm, stages 1..L = encoder blocks L-1..0, stage L+1 = patch
 * embedding.  Calls must cover [0, L+2) in increasing order on one stream with the same arguments.
 * After a call returning stages up to s, the gradients of blocks >= L-s (a contiguous tail of the
 * flat buffer, from layer0 + (L-s)*layer_stride) are final for this backward. */
int vg_vit_backward_stages(const VgVitNet* net, int B, void* ws, const float* dlogits, void* d_img,
                           int want_wgrad, int stage_begin, int stage_end, void* stream);

/* ABI v9.  The WGAN-GP gradient penalty of the reference's Wasserstein step as ONE call (replaces the autograd double backward over
 * the operator set; reference: gradient_penalty, src/v2/utils.py:124-144, and its call site src/v2/training.py:101-106):
 *   penalty = mean_b (|| d sum(D(x^_b)) / d x^_b ||_2 - 1)^2,  x^ = eps real + (1 - eps) fake;   net->G += weight * d penalty / d theta;
 *   *penalty_out = penalty (device float).  real / fake: bf16 [B, C, IH, IH]; eps: device fp32 [B] (the reference draws torch.rand).
 * ws: a vg_vit_ws_bytes(d, B) workspace (the call runs its own forward in it); ws_pen: vg_vit_penalty_ws_bytes(d, B) bytes.
 * The discriminator runs in train mode like the reference's does: net->dropout_p / dropout_seed / dropout_step give the masks (one set
 * for all passes of the call).  net->dense_top and net->ctx are ignored (every row of the top block; one stream).  Every network
 * vg_vit_layout accepts; where the full-row kernels take the shape (E = 384 / 512, B * tokens a multiple of 16) the input gradients +
 * LayerNorm backwards are fused.  -3: attn_fp8 (the second-order attention kernel differentiates the bf16 one), B * E not a multiple of 8. */
long long vg_vit_penalty_ws_bytes(const VgVitDims* d, int B);
int vg_vit_penalty(const VgVitNet* net, int B, const void* real, const void* fake, const float* eps, float weight, void* ws, void* ws_pen,
                   float* penalty_out, void* stream);
/* v1 generator: mapping Linear -> L x TransformerSLN -> SLN -> SIREN x2 (src/v1/generator.py:58-69). */
typedef struct VgGenDims {
  int Z, T, E, H, L, O, CW; /* latent, tokens, embed, heads, layers, siren hidden, output features per token */
  float omega0;
  /* Token geometry.  patch == 0: the reference's v1 layout - one token per image ROW, CW = channels*image_w and the
   * [B, T*CW] result IS the image through a flat view (generator.py:19,25,66-68).
   * patch > 0: tokens on the discriminator's patch grid - T = (IH/patch)^2, CW = C*patch^2 in
   * conv1's (c, py, px) order - and the image is assembled by the un-patchify scatter.  Not in the reference. */
  int patch, C, IH;
} VgGenDims;
typedef struct VgGenLayout {
  long long emb, map_w, map_b;                  /* embedding [T,E], mapping Linear [T*E, Z], bias */
  long long layer0, layer_stride;
  long long wqkv, wo, wm;                       /* per block, relative: all q heads | k heads | v heads [3E,E]; output_linear; mlp */
  long long sln1_w, sln1_b, sln1_s, sln2_w, sln2_b, sln2_s, bo, bm; /* *_s: [gamma, beta] scalars */
  long long layer_weights;
  long long slnf_w, slnf_b, slnf_s, s1_w, s1_b, s2_w, s2_b;
  long long total;
} VgGenLayout;
int vg_gen_layout(const VgGenDims* d, VgGenLayout* out);
long long vg_gen_ws_bytes(const VgGenDims* d, int B);
/* Workspace introspection for the parity tests (host only), byte offsets; R = B*T rows.  Per block l: s1 / cat / htmp /
 * s2 / hout at + l*R*E*2, qkv at + l*R*3E*2.  wmod = the mapping output [R,E]; sf = final SLN output; y1 = first SIREN
 * output, zf1 / zf2 = fp32 SIREN pre-activations.  g[0..2]: gradient buffers of the backward - dL/d h entering block l
 * (from above) is in g[0] when L-1-l is even, g[2] when odd; g[1] holds dL/d h_tmp of the block being processed.
 * dw_acc: fp32 [R,E] gradient of the modulation vector, summed over the SLN uses processed so far. */
typedef struct VgGenWsMap {
  long long wmod, s1, qkv, cat, htmp, s2, hout, sf, y1, zf1, zf2, g[3], dw_acc, total;
} VgGenWsMap;
int vg_gen_ws_map(const VgGenDims* d, int B, VgGenWsMap* out);
typedef struct VgGenNet {
  VgGenDims d;
  const float* P;
  const void* Pb;
  float* G;
  /* Dropout after msha.output_linear and inside the block MLP (src/v1/config.py:36,39: 0.2 / 0.2). */
  float dropout_p;
  unsigned long long dropout_seed;
  const unsigned* dropout_step;
  /* Optional Fourier positional input of the SIREN (named by BASELINE.json's north_star; NOT in the reference):
   * fp32 [T, E] table added to the final SLN output of every image before output_network.0.  NULL = reference. */
  const float* pos_table;
} VgGenNet;
/* z fp32 [B,Z]; img bf16 [B, T*CW]: the flat view of generator.py:66-68 (patch == 0) or NCHW [B,C,IH,IH] (patch > 0). */
int vg_gen_forward(const VgGenNet* net, int B, const float* z, void* ws, void* img, void* stream);
int vg_gen_backward(const VgGenNet* net, int B, void* ws, const void* d_img, void* stream);
/* The same backward in pieces (data-parallel overlap, and block-by-block parity tests): stage 0 = SIREN output layers +
 * final SLN, stages 1..L = blocks L-1..0, stage L+1 = learned embedding + mapping Linear.  Calls must cover [0, L+2) in
 * increasing order on one stream with the same arguments.  After the call that returns stage s (1 <= s <= L) the
 * gradients from layer0 + (L-s)*layer_stride to the end of the flat buffer are final for this backward. */
int vg_gen_backward_stages(const VgGenNet* net, int B, void* ws, const void* d_img, int stage_begin,
                           int stage_end, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITGAN_HIP_H */
