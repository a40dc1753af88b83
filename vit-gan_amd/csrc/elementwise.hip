// Patchify / un-patchify, CLS/positional bookkeeping, classifier head, GAN losses, fused AdamW
// and the casts around them.  All memory-bound, coalesced, deterministic (no float atomics).
#include "vg_common.h"

// ---- patchify: NCHW image -> [B*NP, C*P*P] bf16 rows in (c, py, px) order ----------------------
// (the memory order of conv1.weight[e], src/v2/modules.py:70-72, so the conv is one NT GEMM)
template <typename T>
__global__ __launch_bounds__(256) void vg_patchify_kernel(const T* __restrict__ img, bf16* __restrict__ A, int B, int C,
                                                          int IH, int P) {
  const int G = IH / P, K = C * P * P;
  const long long total = (long long)B * C * IH * G;  // one thread per (b, c, y, gx): P contiguous pixels
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int gx = (int)(i % G);
  long long t = i / G;
  const int y = (int)(t % IH); t /= IH;
  const int c = (int)(t % C);
  const int b = (int)(t / C);
  const int gy = y / P, py = y - gy * P;
  const T* src = img + (((size_t)b * C + c) * IH + y) * IH + gx * P;
  bf16* dst = A + ((size_t)b * G * G + gy * G + gx) * K + (c * P + py) * P;
  for (int px = 0; px < P; ++px) dst[px] = vg_f2bf((float)src[px]);
}
// ---- inverse of the above for gradients: dA [B*NP, K] bf16 -> d_img [B,C,IH,IH] bf16 -----------
__global__ __launch_bounds__(256) void vg_unpatchify_kernel(const bf16* __restrict__ dA, bf16* __restrict__ dimg, int B,
                                                            int C, int IH, int P) {
  const int G = IH / P, K = C * P * P;
  const long long total = (long long)B * C * IH * G;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int gx = (int)(i % G);
  long long t = i / G;
  const int y = (int)(t % IH); t /= IH;
  const int c = (int)(t % C);
  const int b = (int)(t / C);
  const int gy = y / P, py = y - gy * P;
  bf16* dst = dimg + (((size_t)b * C + c) * IH + y) * IH + gx * P;
  const bf16* src = dA + ((size_t)b * G * G + gy * G + gx) * K + (c * P + py) * P;
  for (int px = 0; px < P; ++px) dst[px] = src[px];
}

// ---- CLS rows of the token matrix: x[b*S + 0, :] = dropout(cls) (the embedding dropout, modules.py:99) ----
__global__ __launch_bounds__(256) void vg_fill_cls_kernel(bf16* __restrict__ x, const float* __restrict__ cls, int B, int S,
                                                          int E, unsigned dthr, unsigned dkey0, float dscale, const unsigned* __restrict__ dstep) {
  const unsigned dkey = vg_drop_key(dkey0, dstep);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * E) return;
  const int b = i / E, e = i - b * E;
  float v = cls[e];
  if (dthr) {
    const unsigned idx = (unsigned)(b * S) * (unsigned)E + (unsigned)e;
    v *= vg_drop_factor(vg_drop_word(dkey, idx >> 2), idx & 3, dthr, dscale);
  }
  x[(size_t)b * S * E + e] = vg_f2bf(v);
}
// y[i] = x[i] * mask(i) / keep   (gradient of a dropout site where no producer kernel can fuse it; n % 4 == 0)
__global__ __launch_bounds__(256) void vg_dropout_apply_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long long n,
                                                               unsigned dthr, unsigned dkey0, float dscale,
                                                               const unsigned* __restrict__ dstep) {
  const unsigned dkey = vg_drop_key(dkey0, dstep);
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const bf16x4 v = *(const bf16x4*)(x + i4);
  const unsigned wd = vg_drop_word(dkey, (unsigned)(i4 >> 2));
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = vg_f2bf(vg_bf2f(v[j]) * vg_drop_factor(wd, j, dthr, dscale));
  *(bf16x4*)(y + i4) = o;
}
// ---- gather / scatter of row subsets ---------------------------------------------------------
// out[(b*n_take + j), :] = in[(b*S + first + j), :]      (16 B per thread)
__global__ __launch_bounds__(256) void vg_take_rows_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int B, int S,
                                                           int first, int n_take, int E) {
  const int cpr = E / 8;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * n_take * cpr) return;
  const int c = (int)(i % cpr);
  const long long r = i / cpr;
  const int b = (int)(r / n_take), j = (int)(r - (long long)b * n_take);
  *(u32x4*)(out + (size_t)r * E + 8 * c) = *(const u32x4*)(in + ((size_t)b * S + first + j) * E + 8 * c);
}
// g[(b*S + 0), :] = src[b, :], every other row of g = 0;  gm (nullable): g times the dropout mask of the [B*S, E] buffer
__global__ __launch_bounds__(256) void vg_scatter_cls_kernel(const bf16* __restrict__ src, bf16* __restrict__ g, int B, int S,
                                                             int E, bf16* __restrict__ gm, unsigned dthr, unsigned dkey0, float dscale,
                                                             const unsigned* __restrict__ dstep) {
  const int cpr = E / 8;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * S * cpr) return;
  const int c = (int)(i % cpr);
  const long long r = i / cpr;
  const int b = (int)(r / S), s = (int)(r - (long long)b * S);
  u32x4 v = {0u, 0u, 0u, 0u};
  if (s == 0) v = *(const u32x4*)(src + (size_t)b * E + 8 * c);
  *(u32x4*)(g + (size_t)r * E + 8 * c) = v;
  if (gm) {
    if (s == 0) {  // same index and arithmetic as vg_dropout_apply_kernel over the whole buffer
      const unsigned dkey = vg_drop_key(dkey0, dstep);
      const long long i4 = (r * E + 8 * c) >> 2;
      const bf16x8 x = __builtin_bit_cast(bf16x8, v);
      bf16x8 o;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned wd = vg_drop_word(dkey, (unsigned)(i4 + h));
#pragma unroll
        for (int j = 0; j < 4; ++j) o[4 * h + j] = vg_f2bf(vg_bf2f(x[4 * h + j]) * vg_drop_factor(wd, j, dthr, dscale));
      }
      v = __builtin_bit_cast(u32x4, o);
    }
    *(u32x4*)(gm + (size_t)r * E + 8 * c) = v;
  }
}
// out[s, e] = sum_b g[(b*S + s), e]   (fp32).  A workgroup owns 64 consecutive columns of one token row (a whole
// 128-B line per batch item) x 32 batch lanes: 16-byte loads, 4 independent loads in flight per thread, and a
// fixed-order fold of the 32 batch lanes through LDS (deterministic).
__global__ __launch_bounds__(256) void vg_batch_sum_kernel(const bf16* __restrict__ g, float* __restrict__ out, int B, int S,
                                                           int E) {
  __shared__ float red[32][65];
  const int cl = threadIdx.x & 7, bl = threadIdx.x >> 3;     // 8 chunks of 8 columns, 32 batch lanes
  const int chunks = E / 64;                                  // 64-column groups per row
  const int s = blockIdx.x / chunks, e0 = (blockIdx.x - s * chunks) * 64 + 8 * cl;
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.f;
  const bf16* p = g + (size_t)s * E + e0;
  const size_t bstride = (size_t)S * E;
  int b = bl;
  for (; b + 96 < B; b += 128) {
    bf16x8 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = *(const bf16x8*)(p + (size_t)(b + 32 * u) * bstride);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += vg_bf2f(t[u][j]);
  }
  for (; b < B; b += 32) {
    const bf16x8 t = *(const bf16x8*)(p + (size_t)b * bstride);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += vg_bf2f(t[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[bl][8 * cl + j] = a[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) r += red[k][threadIdx.x];
    out[(size_t)s * E + (blockIdx.x - s * chunks) * 64 + threadIdx.x] = r;
  }
}
// embed grads from tok_sum [S,E]: d_cls += tok_sum[0]; d_pos += tok_sum[1:]; d_convbias += sum_n tok_sum[1+n]
// One thread per (token, column) for the elementwise part; the first E threads also fold the conv-bias column sums.
__global__ __launch_bounds__(256) void vg_embed_small_grads_kernel(const float* __restrict__ tok_sum, float* __restrict__ d_cls,
                                                                   float* __restrict__ d_pos, float* __restrict__ d_bias, int S,
                                                                   int E) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= S * E) return;
  const float t = tok_sum[i];
  if (i < E) d_cls[i] += t; else d_pos[i - E] += t;
  if (i < E) {
    float a = 0.f;
#pragma unroll 8
    for (int n = 1; n < S; ++n) a += tok_sum[(size_t)n * E + i];
    d_bias[i] += a;
  }
}

// ---- classifier tail: logits[b,k] = t[b,:] . W2[k,:] + b2[k]   (one wave per (b,k)) -------------
__global__ __launch_bounds__(256) void vg_head_fc2_kernel(const bf16* __restrict__ t, const float* __restrict__ W2,
                                                          const float* __restrict__ b2, float* __restrict__ logits, int B, int E,
                                                          int Kc) {
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wv >= B * Kc) return;
  const int b = wv / Kc, k = wv - b * Kc;
  float a = 0.f;
  for (int e = lane; e < E; e += 64) a += vg_bf2f(t[(size_t)b * E + e]) * W2[(size_t)k * E + e];
  a = vg_wave_sum(a);
  if (lane == 0) logits[wv] = a + b2[k];
}
// dz1[b,e] = (sum_k dlog[b,k] W2[k,e]) * (1 - t[b,e]^2)
__global__ __launch_bounds__(256) void vg_head_bwd_dz_kernel(const float* __restrict__ dlog, const float* __restrict__ W2,
                                                             const bf16* __restrict__ t, bf16* __restrict__ dz, int B, int E,
                                                             int Kc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * E) return;
  const int b = i / E, e = i - b * E;
  float a = 0.f;
  for (int k = 0; k < Kc; ++k) a += dlog[b * Kc + k] * W2[(size_t)k * E + e];
  const float tv = vg_bf2f(t[i]);
  dz[i] = vg_f2bf(a * (1.f - tv * tv));
}
// dW2[k,e] += sum_b dlog[b,k] t[b,e];  db2[k] += sum_b dlog[b,k]      (one wave per (k, e); lanes over b)
__global__ __launch_bounds__(256) void vg_head_bwd_w2_kernel(const float* __restrict__ dlog, const bf16* __restrict__ t,
                                                             float* __restrict__ dW2, float* __restrict__ db2, int B, int E,
                                                             int Kc) {
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wv >= Kc * (E + 1)) return;
  const int k = wv / (E + 1), e = wv - k * (E + 1);
  float a = 0.f;
  if (e < E) {
    for (int b = lane; b < B; b += 64) a += dlog[b * Kc + k] * vg_bf2f(t[(size_t)b * E + e]);
  } else {
    for (int b = lane; b < B; b += 64) a += dlog[b * Kc + k];
  }
  a = vg_wave_sum(a);
  if (lane == 0) { if (e < E) dW2[(size_t)k * E + e] += a; else db2[k] += a; }
}

// The three of them in one launch (the head is 512 rows: three launches cost more than the work), fully parallel over the rows:
// workgroup (x, y) owns 64 columns e and 32 rows b (4 waves x 8 rows, all loads in flight at once); dz as above, and per workgroup
// ONE partial row  part[y][ k*E + e ] = sum_b dlog[b,k] t[b,e] | part[y][ Kc*E + e ] = sum_b dz[b,e] (of the ROUNDED dz, the tensor
// the fc1 weight gradient reads) | part[y][ (Kc+1)*E + k ] = sum_b dlog[b,k]  - folded over y into dW2 / db1 / db2 by the caller's
// deferred fold (vg_colsum_f32_multi_kernel), fixed order.  part == nullptr: dz only.
#define VG_HEAD_KMAX 16
#define VG_HEAD_ROWS 32
__global__ __launch_bounds__(256) void vg_head_bwd_all_kernel(const float* __restrict__ dlog, const float* __restrict__ W2,
                                                              const bf16* __restrict__ t, bf16* __restrict__ dz, float* __restrict__ part,
                                                              int B, int E, int Kc) {
  __shared__ float red[4][VG_HEAD_KMAX + 1][64];
  __shared__ float red2[4][VG_HEAD_KMAX];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const bool eok = e < E;
  const int b0 = blockIdx.y * VG_HEAD_ROWS + wv * 8;
  float tv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) tv[u] = (eok && b0 + u < B) ? vg_bf2f(t[(size_t)(b0 + u) * E + e]) : 0.f;
  float w2[VG_HEAD_KMAX], aw[VG_HEAD_KMAX], a2[VG_HEAD_KMAX];
#pragma unroll
  for (int k = 0; k < VG_HEAD_KMAX; ++k) { w2[k] = (k < Kc && eok) ? W2[(size_t)k * E + e] : 0.f; aw[k] = 0.f; a2[k] = 0.f; }
  float cs = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int b = b0 + u;
    if (b < B) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < VG_HEAD_KMAX; ++k) {
        if (k < Kc) {
          const float d = dlog[b * Kc + k];
          a += d * w2[k];
          aw[k] += d * tv[u];
          a2[k] += d;
        }
      }
      const bf16 o = vg_f2bf(a * (1.f - tv[u] * tv[u]));
      if (eok) dz[(size_t)b * E + e] = o;
      cs += vg_bf2f(o);
    }
  }
  if (!part) return;
#pragma unroll
  for (int k = 0; k < VG_HEAD_KMAX; ++k) red[wv][k][lane] = aw[k];
  red[wv][VG_HEAD_KMAX][lane] = cs;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < VG_HEAD_KMAX; ++k) red2[wv][k] = a2[k];  // every lane of a wave holds the same a2[k]
  __syncthreads();
  float* row = part + (size_t)blockIdx.y * ((size_t)(Kc + 1) * E + Kc);
  if (wv == 0 && eok) {
    for (int k = 0; k < Kc; ++k) row[(size_t)k * E + e] = (red[0][k][lane] + red[1][k][lane]) + (red[2][k][lane] + red[3][k][lane]);
    row[(size_t)Kc * E + e] = (red[0][VG_HEAD_KMAX][lane] + red[1][VG_HEAD_KMAX][lane]) + (red[2][VG_HEAD_KMAX][lane] + red[3][VG_HEAD_KMAX][lane]);
  }
  if (blockIdx.x == 0 && threadIdx.x < Kc)
    row[(size_t)(Kc + 1) * E + threadIdx.x] = (red2[0][threadIdx.x] + red2[1][threadIdx.x]) + (red2[2][threadIdx.x] + red2[3][threadIdx.x]);
}

// ---- start of a training step --------------------------------------------------------------------------------------------
// g[0, n) = 0 (zero_grad of the discriminator) and the device step counter += 1, in one launch: nothing in THIS kernel reads the
// counter, every later kernel of the step (dropout keys, AdamW's bias correction, the latent noise below) sees the new value.
__global__ __launch_bounds__(256) void vg_zero_tick_kernel(float* __restrict__ g, long long n, int* __restrict__ step) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 == 0 && step) step[0] += 1;
  if (i4 < n) *(f32x4*)(g + i4) = (f32x4){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ uint32_t vg_mix32(uint32_t x) {  // "lowbias32" integer finaliser: every input bit reaches every output bit
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
// The step's inputs in one launch: imgs[0, n_img) = bf16(real) (the GEMM operand type) and, when z != nullptr, the latent batch
// z[0, n_z) ~ N(0, 1) (construct_noise of src/v2/training.py:35-42 is torch.randn): counter-based - element pair j of step s under
// seed k is Box-Muller of two 24-bit uniforms hashed from (k, s, j) - so a replayed hipGraph draws fresh noise every step, a resumed
// run continues the sequence, and no generator state lives on the device.  |z| <= 5.77 (u1 >= 2^-24).
__global__ __launch_bounds__(256) void vg_step_inputs_kernel(const float* __restrict__ real, bf16* __restrict__ imgs, long long n_img,
                                                             float* __restrict__ z, long long n_z, uint32_t seed_lo, uint32_t seed_hi,
                                                             const int* __restrict__ step) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long i4 = t * 4;
  if (real && i4 < n_img) {
    const f32x4 v = *(const f32x4*)(real + i4);
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = vg_f2bf(v[j]);
    *(bf16x4*)(imgs + i4) = o;
  }
  if (z && 2 * t < n_z) {
    const uint32_t key = vg_mix32(seed_lo ^ vg_mix32((uint32_t)step[0] * 0x9E3779B1u + seed_hi));
    const uint32_t a = vg_mix32(vg_mix32((uint32_t)(2 * t) + key) ^ seed_hi), b = vg_mix32(vg_mix32((uint32_t)(2 * t + 1) + key) ^ seed_hi);
    const float u1 = (float)((a >> 8) + 1u) * 5.9604644775390625e-8f;  // (0, 1]
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-8f;         // [0, 1)
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    z[2 * t] = r * cs;
    if (2 * t + 1 < n_z) z[2 * t + 1] = r * sn;
  }
}

// ---- GAN losses on logits [n] ------------------------------------------------------------------
// kind: 0 = non-saturating BCE-with-logits (v1 gan.py:16-20 semantics), 1 = hinge.
// role: 0 = D on real (target 1), 1 = D on fake (target 0), 2 = G (target 1 / -mean).
// loss_out[0] = mean loss; dlog[i] = d(mean loss)/d logit[i] * grad_scale.
__device__ __forceinline__ void vg_gan_loss_body(const float* __restrict__ logit, float* __restrict__ dlog, float* __restrict__ loss_out, int n,
                                                 int kind, int role, float grad_scale) {
  __shared__ float red[4];
  float acc = 0.f;
  const float inv = 1.0f / (float)n;
  // branch-free on purpose: both losses are evaluated and selected (uniform kind/role)
  const float t = (role == 1) ? 0.f : 1.f;      // BCE target
  const float sgn = (role == 1) ? 1.f : -1.f;   // hinge: relu(1 + sgn*x)
  for (int i = threadIdx.x; i < n; i += 256) {
    const float x = logit[i];
    const float l_ns = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
    const float d_ns = 1.f / (1.f + __expf(-x)) - t;
    const float hm = 1.f + sgn * x;
    const float l_h = (role == 2) ? -x : fmaxf(hm, 0.f);
    const float d_h = (role == 2) ? -1.f : ((hm > 0.f) ? sgn : 0.f);
    const float l_w = (role == 1) ? x : -x;        // Wasserstein critic (src/v2/training.py:72,97): -(E[D(real)] - E[D(fake)]), G: -E[D(fake)]
    const float d_w = (role == 1) ? 1.f : -1.f;
    const float l = (kind == 0) ? l_ns : ((kind == 1) ? l_h : l_w);
    const float d = (kind == 0) ? d_ns : ((kind == 1) ? d_h : d_w);
    acc += l;
    dlog[i] = d * inv * grad_scale;
  }
  acc = vg_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss_out[0] = (red[0] + red[1] + red[2] + red[3]) * inv;
}
__global__ __launch_bounds__(256) void vg_gan_loss_kernel(const float* __restrict__ logit, float* __restrict__ dlog,
                                                          float* __restrict__ loss_out, int n, int kind, int role,
                                                          float grad_scale) {
  vg_gan_loss_body(logit, dlog, loss_out, n, kind, role, grad_scale);
}
// Two segments of one logit vector (the fused real + fake discriminator pass): workgroup i takes segment i.
__global__ __launch_bounds__(256) void vg_gan_loss_pair_kernel(const float* __restrict__ logit, float* __restrict__ dlog,
                                                               float* __restrict__ loss_out, int n0, int role0, int n1, int role1, int kind,
                                                               float grad_scale) {
  if (blockIdx.x == 0) vg_gan_loss_body(logit, dlog, loss_out, n0, kind, role0, grad_scale);
  else vg_gan_loss_body(logit + n0, dlog + n0, loss_out + 1, n1, kind, role1, grad_scale);
}

// ---- torch.nn.utils.clip_grad_norm_ over a flat gradient buffer (src/v2/training.py:78,104) -------------------------
// Two deterministic stages: per-workgroup sums of squares, then every workgroup folds the partials in the same fixed
// order, derives coef = min(1, max_norm / (gscale*|g| + 1e-6)) and scales its slice in place.
#define VG_CLIP_PARTS 1024
__global__ __launch_bounds__(256) void vg_sumsq_part_kernel(const float* __restrict__ g, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  float a = 0.f;
  for (long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i4 < n; i4 += (long long)gridDim.x * 1024) {
    const f32x4 v = *(const f32x4*)(g + i4);
    a += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  a = vg_wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void vg_clip_scale_kernel(float* __restrict__ g, long long n, const float* __restrict__ part, int nparts,
                                                            float gscale, float max_norm, float* __restrict__ norm_out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
  a = vg_wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3])) * gscale;
  const float coef = fminf(1.f, max_norm / (norm + 1e-6f));
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
  if (coef >= 1.f) return;
  for (long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i4 < n; i4 += (long long)gridDim.x * 1024) {
    f32x4 v = *(f32x4*)(g + i4);
    v *= coef;
    *(f32x4*)(g + i4) = v;
  }
}
int vg_grad_clip_launch(float* g, long long n, float gscale, float max_norm, float* scratch, hipStream_t st) {
  if (n & 3) return -3;
  long long blocks = (n / 4 + 255) / 256;
  const int nb = (int)(blocks < VG_CLIP_PARTS ? blocks : VG_CLIP_PARTS);
  hipLaunchKernelGGL(vg_sumsq_part_kernel, dim3(nb), dim3(256), 0, st, g, n, scratch + 1);
  hipLaunchKernelGGL(vg_clip_scale_kernel, dim3(nb), dim3(256), 0, st, g, n, scratch + 1, nb, gscale, max_norm, scratch);
  return (int)hipGetLastError();
}

// ---- diversity loss of the reference's unreached generator step (src/v2/utils.py:147-152, training.py:73-74) --------
// L = sum_{i,j} |x_i - x_j|_1 / (B (B-1)) over the batch of flattened images.  One workgroup per 256 features: the B
// values of a feature sit in LDS, thread i (feature f, sample b) sums |x_b - x_j| and sign(x_b - x_j) over j:
//   loss += sum_j |x_b - x_j|;   dL/dx_b[f] = 2 sum_j sign(x_b[f] - x_j[f]) / (B (B-1))  (both orders of a pair count).
// d_img += weight * dL/dx (bf16), part[block] = partial loss (folded by the caller's second launch: deterministic).
__global__ __launch_bounds__(256) void vg_diversity_kernel(const bf16* __restrict__ x, bf16* __restrict__ d_img, float* __restrict__ part,
                                                           int B, int D, float weight) {
  extern __shared__ float col[];  // [16 features][B]
  __shared__ float red[4];
  const int f0 = blockIdx.x * 16;
  for (int i = threadIdx.x; i < 16 * B; i += 256) {
    const int f = i & 15, b = i >> 4;
    col[f * B + b] = (f0 + f < D) ? vg_bf2f(x[(size_t)b * D + f0 + f]) : 0.f;
  }
  __syncthreads();
  float lsum = 0.f;
  const float inv = 1.0f / ((float)B * (float)(B - 1));
  for (int i = threadIdx.x; i < 16 * B; i += 256) {
    const int f = i & 15, b = i >> 4;
    if (f0 + f >= D) continue;
    const float v = col[f * B + b];
    float a = 0.f, sg = 0.f;
    for (int j = 0; j < B; ++j) {
      const float d = v - col[f * B + j];
      a += fabsf(d);
      sg += (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
    }
    lsum += a;
    if (d_img) {
      bf16* g = d_img + (size_t)b * D + f0 + f;
      *g = vg_f2bf(vg_bf2f(*g) + weight * 2.f * sg * inv);
    }
  }
  lsum = vg_wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * inv;
}
__global__ __launch_bounds__(256) void vg_fold_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += part[i];
  a = vg_wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}
int vg_diversity_launch(const bf16* x, bf16* d_img, float* loss_out, float* scratch, int B, int D, float weight, hipStream_t st) {
  if (B < 2 || D < 1 || (size_t)16 * B * 4 > 64 * 1024) return -3;
  const int nb = (D + 15) / 16;
  hipLaunchKernelGGL(vg_diversity_kernel, dim3(nb), dim3(256), (size_t)16 * B * 4, st, x, d_img, scratch, B, D, weight);
  hipLaunchKernelGGL(vg_fold_sum_kernel, dim3(1), dim3(256), 0, st, scratch, nb, loss_out);
  return (int)hipGetLastError();
}

// ---- fused AdamW over a flat parameter buffer (torch.optim.AdamW semantics) ----------------------
// p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps);  shadow = bf16(p).   g is pre-scaled by gscale.
__global__ __launch_bounds__(256) void vg_adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, bf16* __restrict__ shadow, long long n, float lr,
                                                       float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                       float gscale, const int* __restrict__ step_dev) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (step_dev) {  // step counter kept on the device so a captured hipGraph replays correctly
    const float t = (float)step_dev[0];
    bc1 = 1.f - __powf(b1, t);
    bc2_sqrt = sqrtf(1.f - __powf(b2, t));
  }
  f32x4 pv = *(f32x4*)(p + i4), gv = *(const f32x4*)(g + i4), mv = *(f32x4*)(m + i4), vv = *(f32x4*)(v + i4);
  bf16x4 sh;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float gg = gv[j] * gscale;
    float pp = pv[j] * (1.f - lr * wd);
    const float mm = b1 * mv[j] + (1.f - b1) * gg;
    const float v2 = b2 * vv[j] + (1.f - b2) * gg * gg;
    pp -= (lr / bc1) * mm / (sqrtf(v2) / bc2_sqrt + eps);
    pv[j] = pp; mv[j] = mm; vv[j] = v2; sh[j] = vg_f2bf(pp);
  }
  *(f32x4*)(p + i4) = pv; *(f32x4*)(m + i4) = mv; *(f32x4*)(v + i4) = vv;
  *(bf16x4*)(shadow + i4) = sh;
}
__global__ __launch_bounds__(256) void vg_cast_f32_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long long n) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const f32x4 s = *(const f32x4*)(src + i4);
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = vg_f2bf(s[j]);
  *(bf16x4*)(dst + i4) = o;
}
// dst[i] (+)= sum_s slab[s][i]   (wgrad split-K slabs -> gradient buffer)
__global__ __launch_bounds__(256) void vg_slab_reduce_kernel(const float* __restrict__ slab, long long stride, int nslab,
                                                             float* __restrict__ dst, long long n, int accumulate) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  f32x4 a = accumulate ? *(const f32x4*)(dst + i4) : (f32x4){0.f, 0.f, 0.f, 0.f};
  // the slices are added in slice order (the fold is deterministic), but their loads are independent: eight in flight per thread
  // (the embedding's 32-slice fold is 18 workgroups - one load at a time it took 11 us)
#pragma unroll 8
  for (int s = 0; s < nslab; ++s) a += *(const f32x4*)(slab + (size_t)s * stride + i4);
  *(f32x4*)(dst + i4) = a;
}
__global__ __launch_bounds__(256) void vg_slab_reduce2_kernel(const float* __restrict__ slab0, const float* __restrict__ slab1, long long stride,
                                                              int nslab, float* __restrict__ dst0, float* __restrict__ dst1, long long n,
                                                              int accumulate) {
  const float* slab = blockIdx.y ? slab1 : slab0;
  float* dst = blockIdx.y ? dst1 : dst0;
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  f32x4 a = accumulate ? *(const f32x4*)(dst + i4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int s = 0; s < nslab; ++s) a += *(const f32x4*)(slab + (size_t)s * stride + i4);
  *(f32x4*)(dst + i4) = a;
}
// SIREN output-layer gradient: dz = dy * w0 * cos(w0 * z)   (z fp32 pre-activation)
__global__ __launch_bounds__(256) void vg_sin_grad_kernel(const bf16* __restrict__ dy, const float* __restrict__ z,
                                                          bf16* __restrict__ dz, long long n, float w0) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const bf16x4 d = *(const bf16x4*)(dy + i4);
  const f32x4 zz = *(const f32x4*)(z + i4);
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = vg_f2bf(vg_bf2f(d[j]) * w0 * __cosf(w0 * zz[j]));
  *(bf16x4*)(dz + i4) = o;
}

// x[r, :] += table[r % period, :]   (bf16 rows, fp32 table): the generator's optional Fourier position signal
__global__ __launch_bounds__(256) void vg_add_table_kernel(bf16* __restrict__ x, const float* __restrict__ table, long long n, int E,
                                                           int period) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const long long r = i4 / E;
  const int c = (int)(i4 - r * E);
  const f32x4 t = *(const f32x4*)(table + (size_t)(r % period) * E + c);
  bf16x4 v = *(bf16x4*)(x + i4);
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = vg_f2bf(vg_bf2f(v[j]) + t[j]);
  *(bf16x4*)(x + i4) = v;
}

// --------------------------------------- launchers ---------------------------------------------
static inline unsigned nblk(long long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// ---- v1 overlapping-window tokeniser (src/v1/patch_encoder.py:54-73) ---------------------------------------------
// images.unfold(2, W, stride).unfold(3, W, stride) has shape [B, C, n, n, W, W]; the reference then takes a FLAT view
// of that (b, c, ty, tx, wy, wx) memory order as [B, n*n, C*W*W] - no permute - and so does this kernel: element f of
// image b's output is window pixel (c, ty, tx, wy, wx) with f = (((c*n + ty)*n + tx)*W + wy)*W + wx.
template <typename T>
__global__ __launch_bounds__(256) void vg_unfold_tokens_kernel(const T* __restrict__ img, bf16* __restrict__ out, int B, int C,
                                                               int IH, int W, int stride, int n) {
  const long long per = (long long)C * n * n * W * W, total = (long long)B * per;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / per);
  long long f = i - (long long)b * per;
  const int wx = (int)(f % W); f /= W;
  const int wy = (int)(f % W); f /= W;
  const int tx = (int)(f % n); f /= n;
  const int ty = (int)(f % n);
  const int c = (int)(f / n);
  out[i] = vg_f2bf((float)img[(((size_t)b * C + c) * IH + ty * stride + wy) * IH + tx * stride + wx]);
}
// adjoint: every pixel gathers the gradient of each window that covers it (fixed order: deterministic, no atomics)
__global__ __launch_bounds__(256) void vg_unfold_tokens_bwd_kernel(const bf16* __restrict__ dout, bf16* __restrict__ dimg, int B,
                                                                   int C, int IH, int W, int stride, int n) {
  const long long total = (long long)B * C * IH * IH;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % IH);
  long long t = i / IH;
  const int y = (int)(t % IH); t /= IH;
  const int c = (int)(t % C);
  const int b = (int)(t / C);
  const long long per = (long long)C * n * n * W * W;
  const int ty0 = y >= W ? (y - W) / stride + 1 : 0, tx0 = x >= W ? (x - W) / stride + 1 : 0;
  float a = 0.f;
  for (int ty = ty0; ty < n && ty * stride <= y; ++ty)
    for (int tx = tx0; tx < n && tx * stride <= x; ++tx) {
      const int wy = y - ty * stride, wx = x - tx * stride;
      a += vg_bf2f(dout[(size_t)b * per + ((((size_t)c * n + ty) * n + tx) * W + wy) * W + wx]);
    }
  dimg[i] = vg_f2bf(a);
}
static int unfold_geometry(int IH, int P, int overlap, int* W, int* stride, int* n) {
  if (IH < 1 || P < 1 || overlap < 0) return -3;
  *W = P + 2 * overlap;
  if (*W > IH) return -3;
  *stride = (IH - P - 2 * overlap) / P + 1;  // patch_encoder.py:20-22
  *n = (IH - (*W - 1) - 1) / *stride + 1;    // :23-27 (== unfold's window count)
  return 0;
}
int vg_unfold_tokens_launch(const void* img, int img_is_bf16, bf16* out, int B, int C, int IH, int P, int overlap, hipStream_t st) {
  int W, stride, n;
  if (unfold_geometry(IH, P, overlap, &W, &stride, &n)) return -3;
  const long long total = (long long)B * C * n * n * W * W;
  if (img_is_bf16) hipLaunchKernelGGL(vg_unfold_tokens_kernel<bf16>, dim3(nblk(total)), dim3(256), 0, st, (const bf16*)img, out, B, C, IH, W, stride, n);
  else hipLaunchKernelGGL(vg_unfold_tokens_kernel<float>, dim3(nblk(total)), dim3(256), 0, st, (const float*)img, out, B, C, IH, W, stride, n);
  return (int)hipGetLastError();
}
int vg_unfold_tokens_bwd_launch(const bf16* dout, bf16* dimg, int B, int C, int IH, int P, int overlap, hipStream_t st) {
  int W, stride, n;
  if (unfold_geometry(IH, P, overlap, &W, &stride, &n)) return -3;
  hipLaunchKernelGGL(vg_unfold_tokens_bwd_kernel, dim3(nblk((long long)B * C * IH * IH)), dim3(256), 0, st, dout, dimg, B, C, IH, W, stride, n);
  return (int)hipGetLastError();
}
int vg_patchify_launch(const void* img, int img_is_bf16, bf16* A, int B, int C, int IH, int P, hipStream_t st) {
  if (IH % P) return -3;
  const long long total = (long long)B * C * IH * (IH / P);
  if (img_is_bf16) hipLaunchKernelGGL(vg_patchify_kernel<bf16>, dim3(nblk(total)), dim3(256), 0, st, (const bf16*)img, A, B, C, IH, P);
  else hipLaunchKernelGGL(vg_patchify_kernel<float>, dim3(nblk(total)), dim3(256), 0, st, (const float*)img, A, B, C, IH, P);
  return (int)hipGetLastError();
}
int vg_unpatchify_launch(const bf16* dA, bf16* dimg, int B, int C, int IH, int P, hipStream_t st) {
  const long long total = (long long)B * C * IH * (IH / P);
  hipLaunchKernelGGL(vg_unpatchify_kernel, dim3(nblk(total)), dim3(256), 0, st, dA, dimg, B, C, IH, P);
  return (int)hipGetLastError();
}
int vg_fill_cls_launch(bf16* x, const float* cls, int B, int S, int E, unsigned dthr, unsigned dkey, float dscale, const unsigned* dstep,
                       hipStream_t st) {
  hipLaunchKernelGGL(vg_fill_cls_kernel, dim3(nblk((long long)B * E)), dim3(256), 0, st, x, cls, B, S, E, dthr, dkey, dscale, dstep);
  return (int)hipGetLastError();
}
int vg_dropout_apply_launch(const bf16* x, bf16* y, long long n, unsigned dthr, unsigned dkey, float dscale, const unsigned* dstep,
                            hipStream_t st) {
  if (n & 3) return -3;
  hipLaunchKernelGGL(vg_dropout_apply_kernel, dim3(nblk(n / 4)), dim3(256), 0, st, x, y, n, dthr, dkey, dscale, dstep);
  return (int)hipGetLastError();
}
int vg_take_rows_launch(const bf16* in, bf16* out, int B, int S, int first, int n_take, int E, hipStream_t st) {
  hipLaunchKernelGGL(vg_take_rows_kernel, dim3(nblk((long long)B * n_take * (E / 8))), dim3(256), 0, st, in, out, B, S, first, n_take, E);
  return (int)hipGetLastError();
}
int vg_scatter_cls_launch(const bf16* src, bf16* g, int B, int S, int E, hipStream_t st, bf16* gm, unsigned dthr, unsigned dkey, float dscale,
                          const unsigned* dstep) {
  if (E & 7) return -3;
  hipLaunchKernelGGL(vg_scatter_cls_kernel, dim3(nblk((long long)B * S * (E / 8))), dim3(256), 0, st, src, g, B, S, E, (dthr ? gm : nullptr), dthr, dkey,
                     dscale, dstep);
  return (int)hipGetLastError();
}
__global__ __launch_bounds__(256) void vg_scatter_cls2_kernel(const bf16* __restrict__ src_a, bf16* __restrict__ dst_a, const bf16* __restrict__ src_b,
                                                              bf16* __restrict__ dst_b, int B, int S, int E) {
  const int cpr = E / 8;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * S * cpr) return;
  const int c = (int)(i % cpr);
  const long long r = i / cpr;
  const int b = (int)(r / S), s = (int)(r - (long long)b * S);
  u32x4 va = {0u, 0u, 0u, 0u}, vb = va;
  if (s == 0) { va = *(const u32x4*)(src_a + (size_t)b * E + 8 * c); vb = *(const u32x4*)(src_b + (size_t)b * E + 8 * c); }
  *(u32x4*)(dst_a + (size_t)r * E + 8 * c) = va;
  *(u32x4*)(dst_b + (size_t)r * E + 8 * c) = vb;
}
int vg_scatter_cls2_launch(const bf16* src_a, bf16* dst_a, const bf16* src_b, bf16* dst_b, int B, int S, int E, hipStream_t st) {
  if ((E & 7) || !src_a || !dst_a || !src_b || !dst_b) return -3;
  hipLaunchKernelGGL(vg_scatter_cls2_kernel, dim3(nblk((long long)B * S * (E / 8))), dim3(256), 0, st, src_a, dst_a, src_b, dst_b, B, S, E);
  return (int)hipGetLastError();
}
int vg_batch_sum_launch(const bf16* g, float* out, int B, int S, int E, hipStream_t st) {
  if (E & 63) return -3;
  hipLaunchKernelGGL(vg_batch_sum_kernel, dim3(S * (E / 64)), dim3(256), 0, st, g, out, B, S, E);
  return (int)hipGetLastError();
}
int vg_embed_small_grads_launch(const float* tok_sum, float* d_cls, float* d_pos, float* d_bias, int S, int E, hipStream_t st) {
  hipLaunchKernelGGL(vg_embed_small_grads_kernel, dim3(nblk((long long)S * E)), dim3(256), 0, st, tok_sum, d_cls, d_pos, d_bias, S, E);
  return (int)hipGetLastError();
}
int vg_head_fc2_launch(const bf16* t, const float* W2, const float* b2, float* logits, int B, int E, int Kc, hipStream_t st) {
  hipLaunchKernelGGL(vg_head_fc2_kernel, dim3(nblk((long long)B * Kc, 4)), dim3(256), 0, st, t, W2, b2, logits, B, E, Kc);
  return (int)hipGetLastError();
}
int vg_head_bwd_parts(int B) { return (B + VG_HEAD_ROWS - 1) / VG_HEAD_ROWS; }
int vg_head_bwd_part_width(int E, int Kc) { return (Kc + 1) * E + Kc; }
// Kc <= 16: ONE launch; with `part` ([vg_head_bwd_parts(B)][vg_head_bwd_part_width(E, Kc)] floats) the gradients of W2, b1 and b2 are
// left as partial rows for the caller's deferred fold and 1 is returned.  Otherwise (or part == nullptr with want_wgrad): the
// separate kernels accumulate dW2 / db2 directly (db1 is then the caller's column sum) and 0 is returned.  < 0: -hipError.
int vg_head_bwd_launch(const float* dlog, const float* W2, const bf16* t, bf16* dz, float* dW2, float* db2, int B, int E, int Kc,
                       int want_wgrad, hipStream_t st, float* part) {
  if (Kc <= VG_HEAD_KMAX && (!want_wgrad || part)) {
    hipLaunchKernelGGL(vg_head_bwd_all_kernel, dim3((E + 63) / 64, vg_head_bwd_parts(B)), dim3(256), 0, st, dlog, W2, t, dz,
                       want_wgrad ? part : nullptr, B, E, Kc);
    const int rc = (int)hipGetLastError();
    return rc ? -rc : (want_wgrad ? 1 : 0);
  }
  hipLaunchKernelGGL(vg_head_bwd_dz_kernel, dim3(nblk((long long)B * E)), dim3(256), 0, st, dlog, W2, t, dz, B, E, Kc);
  if (want_wgrad)
    hipLaunchKernelGGL(vg_head_bwd_w2_kernel, dim3(nblk((long long)Kc * (E + 1), 4)), dim3(256), 0, st, dlog, t, dW2, db2, B, E, Kc);
  const int rc = (int)hipGetLastError();
  return rc ? -rc : 0;
}
int vg_zero_tick_launch(float* g, long long n, int* step, hipStream_t st) {
  if (n < 4 || (n & 3)) return -3;
  hipLaunchKernelGGL(vg_zero_tick_kernel, dim3(nblk(n / 4)), dim3(256), 0, st, g, n, step);
  return (int)hipGetLastError();
}
int vg_step_inputs_launch(const float* real, bf16* imgs, long long n_img, float* z, long long n_z, unsigned long long seed, const int* step,
                          hipStream_t st) {
  if ((real && (n_img & 3)) || (z && !step)) return -3;
  long long thr = real ? n_img / 4 : 0;
  if (z && (n_z + 1) / 2 > thr) thr = (n_z + 1) / 2;
  if (thr < 1) return -2;
  hipLaunchKernelGGL(vg_step_inputs_kernel, dim3(nblk(thr)), dim3(256), 0, st, real, imgs, real ? n_img : 0, z, z ? n_z : 0, (uint32_t)seed,
                     (uint32_t)(seed >> 32), step);
  return (int)hipGetLastError();
}
int vg_gan_loss_pair_launch(const float* logit, float* dlog, float* loss_out, int n0, int role0, int n1, int role1, int kind, float grad_scale,
                            hipStream_t st) {
  if (kind < 0 || kind > 2 || role0 < 0 || role0 > 2 || role1 < 0 || role1 > 2 || n0 < 1 || n1 < 1) return -2;
  hipLaunchKernelGGL(vg_gan_loss_pair_kernel, dim3(2), dim3(256), 0, st, logit, dlog, loss_out, n0, role0, n1, role1, kind, grad_scale);
  return (int)hipGetLastError();
}
int vg_gan_loss_launch(const float* logit, float* dlog, float* loss_out, int n, int kind, int role, float grad_scale,
                       hipStream_t st) {
  if (kind < 0 || kind > 2 || role < 0 || role > 2) return -2;
  hipLaunchKernelGGL(vg_gan_loss_kernel, dim3(1), dim3(256), 0, st, logit, dlog, loss_out, n, kind, role, grad_scale);
  return (int)hipGetLastError();
}
int vg_adamw_launch(float* p, const float* g, float* m, float* v, bf16* shadow, long long n, float lr, float b1, float b2,
                    float eps, float wd, int step, const int* step_dev, float gscale, hipStream_t st) {
  if (n & 3) return -3;
  const float bc1 = 1.f - powf(b1, (float)step), bc2s = sqrtf(1.f - powf(b2, (float)step));
  hipLaunchKernelGGL(vg_adamw_kernel, dim3(nblk(n / 4)), dim3(256), 0, st, p, g, m, v, shadow, n, lr, b1, b2, eps, wd, bc1, bc2s, gscale, step_dev);
  return (int)hipGetLastError();
}
int vg_cast_f32_bf16_launch(const float* src, bf16* dst, long long n, hipStream_t st) {
  if (n & 3) return -3;
  hipLaunchKernelGGL(vg_cast_f32_bf16_kernel, dim3(nblk(n / 4)), dim3(256), 0, st, src, dst, n);
  return (int)hipGetLastError();
}
int vg_slab_reduce_launch(const float* slab, long long stride, int nslab, float* dst, long long n, int accumulate, hipStream_t st) {
  if (n & 3) return -3;
  hipLaunchKernelGGL(vg_slab_reduce_kernel, dim3(nblk(n / 4)), dim3(256), 0, st, slab, stride, nslab, dst, n, accumulate);
  return (int)hipGetLastError();
}
int vg_slab_reduce2_launch(const float* slab0, const float* slab1, long long stride, int nslab, float* dst0, float* dst1, long long n, int accumulate,
                           hipStream_t st) {
  if (n & 3) return -3;
  hipLaunchKernelGGL(vg_slab_reduce2_kernel, dim3(nblk(n / 4), 2), dim3(256), 0, st, slab0, slab1, stride, nslab, dst0, dst1, n, accumulate);
  return (int)hipGetLastError();
}
int vg_add_table_launch(bf16* x, const float* table, long long rows, int E, int period, hipStream_t st) {
  if ((E & 3) || period < 1) return -3;
  const long long n = rows * E;
  hipLaunchKernelGGL(vg_add_table_kernel, dim3(nblk(n / 4)), dim3(256), 0, st, x, table, n, E, period);
  return (int)hipGetLastError();
}
int vg_sin_grad_launch(const bf16* dy, const float* z, bf16* dz, long long n, float w0, hipStream_t st) {
  if (n & 3) return -3;
  hipLaunchKernelGGL(vg_sin_grad_kernel, dim3(nblk(n / 4)), dim3(256), 0, st, dy, z, dz, n, w0);
  return (int)hipGetLastError();
}
