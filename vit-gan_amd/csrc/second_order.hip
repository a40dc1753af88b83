// Second-order kernels: the derivative OF the backward operators, which the gradient penalty of the reference's
// Wasserstein step needs (src/v2/utils.py:124-144: d/d theta of ||d D(x)/d x||).  With these every
// backward operator of the discriminator has its own backward; the GEMM-shaped second-order terms reuse the GEMM family
// (the backward of dX = dY W is d(dY) = ddX W^T and dW = dY^T ddX).  They serve the penalty term only - a few launches per
// step beside the fused passes - so the elementwise ones are written for clarity and determinism (fp32 math, no atomics); the
// attention double backward, which was 18 % of the penalty step as fp32 FMA loops, runs on the MFMA pipe since round 3 (attention.hip).
#include "vg_common.h"
// the MFMA form of the attention double backward lives with the attention kernels (attention.hip)
int vg_attn_bwd_bwd_mfma_launch(const bf16* qkv, const bf16* d_o, const float* lse, const bf16* uqkv, bf16* d_do, bf16* d_qkv, int B, int H,
                                int S, int HE, float scale, hipStream_t st);

// ---------------------------------------------------------------------------------------------------------------------
// activations: f = gelu (mode 1, exact erf) or tanh (mode 3) on a stored bf16 pre-activation h
//   fwd      y  = f(h)
//   bwd      dh = dy f'(h)
//   bwd_bwd  given u = dL/d(dh):  d(dy) = u f'(h),   d(h) = u dy f''(h)
// gelu'(h) = Phi(h) + h phi(h);  gelu''(h) = phi(h) (2 - h^2);  tanh' = 1 - t^2;  tanh'' = -2 t (1 - t^2)
__device__ __forceinline__ void act_derivs(int mode, float h, float& f, float& d1, float& d2) {
  if (mode == 1) {
    float Phi, e;
    vg_phi_e(h, Phi, e);
    const float phi = 0.39894228040143268f * e;
    f = h * Phi; d1 = Phi + h * phi; d2 = phi * (2.0f - h * h);
  } else {
    const float t = vg_tanh(h);
    f = t; d1 = 1.0f - t * t; d2 = -2.0f * t * d1;
  }
}
__global__ __launch_bounds__(256) void vg_act2_kernel(const bf16* __restrict__ h, const bf16* __restrict__ dy, const bf16* __restrict__ u,
                                                      bf16* __restrict__ o0, bf16* __restrict__ o1, long long n, int mode, int what) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const bf16x4 hv = *(const bf16x4*)(h + i4);
  bf16x4 dv = {0, 0, 0, 0}, uv = {0, 0, 0, 0};
  if (what >= 1) dv = *(const bf16x4*)(dy + i4);
  if (what == 2) uv = *(const bf16x4*)(u + i4);
  bf16x4 a, b;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float f, d1, d2;
    act_derivs(mode, vg_bf2f(hv[j]), f, d1, d2);
    if (what == 0) a[j] = vg_f2bf(f);
    else if (what == 1) a[j] = vg_f2bf(vg_bf2f(dv[j]) * d1);
    else { a[j] = vg_f2bf(vg_bf2f(uv[j]) * d1); b[j] = vg_f2bf(vg_bf2f(uv[j]) * vg_bf2f(dv[j]) * d2); }
  }
  *(bf16x4*)(o0 + i4) = a;
  if (what == 2) *(bf16x4*)(o1 + i4) = b;
}
int vg_act2_launch(const bf16* h, const bf16* dy, const bf16* u, bf16* o0, bf16* o1, long long n, int mode, int what, hipStream_t st) {
  if ((n & 3) || (mode != 1 && mode != 3) || what < 0 || what > 2) return -3;
  hipLaunchKernelGGL(vg_act2_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, h, dy, u, o0, o1, n, mode, what);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm backward's backward.  LN backward: g = dy*gamma, a = mean(g), b = mean(g*xh), dx = r (g - a - xh b) with
// xh = (x - mu) r, r = rstd.  Given u = dL/d(dx) (one row each):
//   d(g)  = r (u - mean(u) - xh mean(u xh))                      -> d(dy) = d(g) gamma,  d(gamma) += d(g) dy  (per column)
//   d(xh) = -r (g mean(u xh) + b u)          (explicit)           d(r) = <u, dx> / r =: c   (explicit)
//   d(x)  = r (d(xh) - mean(d(xh)) - xh mean(d(xh) xh)) - r^2 xh c / E
// One wave per row (E <= 1024: up to 16 elements per lane), fp32 math; d(gamma) leaves as one partial row per workgroup.
// NE = E / 64 elements per lane, a compile-time constant: with the runtime count of the first version (16-element arrays behind
// `if (i < NE)`) the kernel held 256 registers and 464 bytes of scratch at one wave per SIMD - 292 us per launch over 16 640 rows.
template <int NE>
__global__ __launch_bounds__(256) void vg_ln_bwd_bwd_kernel(const bf16* __restrict__ u, const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, bf16* __restrict__ d_dy, bf16* __restrict__ d_x,
                                                            float* __restrict__ part, int R) {
  constexpr int E = 64 * NE;
  __shared__ float red[4][E];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float accg[NE], gam[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) { accg[i] = 0.f; gam[i] = gamma[lane + 64 * i]; }
  constexpr float invE = 1.0f / (float)E;
  for (int row = blockIdx.x * 4 + wv; row < R; row += gridDim.x * 4) {
    const float mu = mean[row], r = rstd[row];
    float xh[NE], g[NE], uu[NE], dyv[NE];
    float s_u = 0.f, s_ux = 0.f, s_g = 0.f, s_gx = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const size_t o = (size_t)row * E + lane + 64 * i;
      xh[i] = (vg_bf2f(x[o]) - mu) * r;
      dyv[i] = vg_bf2f(dy[o]);
      uu[i] = vg_bf2f(u[o]);
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      g[i] = dyv[i] * gam[i];
      s_u += uu[i]; s_ux += uu[i] * xh[i]; s_g += g[i]; s_gx += g[i] * xh[i];
    }
    s_u = vg_wave_sum(s_u) * invE; s_ux = vg_wave_sum(s_ux) * invE; s_g = vg_wave_sum(s_g) * invE; s_gx = vg_wave_sum(s_gx) * invE;
    // c = <u, dx> / r = sum u (g - a - xh b)
    float c = 0.f, s_d = 0.f, s_dx = 0.f;
    float dxh[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      c += uu[i] * (g[i] - s_g - xh[i] * s_gx);
      dxh[i] = -r * (g[i] * s_ux + s_gx * uu[i]);
      s_d += dxh[i]; s_dx += dxh[i] * xh[i];
    }
    c = vg_wave_sum(c); s_d = vg_wave_sum(s_d) * invE; s_dx = vg_wave_sum(s_dx) * invE;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const size_t o = (size_t)row * E + lane + 64 * i;
      const float dg = r * (uu[i] - s_u - xh[i] * s_ux);
      d_dy[o] = vg_f2bf(dg * gam[i]);
      accg[i] += dg * dyv[i];
      d_x[o] = vg_f2bf(r * (dxh[i] - s_d - xh[i] * s_dx) - r * r * xh[i] * c * invE);
    }
  }
#pragma unroll
  for (int i = 0; i < NE; ++i) red[wv][lane + 64 * i] = accg[i];
  __syncthreads();
  for (int c = threadIdx.x; c < E; c += 256) part[(size_t)blockIdx.x * E + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
// one partial row of d(gamma) per workgroup.  2048, not 256: the kernel is a chain of row reductions, latency-bound - at 256 workgroups (one wave per
// SIMD) a launch over 16 640 rows took 292 us for 64 MB; with 8 workgroups per CU the other waves cover the chain
int vg_ln_bwd_bwd_nparts(int R) { const int n = (R + 3) / 4; return n < 2048 ? n : 2048; }
int vg_ln_bwd_bwd_launch(const bf16* u, const bf16* dy, const bf16* x, const float* mean, const float* rstd, const float* gamma,
                         bf16* d_dy, bf16* d_x, float* part, int R, int E, hipStream_t st) {
  if ((E & 63) || E > 1024 || R < 1) return -3;
#define VG_LNBB(NE_) hipLaunchKernelGGL((vg_ln_bwd_bwd_kernel<NE_>), dim3(vg_ln_bwd_bwd_nparts(R)), dim3(256), 0, st, u, dy, x, mean, rstd, gamma, d_dy, d_x, part, R)
  // every width the first-order LayerNorm and vg_vit_layout accept (E % 128 == 0, E <= 1024): the penalty step must not refuse a
  // network the plain step trains (640 and 896 were missing)
  switch (E / 64) {
    case 2: VG_LNBB(2); break;
    case 4: VG_LNBB(4); break;
    case 6: VG_LNBB(6); break;
    case 8: VG_LNBB(8); break;
    case 10: VG_LNBB(10); break;
    case 12: VG_LNBB(12); break;
    case 14: VG_LNBB(14); break;
    case 16: VG_LNBB(16); break;
    default: return -3;
  }
#undef VG_LNBB
  return (int)hipGetLastError();
}

#ifdef VG_ABB_FMA
// ---------------------------------------------------------------------------------------------------------------------
// Attention backward's backward, round 2's form (A/B builds only: make var SRC=second_order DEFS=-DVG_ABB_FMA; the product runs
// vg_attn_bwd_bwd2_kernel of attention.hip, 18 x faster, same formulas).  One workgroup per (image, head).  Forward: P = softmax(s Q K^T), O = P V.  Backward:
//   dV = P^T dO ; dP = dO V^T ; delta_i = sum_j P_ij dP_ij ; dS = P (dP - delta) ; dQ = s dS K ; dK = s dS^T Q.
// Given (uQ, uK, uV) = dL/d(dQ, dK, dV):
//   G = s (uQ K^T + Q uK^T) ; gam_i = sum_j G_ij P_ij ; H = P (G - gam)                       [dL/d dP]
//   Pi = dO uV^T + G (dP - delta) - gam dP ; pi_i = sum_j P_ij Pi_ij ; Sg = P (Pi - pi)       [dL/d S]
//   d(dO) = P uV + H V ;  d(Q) = s (dS uK + Sg K) ;  d(K) = s (dS^T uQ + Sg^T Q) ;  d(V) = H^T dO.
// LDS: the seven [S x HE] operands as bf16 (what they are in HBM) and four S x S matrices (P, A = dP - delta, G then Pi, H)
// as fp32: 155 KB for S = 65, HE = 96 - one workgroup per CU.  Plain FMA loops (see the file header).
template <int HE>
__global__ __launch_bounds__(256) void vg_attn_bwd_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                              const bf16* __restrict__ uqkv, bf16* __restrict__ d_do, bf16* __restrict__ d_qkv,
                                                              int S, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
  const int SP = S;  // rows
  bf16* Q = (bf16*)smem2; bf16* K = Q + SP * HE; bf16* V = K + SP * HE; bf16* DO = V + SP * HE;
  bf16* UQ = DO + SP * HE; bf16* UK = UQ + SP * HE; bf16* UV = UK + SP * HE;
  float* P = (float*)(UV + SP * HE);        // [S][S]
  float* A = P + S * S;                     // dP, then dP - delta
  float* G = A + S * S;                     // G, then Pi
  float* Hm = G + S * S;                    // H
  float* dl = Hm + S * S;                   // delta[S], later pi[S]
  float* gm = dl + S;                       // gam[S]
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int E = H * HE, tid = threadIdx.x, nt = blockDim.x;
  const size_t ld = 3 * (size_t)E;
  const bf16* qb = qkv + (size_t)b * S * ld + h * HE;
  const bf16* ub = uqkv + (size_t)b * S * ld + h * HE;
  const bf16* dob = d_o + (size_t)b * S * E + h * HE;
  for (int i = tid; i < S * HE; i += nt) {
    const int r = i / HE, d = i - r * HE;
    Q[i] = qb[r * ld + d]; K[i] = qb[r * ld + E + d]; V[i] = qb[r * ld + 2 * E + d];
    UQ[i] = ub[r * ld + d]; UK[i] = ub[r * ld + E + d]; UV[i] = ub[r * ld + 2 * E + d];
    DO[i] = dob[(size_t)r * E + d];
  }
  __syncthreads();
  const float* lb = lse + ((size_t)b * H + h) * S;
  // pass 1: P, dP, G  (one (i, j) entry per thread step)
  for (int e = tid; e < S * S; e += nt) {
    const int i = e / S, j = e - i * S;
    float s = 0.f, dp = 0.f, g = 0.f;
    for (int d = 0; d < HE; ++d) {
      const float q = vg_bf2f(Q[i * HE + d]), k = vg_bf2f(K[j * HE + d]);
      s += q * k;
      dp += vg_bf2f(DO[i * HE + d]) * vg_bf2f(V[j * HE + d]);
      g += vg_bf2f(UQ[i * HE + d]) * k + q * vg_bf2f(UK[j * HE + d]);
    }
    P[e] = __expf(s * scale - lb[i]);
    A[e] = dp;
    G[e] = g * scale;
  }
  __syncthreads();
  for (int i = tid; i < S; i += nt) {  // row sums: delta_i = sum_j P dP, gam_i = sum_j P G
    float d = 0.f, gmm = 0.f;
    for (int j = 0; j < S; ++j) { d += P[i * S + j] * A[i * S + j]; gmm += P[i * S + j] * G[i * S + j]; }
    dl[i] = d; gm[i] = gmm;
  }
  __syncthreads();
  // pass 2: H = P (G - gam);  Pi = dO uV^T + G (dP - delta) - gam dP;  A <- dP - delta,  G <- Pi
  for (int e = tid; e < S * S; e += nt) {
    const int i = e / S, j = e - i * S;
    float t = 0.f;
    for (int d = 0; d < HE; ++d) t += vg_bf2f(DO[i * HE + d]) * vg_bf2f(UV[j * HE + d]);
    const float dp = A[e], a = dp - dl[i], g = G[e];
    Hm[e] = P[e] * (g - gm[i]);
    A[e] = a;
    G[e] = t + g * a - gm[i] * dp;
  }
  __syncthreads();
  for (int i = tid; i < S; i += nt) {  // pi_i = sum_j P Pi  (delta is no longer needed: overwrite)
    float s = 0.f;
    for (int j = 0; j < S; ++j) s += P[i * S + j] * G[i * S + j];
    dl[i] = s;
  }
  __syncthreads();
  // outputs.  dS = P A ;  Sg = P (Pi - pi)
  bf16* ddo = d_do + (size_t)b * S * E + h * HE;
  bf16* dq = d_qkv + (size_t)b * S * ld + h * HE;
  for (int e = tid; e < S * HE; e += nt) {   // d(dO)_id = sum_j P_ij uV_jd + H_ij V_jd ;  d(Q)_id = s sum_j dS_ij uK_jd + Sg_ij K_jd
    const int i = e / HE, d = e - i * HE;
    float o1 = 0.f, o2 = 0.f;
    for (int j = 0; j < S; ++j) {
      const float p = P[i * S + j];
      o1 += p * vg_bf2f(UV[j * HE + d]) + Hm[i * S + j] * vg_bf2f(V[j * HE + d]);
      o2 += p * A[i * S + j] * vg_bf2f(UK[j * HE + d]) + p * (G[i * S + j] - dl[i]) * vg_bf2f(K[j * HE + d]);
    }
    ddo[(size_t)i * E + d] = vg_f2bf(o1);
    dq[(size_t)i * ld + d] = vg_f2bf(o2 * scale);
  }
  for (int e = tid; e < S * HE; e += nt) {   // d(K)_jd = s sum_i dS_ij uQ_id + Sg_ij Q_id ;  d(V)_jd = sum_i H_ij dO_id
    const int j = e / HE, d = e - j * HE;
    float o1 = 0.f, o2 = 0.f;
    for (int i = 0; i < S; ++i) {
      const float p = P[i * S + j];
      o1 += p * A[i * S + j] * vg_bf2f(UQ[i * HE + d]) + p * (G[i * S + j] - dl[i]) * vg_bf2f(Q[i * HE + d]);
      o2 += Hm[i * S + j] * vg_bf2f(DO[i * HE + d]);
    }
    dq[(size_t)j * ld + E + d] = vg_f2bf(o1 * scale);
    dq[(size_t)j * ld + 2 * E + d] = vg_f2bf(o2);
  }
}
#endif
int vg_attn_bwd_bwd_launch(const bf16* qkv, const bf16* d_o, const float* lse, const bf16* uqkv, bf16* d_do, bf16* d_qkv, int B, int H,
                           int S, int HE, float scale, hipStream_t st) {
  if (S < 1 || S > 80 || B < 1 || H < 1) return -2;
#ifndef VG_ABB_FMA
  return vg_attn_bwd_bwd_mfma_launch(qkv, d_o, lse, uqkv, d_do, d_qkv, B, H, S, HE, scale, st);  // attention.hip: on the MFMA pipe since round 3
#else
  const size_t lds = (size_t)7 * S * HE * 2 + (size_t)4 * S * S * 4 + (size_t)2 * S * 4;  // 155 KB at S = 65, HE = 96
  if (lds > 160 * 1024) return -3;
#define VG_ABB(HE_)                                                                                                           \
  do {                                                                                                                        \
    hipError_t e = hipFuncSetAttribute((const void*)vg_attn_bwd_bwd_kernel<HE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return (int)e;                                                                                       \
    hipLaunchKernelGGL((vg_attn_bwd_bwd_kernel<HE_>), dim3(B * H), dim3(256), lds, st, qkv, d_o, lse, uqkv, d_do, d_qkv, S, H, scale); \
  } while (0)
  if (HE == 96) VG_ABB(96); else if (HE == 64) VG_ABB(64); else if (HE == 32) VG_ABB(32); else return -3;
#undef VG_ABB
  return (int)hipGetLastError();
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Small kernels of the gradient penalty as ONE C call (vg_vit_penalty, engine.hip; reference: src/v2/utils.py:124-144).
// ---------------------------------------------------------------------------------------------------------------------
// interpolated = eps * real + (1 - eps) * fake, fp32 like the reference forms it (utils.py:130); eps [B]
__global__ __launch_bounds__(256) void vg_pen_interp_kernel(const bf16* __restrict__ real, const bf16* __restrict__ fake, const float* __restrict__ eps,
                                                            float* __restrict__ out, long long per, long long n) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const float e = eps[i4 / per], f = 1.0f - e;
  const bf16x4 r = *(const bf16x4*)(real + i4), k = *(const bf16x4*)(fake + i4);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = __fadd_rn(__fmul_rn(e, vg_bf2f(r[j])), __fmul_rn(f, vg_bf2f(k[j])));
  *(f32x4*)(out + i4) = o;
}
int vg_pen_interp_launch(const bf16* real, const bf16* fake, const float* eps, float* out, int B, long long per, hipStream_t st) {
  if (B < 1 || per < 4 || (per & 3)) return -3;
  const long long n = (long long)B * per;
  hipLaunchKernelGGL(vg_pen_interp_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, real, fake, eps, out, per, n);
  return (int)hipGetLastError();
}
// One workgroup per image: n_b = ||g_b||_2 over the image's `per` gradient elements (the patch rows of d A - the same numbers as the
// image gradient, permuted), pen_img[b] = (n_b - 1)^2 / B, and the direction the second backward starts from,
// u_b = coef (n_b - 1) / n_b g_b = d(weight * mean_b (n_b - 1)^2) / d g_b with coef = 2 weight / B  (utils.py:143-144).
__global__ __launch_bounds__(256) void vg_pen_norm_kernel(const bf16* __restrict__ g, bf16* __restrict__ u, float* __restrict__ pen_img, long long per,
                                                          float coef, float inv_b) {
  __shared__ float red[4];
  __shared__ float fac;
  const bf16* gb = g + (size_t)blockIdx.x * per;
  bf16* ub = u + (size_t)blockIdx.x * per;
  float a = 0.f;
  for (long long i = (long long)threadIdx.x * 4; i < per; i += 1024) {
    const bf16x4 v = *(const bf16x4*)(gb + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float x = vg_bf2f(v[j]); a = fmaf(x, x, a); }
  }
  a = vg_wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nb = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    pen_img[blockIdx.x] = (nb - 1.0f) * (nb - 1.0f) * inv_b;
    fac = nb > 0.f ? coef * (nb - 1.0f) / nb : 0.f;  // (torch's norm has the zero subgradient at 0)
  }
  __syncthreads();
  const float f = fac;
  for (long long i = (long long)threadIdx.x * 4; i < per; i += 1024) {
    const bf16x4 v = *(const bf16x4*)(gb + i);
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = vg_f2bf(f * vg_bf2f(v[j]));
    *(bf16x4*)(ub + i) = o;
  }
}
__global__ __launch_bounds__(256) void vg_pen_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += part[i];
  a = vg_wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}
int vg_pen_norm_launch(const bf16* g, bf16* u, float* pen_img, float* pen_out, int B, long long per, float weight, hipStream_t st) {
  if (B < 1 || per < 4 || (per & 3)) return -3;
  hipLaunchKernelGGL(vg_pen_norm_kernel, dim3(B), dim3(256), 0, st, g, u, pen_img, per, 2.0f * weight / (float)B, 1.0f / (float)B);
  hipLaunchKernelGGL(vg_pen_sum_kernel, dim3(1), dim3(256), 0, st, pen_img, B, pen_out);
  return (int)hipGetLastError();
}
// Classifier head, second order: the first backward is g_pre = g_t (1 - t^2) with g_t[e] = sum_k W2[k, e] (grad_outputs = ones) and t = tanh(p).
// Given u = dL/d g_pre:  u_gt = u (1 - t^2)  (its batch sum is dL/dW2[k, :], every k)  and  s_p = u g_t (-2 t (1 - t^2)) = dL/dp.
__global__ __launch_bounds__(256) void vg_pen_head2_kernel(const bf16* __restrict__ u, const bf16* __restrict__ t, const float* __restrict__ W2,
                                                           bf16* __restrict__ u_gt, bf16* __restrict__ s_p, int B, int E, int Kc) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * E) return;
  const int e = (int)(i % E);
  float gt = 0.f;
  for (int k = 0; k < Kc; ++k) gt += W2[(size_t)k * E + e];
  gt = vg_bf2f(vg_f2bf(gt));  // (the operator chain hands g_t on as a bf16 tensor)
  const float tv = vg_bf2f(t[i]), uv = vg_bf2f(u[i]);
  const float d1 = 1.0f - tv * tv;
  u_gt[i] = vg_f2bf(uv * d1);
  s_p[i] = vg_f2bf(uv * gt * (-2.0f * tv * d1));
}
int vg_pen_head2_launch(const bf16* u, const bf16* t, const float* W2, bf16* u_gt, bf16* s_p, int B, int E, int Kc, hipStream_t st) {
  if (B < 1 || E < 1 || Kc < 1) return -3;
  hipLaunchKernelGGL(vg_pen_head2_kernel, dim3((unsigned)(((long long)B * E + 255) / 256)), dim3(256), 0, st, u, t, W2, u_gt, s_p, B, E, Kc);
  return (int)hipGetLastError();
}
// out = a + b (one rounding; out may alias a): where the second backward meets a gradient the double backward injected
__global__ __launch_bounds__(256) void vg_add_bf16_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ out, long long n) {
  const long long i8 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i8 >= n) return;
  const bf16x8 x = *(const bf16x8*)(a + i8), y = *(const bf16x8*)(b + i8);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = vg_f2bf(vg_bf2f(x[j]) + vg_bf2f(y[j]));
  *(bf16x8*)(out + i8) = o;
}
int vg_add_bf16_launch(const bf16* a, const bf16* b, bf16* out, long long n, hipStream_t st) {
  if (n < 8 || (n & 7)) return -3;
  hipLaunchKernelGGL(vg_add_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, a, b, out, n);
  return (int)hipGetLastError();
}
__global__ __launch_bounds__(256) void vg_fill_f32_kernel(float* __restrict__ p, long long n, float v) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}
int vg_fill_f32_launch(float* p, long long n, float v, hipStream_t st) {
  if (n < 1) return -3;
  hipLaunchKernelGGL(vg_fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
  return (int)hipGetLastError();
}
