// LayerNorm / self-modulated LayerNorm (SLN) forward + backward and the small deterministic
// reductions that go with them.  gfx950, wave64: one wave per row, E/64 elements per lane.
//
// Backward kernels never use float atomics: every workgroup writes one row of partial column
// sums ([n_wg][part_width] fp32) and vg_colsum_f32 folds them in a fixed order, so results are
// bitwise reproducible run to run.
#include "vg_common.h"

#define LN_MAX_PER_LANE 16  // E <= 1024
#define LN_MAX_PARTS 512    // partial rows written by the backward kernels (fixed upper bound)

// ------------------------------------------------------------------------------------------
// y = LN(x) * gamma + beta ; stats saved for backward.  Row r of x at x + r*xs (elements).
__global__ __launch_bounds__(256) void vg_ln_fwd_kernel(const bf16* __restrict__ x, long long xs,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        bf16* __restrict__ y, long long ys, float* __restrict__ mean,
                                                        float* __restrict__ rstd, int R, int E, float eps) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= R) return;
  const int npl = E >> 7;  // bf16x2 per lane
  const bf16x2* xr = (const bf16x2*)(x + (size_t)row * xs);
  float v[LN_MAX_PER_LANE];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE / 2; ++i)
    if (i < npl) {
      const bf16x2 t = xr[lane + 64 * i];
      v[2 * i] = vg_bf2f(t[0]); v[2 * i + 1] = vg_bf2f(t[1]);
      s += v[2 * i] + v[2 * i + 1];
    }
  const float mu = vg_wave_sum(s) / (float)E;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE / 2; ++i)
    if (i < npl) {
      const float a = v[2 * i] - mu, c = v[2 * i + 1] - mu;
      q += a * a + c * c;
    }
  const float rs = rsqrtf(vg_wave_sum(q) / (float)E + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  bf16x2* yr = (bf16x2*)(y + (size_t)row * ys);
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE / 2; ++i)
    if (i < npl) {
      const int c = 2 * (lane + 64 * i);
      bf16x2 o;
      o[0] = vg_f2bf((v[2 * i] - mu) * rs * gamma[c] + beta[c]);
      o[1] = vg_f2bf((v[2 * i + 1] - mu) * rs * gamma[c + 1] + beta[c + 1]);
      yr[lane + 64 * i] = o;
    }
}

// ------------------------------------------------------------------------------------------
// SLN forward (src/v1/spectral_layer_norm.py:19-20): out = w * (gs * (LN(h)*lw + lb) + bs)
// gs / bs are device scalars.  h_bcast_rows > 0: h has only that many rows (the generator's
// learned embedding [T,E], broadcast over the batch): row r reads h row r % h_bcast_rows.
__global__ __launch_bounds__(256) void vg_sln_fwd_kernel(const bf16* __restrict__ h, int h_bcast_rows,
                                                         const bf16* __restrict__ wmod, const float* __restrict__ lw,
                                                         const float* __restrict__ lb, const float* __restrict__ gs,
                                                         const float* __restrict__ bs, bf16* __restrict__ y,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int R, int E,
                                                         float eps) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= R) return;
  const int npl = E >> 7;
  const int hrow = h_bcast_rows > 0 ? row % h_bcast_rows : row;
  const bf16x2* xr = (const bf16x2*)(h + (size_t)hrow * E);
  const bf16x2* wr = (const bf16x2*)(wmod + (size_t)row * E);
  float v[LN_MAX_PER_LANE];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE / 2; ++i)
    if (i < npl) {
      const bf16x2 t = xr[lane + 64 * i];
      v[2 * i] = vg_bf2f(t[0]); v[2 * i + 1] = vg_bf2f(t[1]);
      s += v[2 * i] + v[2 * i + 1];
    }
  const float mu = vg_wave_sum(s) / (float)E;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE / 2; ++i)
    if (i < npl) {
      const float a = v[2 * i] - mu, c = v[2 * i + 1] - mu;
      q += a * a + c * c;
    }
  const float rs = rsqrtf(vg_wave_sum(q) / (float)E + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  const float g_s = gs[0], b_s = bs[0];
  bf16x2* yr = (bf16x2*)(y + (size_t)row * E);
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE / 2; ++i)
    if (i < npl) {
      const int c = 2 * (lane + 64 * i);
      const bf16x2 wm = wr[lane + 64 * i];
      bf16x2 o;
      o[0] = vg_f2bf(vg_bf2f(wm[0]) * (g_s * ((v[2 * i] - mu) * rs * lw[c] + lb[c]) + b_s));
      o[1] = vg_f2bf(vg_bf2f(wm[1]) * (g_s * ((v[2 * i + 1] - mu) * rs * lw[c + 1] + lb[c + 1]) + b_s));
      yr[lane + 64 * i] = o;
    }
}

// ------------------------------------------------------------------------------------------
// LN / SLN backward.  One workgroup = LN_ROWS_PER_WG rows; per-lane column accumulators are
// folded across the 4 waves through LDS and written as ONE partial row:
//   part[wg][0:E]      = sum_rows dy_eff * xhat        (d gamma / d lw)
//   part[wg][E:2E]     = sum_rows dy_eff               (d beta  / d lb)
//   part[wg][2E:3E]    = sum_rows dx_out               (bias grad of the Linear feeding the residual)
//   part[wg][3E], [3E+1] = SLN scalars d gs, d bs      (SLN only; width 3E+64)
// dx_out = (gres ? gres : 0) + LN-backward(dy_eff).
// SLN: dy_eff = dy * w * gs;  dw_acc (+)= dy * (gs*(xhat*lw+lb)+bs)  (fp32 accumulator [R,E]).
template <bool SLN, int NPL>
__global__ __launch_bounds__(256) void vg_ln_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                        int x_bcast_rows, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ lbias, const bf16* __restrict__ gres,
                                                        bf16* __restrict__ dx, float* __restrict__ part, int part_w,
                                                        const bf16* __restrict__ wmod, const float* __restrict__ gs,
                                                        const float* __restrict__ bs, float* __restrict__ dw_acc,
                                                        int dw_accumulate, int R, int E, bf16* __restrict__ dxm,
                                                        unsigned dthr, unsigned dkey0, float dscale,
                                                        const unsigned* __restrict__ dstep) {
  const unsigned dkey = vg_drop_key(dkey0, dstep);
  __shared__ float red[4 * 64 * LN_MAX_PER_LANE];  // [wave][E]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float ag[2 * NPL], ab[2 * NPL], ac[2 * NPL];
#pragma unroll
  for (int i = 0; i < 2 * NPL; ++i) { ag[i] = 0.f; ab[i] = 0.f; ac[i] = 0.f; }
  float s_gs = 0.f, s_bs = 0.f;
  const float g_s = SLN ? gs[0] : 1.f, b_s = SLN ? bs[0] : 0.f;
  // rows are dealt to (workgroup, wave) pairs round-robin: wave w of block b takes rows
  // 4*b + w, 4*b + w + 4*gridDim.x, ...  (fixed assignment -> deterministic partial sums)
  for (int row = blockIdx.x * 4 + w; row < R; row += 4 * gridDim.x) {
    const int xrow = x_bcast_rows > 0 ? row % x_bcast_rows : row;
    const bf16x2* xr = (const bf16x2*)(x + (size_t)xrow * E);
    const bf16x2* dr = (const bf16x2*)(dy + (size_t)row * E);
    const float mu = mean[row], rs = rstd[row];
    float xh[2 * NPL], gg[2 * NPL];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i)
      {
        const int c = 2 * (lane + 64 * i);
        const bf16x2 xv = xr[lane + 64 * i];
        const bf16x2 dv = dr[lane + 64 * i];
        float d0 = vg_bf2f(dv[0]), d1 = vg_bf2f(dv[1]);
        const float h0 = (vg_bf2f(xv[0]) - mu) * rs, h1 = (vg_bf2f(xv[1]) - mu) * rs;
        if (SLN) {
          const bf16x2 wv = ((const bf16x2*)(wmod + (size_t)row * E))[lane + 64 * i];
          const float w0 = vg_bf2f(wv[0]), w1 = vg_bf2f(wv[1]);
          const float l0 = h0 * gamma[c] + lbias[c], l1 = h1 * gamma[c + 1] + lbias[c + 1];
          float* dwp = dw_acc + (size_t)row * E + c;
          const float t0 = d0 * (g_s * l0 + b_s), t1 = d1 * (g_s * l1 + b_s);
          if (dw_accumulate) { dwp[0] += t0; dwp[1] += t1; } else { dwp[0] = t0; dwp[1] = t1; }
          s_gs += d0 * w0 * l0 + d1 * w1 * l1;
          s_bs += d0 * w0 + d1 * w1;
          d0 *= w0 * g_s; d1 *= w1 * g_s;
        }
        xh[2 * i] = h0; xh[2 * i + 1] = h1;
        ag[2 * i] += d0 * h0; ag[2 * i + 1] += d1 * h1;
        ab[2 * i] += d0; ab[2 * i + 1] += d1;
        const float g0 = d0 * gamma[c], g1 = d1 * gamma[c + 1];
        gg[2 * i] = g0; gg[2 * i + 1] = g1;
        c1 += g0 + g1;
        c2 += g0 * h0 + g1 * h1;
      }
    c1 = vg_wave_sum(c1) / (float)E;
    c2 = vg_wave_sum(c2) / (float)E;
    bf16x2* oxr = (bf16x2*)(dx + (size_t)row * E);
#pragma unroll
    for (int i = 0; i < NPL; ++i)
      {
        float o0 = rs * (gg[2 * i] - c1 - xh[2 * i] * c2);
        float o1 = rs * (gg[2 * i + 1] - c1 - xh[2 * i + 1] * c2);
        if (gres) {
          const bf16x2 rv = ((const bf16x2*)(gres + (size_t)row * E))[lane + 64 * i];
          o0 += vg_bf2f(rv[0]); o1 += vg_bf2f(rv[1]);
        }
        bf16x2 o; o[0] = vg_f2bf(o0); o[1] = vg_f2bf(o1);
        oxr[lane + 64 * i] = o;
        if (dxm) {  // gradient entering the dropped branch: dx * mask / keep  (same mask as the forward epilogue)
          const unsigned idx = (unsigned)row * (unsigned)E + 2u * (lane + 64 * i);
          const unsigned wd = vg_drop_word(dkey, idx >> 2);
          bf16x2 om;
          om[0] = vg_f2bf(vg_bf2f(o[0]) * vg_drop_factor(wd, idx & 3, dthr, dscale));
          om[1] = vg_f2bf(vg_bf2f(o[1]) * vg_drop_factor(wd, (idx & 3) + 1, dthr, dscale));
          ((bf16x2*)(dxm + (size_t)row * E))[lane + 64 * i] = om;
          o = om;
        }
        ac[2 * i] += vg_bf2f(o[0]); ac[2 * i + 1] += vg_bf2f(o[1]);
      }
  }
  // fold the 4 waves in a fixed order through LDS (reuse `red` as [4][E] three times)
  float* rp = red;
  float* out = part + (size_t)blockIdx.x * part_w;
#pragma unroll 1
  for (int which = 0; which < 3; ++which) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPL; ++i)
      {
        const int c = 2 * (lane + 64 * i);
        const float a0 = which == 0 ? ag[2 * i] : (which == 1 ? ab[2 * i] : ac[2 * i]);
        const float a1 = which == 0 ? ag[2 * i + 1] : (which == 1 ? ab[2 * i + 1] : ac[2 * i + 1]);
        rp[w * E + c] = a0; rp[w * E + c + 1] = a1;
      }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256) out[which * E + c] = rp[c] + rp[E + c] + rp[2 * E + c] + rp[3 * E + c];
  }
  if (SLN) {
    s_gs = vg_wave_sum(s_gs); s_bs = vg_wave_sum(s_bs);
    __syncthreads();
    if (lane == 0) { rp[2 * w] = s_gs; rp[2 * w + 1] = s_bs; }
    __syncthreads();
    if (threadIdx.x == 0) {
      out[3 * E] = rp[0] + rp[2] + rp[4] + rp[6];
      out[3 * E + 1] = rp[1] + rp[3] + rp[5] + rp[7];
    }
  }
}

// ------------------------------------------------------------------------------------------
// dst_k[c] (+)= sum_r part[r][off_k + c]  for up to 4 consecutive column segments.
struct VgSeg { float* dst; int n; };
struct VgSegs { VgSeg s[4]; };
__global__ __launch_bounds__(256) void vg_colsum_f32_kernel(const float* __restrict__ part, int rows, int width,
                                                            VgSegs segs, int accumulate) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a = 0.f;
  if (c < width)
    for (int r = rl; r < rows; r += 16) a += part[(size_t)r * width + c];
  red[rl][cl] = a;
  __syncthreads();
  if (rl != 0 || c >= width) return;
  a = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) a += red[k][cl];
  int off = 0; float* dst = nullptr;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (c >= off && c < off + segs.s[k].n) dst = segs.s[k].dst ? segs.s[k].dst + (c - off) : nullptr;
    off += segs.s[k].n;
  }
  if (!dst) return;
  if (accumulate) *dst += a; else *dst = a;
}

// partial column sums of a bf16 matrix: part[chunk][c] = sum over the chunk's rows of X[r][c].
// One workgroup = 256 columns x CS_ROWS rows: thread (cg, rl) loads 16 B (8 columns) of rows rl, rl+8, ...
#define CS_ROWS 256
__global__ __launch_bounds__(256) void vg_colsum_bf16_part_kernel(const bf16* __restrict__ X, long long ld, int R, int N,
                                                                  float* __restrict__ part) {
  __shared__ float red[8][256 + 8];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + cg * 8;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(R, r0 + CS_ROWS);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 8) {
      const bf16x8 t = *(const bf16x8*)(X + (size_t)r * ld + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += vg_bf2f(t[j]);
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = a[j];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
    part[(size_t)blockIdx.y * N + cc] = t;
  }
}

// ---------------------------------- host launchers ----------------------------------------
int vg_ln_fwd_launch(const bf16* x, long long xs, const float* gamma, const float* beta, bf16* y, long long ys,
                     float* mean, float* rstd, int R, int E, float eps, hipStream_t st) {
  if ((E & 127) || E > 64 * LN_MAX_PER_LANE || R < 1) return -3;
  hipLaunchKernelGGL(vg_ln_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, st, x, xs, gamma, beta, y, ys, mean, rstd, R, E, eps);
  return (int)hipGetLastError();
}
int vg_sln_fwd_launch(const bf16* h, int h_bcast_rows, const bf16* wmod, const float* lw, const float* lb,
                      const float* gs, const float* bs, bf16* y, float* mean, float* rstd, int R, int E, float eps,
                      hipStream_t st) {
  if ((E & 127) || E > 64 * LN_MAX_PER_LANE || R < 1) return -3;
  hipLaunchKernelGGL(vg_sln_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, st, h, h_bcast_rows, wmod, lw, lb, gs, bs, y,
                     mean, rstd, R, E, eps);
  return (int)hipGetLastError();
}
int vg_ln_bwd_nparts(int R) { const int n = (R + 7) / 8; return n < LN_MAX_PARTS ? n : LN_MAX_PARTS; }
int vg_ln_bwd_launch(const bf16* dy, const bf16* x, const float* mean, const float* rstd, const float* gamma,
                     const bf16* gres, bf16* dx, float* part, int R, int E, bf16* dxm, unsigned dthr, unsigned dkey,
                     float dscale, const unsigned* dstep, hipStream_t st) {
  if ((E & 127) || E > 64 * LN_MAX_PER_LANE || R < 1) return -3;
#define LN_BWD(NPL_) hipLaunchKernelGGL((vg_ln_bwd_kernel<false, NPL_>), dim3(vg_ln_bwd_nparts(R)), dim3(256), 0, st, dy, x, 0, mean, rstd, gamma, \
                     (const float*)nullptr, gres, dx, part, 3 * E, (const bf16*)nullptr, (const float*)nullptr,                                      \
                     (const float*)nullptr, (float*)nullptr, 0, R, E, dxm, dthr, dkey, dscale, dstep)
  switch (E >> 7) {
    case 1: LN_BWD(1); break; case 2: LN_BWD(2); break; case 3: LN_BWD(3); break; case 4: LN_BWD(4); break;
    case 5: LN_BWD(5); break; case 6: LN_BWD(6); break; case 7: LN_BWD(7); break; case 8: LN_BWD(8); break;
    default: return -3;
  }
#undef LN_BWD
  return (int)hipGetLastError();
}
int vg_sln_bwd_launch(const bf16* dy, const bf16* h, int h_bcast_rows, const bf16* wmod, const float* mean,
                      const float* rstd, const float* lw, const float* lb, const float* gs, const float* bs,
                      const bf16* gres, bf16* dh, float* dw_acc, int dw_accumulate, float* part, int R, int E,
                      bf16* dhm, unsigned dthr, unsigned dkey, float dscale, const unsigned* dstep, hipStream_t st) {
  if ((E & 127) || E > 64 * LN_MAX_PER_LANE || R < 1) return -3;
#define SLN_BWD(NPL_) hipLaunchKernelGGL((vg_ln_bwd_kernel<true, NPL_>), dim3(vg_ln_bwd_nparts(R)), dim3(256), 0, st, dy, h, h_bcast_rows, mean, \
                     rstd, lw, lb, gres, dh, part, 3 * E + 64, wmod, gs, bs, dw_acc, dw_accumulate, R, E, dhm, dthr, dkey, dscale, dstep)
  switch (E >> 7) {
    case 1: SLN_BWD(1); break; case 2: SLN_BWD(2); break; case 3: SLN_BWD(3); break; case 4: SLN_BWD(4); break;
    case 5: SLN_BWD(5); break; case 6: SLN_BWD(6); break; case 7: SLN_BWD(7); break; case 8: SLN_BWD(8); break;
    default: return -3;
  }
#undef SLN_BWD
  return (int)hipGetLastError();
}
int vg_colsum_f32_launch(const float* part, int rows, int width, float* d0, int n0, float* d1, int n1, float* d2, int n2,
                         float* d3, int n3, int accumulate, hipStream_t st) {
  VgSegs s; s.s[0] = {d0, n0}; s.s[1] = {d1, n1}; s.s[2] = {d2, n2}; s.s[3] = {d3, n3};
  hipLaunchKernelGGL(vg_colsum_f32_kernel, dim3((width + 15) / 16), dim3(256), 0, st, part, rows, width, s, accumulate);
  return (int)hipGetLastError();
}
int vg_colsum_bf16_nparts(int R) { return (R + CS_ROWS - 1) / CS_ROWS; }
// dst[c] (+)= sum_r X[r][c]; `part` needs vg_colsum_bf16_nparts(R) * N floats of scratch.
int vg_colsum_bf16_launch(const bf16* X, long long ld, int R, int N, float* part, float* dst, int accumulate,
                          hipStream_t st) {
  if ((N & 7) || (ld & 7) || R < 1) return -3;
  const int chunks = vg_colsum_bf16_nparts(R);
  hipLaunchKernelGGL(vg_colsum_bf16_part_kernel, dim3((N + 255) / 256, chunks), dim3(256), 0, st, X, ld, R, N, part);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  return vg_colsum_f32_launch(part, chunks, N, dst, N, nullptr, 0, nullptr, 0, nullptr, 0, accumulate, st);
}
