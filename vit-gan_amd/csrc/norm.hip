// LayerNorm / self-modulated LayerNorm (SLN) forward + backward and the small deterministic
// reductions that go with them.  gfx950, wave64: 16 lanes per row, 4 rows per wave, 16-byte accesses.
//
// Backward kernels never use float atomics: every workgroup writes one row of partial column
// sums ([n_wg][part_width] fp32) and vg_colsum_f32 folds them in a fixed order, so results are
// bitwise reproducible run to run.
#include "vg_common.h"

// Lanes per row of the backward kernels: 32 = 8-byte accesses but half the per-lane state of 16 (the three column
// accumulators, x_hat and dy*gamma all scale with the columns a lane owns): LayerNorm 210 -> 126 registers (2 -> 4
// waves/SIMD, 22.3 -> 20.3 us), SLN > 256 -> 180 registers (25.6 -> 17.5 us).  The forward kernels keep 16.
#ifndef LN_BWD_LPR
#define LN_BWD_LPR 32
#endif
#ifndef SLN_BWD_LPR
#define SLN_BWD_LPR 32
#endif
#ifndef LN_MAX_PARTS
#define LN_MAX_PARTS 512    // partial rows written by the backward kernels (fixed upper bound)
#endif

// Thread layout of every kernel below: a wave handles 4 rows at a time, 16 lanes per row; lane `sub` of a
// row owns the 16-byte chunks sub, sub+16, ... (NV = E/128 chunks of 8 bf16), so each wave-instruction
// moves 4 x 256 contiguous bytes with 16 B per lane (cdna_hip_programming.md Guideline 13).
__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a row group
  v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
  return v;
}
__device__ __forceinline__ void unpack8(const bf16x8 t, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = vg_bf2f(t[j]);
}

// ------------------------------------------------------------------------------------------
// LN forward:   y = LN(x) * gamma + beta                    (SLN = false; row r of x at x + r*xs)
// SLN forward:  y = w * (gs * (LN(h) * lw + lb) + bs)       (src/v1/spectral_layer_norm.py:19-20; gs/bs device
//               scalars; bcast_rows > 0: h has that many rows, broadcast over the batch)
template <bool SLN, int NV>
__global__ __launch_bounds__(256) void vg_ln_fwd_kernel(const bf16* __restrict__ x, long long xs, int bcast_rows,
                                                        const bf16* __restrict__ wmod, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ gs,
                                                        const float* __restrict__ bs, bf16* __restrict__ y, long long ys,
                                                        float* __restrict__ mean, float* __restrict__ rstd, int R, float eps) {
  constexpr int E = NV * 128;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sub = lane & 15;
  const int row = (blockIdx.x * 4 + wv) * 4 + (lane >> 4);
  const bool ok = row < R;
  const int xrow = ok ? (bcast_rows > 0 ? row % bcast_rows : row) : 0;
  const bf16* xr = x + (size_t)xrow * xs;
  float v[NV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    bf16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) t = *(const bf16x8*)(xr + 8 * (sub + 16 * i));
    unpack8(t, v[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[i][j];
  }
  const float mu = row16_sum(s) * (1.0f / E);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float c = v[i][j] - mu; q += c * c; }
  const float rs = rsqrtf(row16_sum(q) * (1.0f / E) + eps);
  if (!ok) return;
  if (sub == 0) { mean[row] = mu; rstd[row] = rs; }
  const float g_s = SLN ? gs[0] : 1.f, b_s = SLN ? bs[0] : 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 8 * (sub + 16 * i);
    const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
    const f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
    float wm[8];
    if (SLN) unpack8(*(const bf16x8*)(wmod + (size_t)row * E + c), wm);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gg = j < 4 ? g0[j] : g1[j - 4], bb = j < 4 ? b0[j] : b1[j - 4];
      float r = (v[i][j] - mu) * rs * gg + bb;
      if (SLN) r = wm[j] * (g_s * r + b_s);
      o[j] = vg_f2bf(r);
    }
    *(bf16x8*)(y + (size_t)row * ys + c) = o;
  }
}

// ------------------------------------------------------------------------------------------
// LN / SLN backward.  Rows are dealt to (workgroup, wave, row-group) slots round-robin (fixed assignment ->
// deterministic partial sums); per-lane column accumulators are folded across the 4 row groups of a wave by
// shuffles and across the 4 waves through LDS, and written as ONE partial row per workgroup:
//   part[wg][0:E]      = sum_rows dy_eff * xhat        (d gamma / d lw)
//   part[wg][E:2E]     = sum_rows dy_eff               (d beta  / d lb)
//   part[wg][2E:3E]    = sum_rows dx_out (masked copy when dxm != null: bias grad of the Linear feeding the branch)
//   part[wg][3E], [3E+1] = SLN scalars d gs, d bs      (SLN only; width 3E+64)
// dx = (gres ? gres : 0) + LN-backward(dy_eff);  dxm = dx * dropout mask (optional second output).
// SLN: dy_eff = dy * w * gs;  dw_acc (+)= dy * (gs*(xhat*lw+lb)+bs)  (fp32 accumulator [R,E]).
template <int CH> struct ChunkT;
template <> struct ChunkT<8> { typedef bf16x8 bt; };
template <> struct ChunkT<4> { typedef bf16x4 bt; };
template <int CH>
__device__ __forceinline__ void ld_chunk(const bf16* p, bool ok, float (&o)[CH]) {
  typename ChunkT<CH>::bt t;
#pragma unroll
  for (int j = 0; j < CH; ++j) t[j] = (bf16)0.f;
  if (ok) t = *(const typename ChunkT<CH>::bt*)p;
#pragma unroll
  for (int j = 0; j < CH; ++j) o[j] = vg_bf2f(t[j]);
}
template <int CH>
__device__ __forceinline__ void ld_f32(const float* p, float (&o)[CH]) {
#pragma unroll
  for (int q = 0; q < CH / 4; ++q) {
    const f32x4 t = *(const f32x4*)(p + 4 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[4 * q + j] = t[j];
  }
}

// LPR lanes per row (16: 16-byte chunks, 4 rows per wave; 32: 8-byte chunks, 2 rows per wave - half the
// per-lane state, used where the backward's accumulators would otherwise cap occupancy at 1-2 waves/SIMD).
template <bool SLN, int NV, int LPR>
__global__ __launch_bounds__(256) void vg_ln_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                        int x_bcast_rows, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ lbias, const bf16* __restrict__ gres,
                                                        bf16* __restrict__ dx, float* __restrict__ part, int part_w,
                                                        const bf16* __restrict__ wmod, const float* __restrict__ gs,
                                                        const float* __restrict__ bs, float* __restrict__ dw_acc,
                                                        int dw_accumulate, int R, bf16* __restrict__ dxm,
                                                        unsigned dthr, unsigned dkey0, float dscale,
                                                        const unsigned* __restrict__ dstep, int drop_row_mul) {
  constexpr int E = NV * 128, CH = 128 / LPR, RPW = 64 / LPR;
  __shared__ float red[3][4][E];
  __shared__ float reds[4][2];
  const unsigned dkey = vg_drop_key(dkey0, dstep);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sub = lane % LPR, rg = lane / LPR;
  float gam[NV][CH];
#pragma unroll
  for (int i = 0; i < NV; ++i) ld_f32<CH>(gamma + CH * (sub + LPR * i), gam[i]);
  float ag[NV][CH], ab[NV][CH], ac[NV][CH];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < CH; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; ac[i][j] = 0.f; }
  float s_gs = 0.f, s_bs = 0.f;
  const float g_s = SLN ? gs[0] : 1.f, b_s = SLN ? bs[0] : 0.f;
  for (int row0 = (blockIdx.x * 4 + wv) * RPW; row0 < R; row0 += 4 * RPW * gridDim.x) {
    const int row = row0 + rg;
    const bool ok = row < R;
    const int rr = ok ? row : 0;
    // x_bcast_rows > 0: x has that many rows, broadcast over the batch; < 0: row r of the problem is row -x_bcast_rows * r of x
    const int xrow = x_bcast_rows > 0 ? rr % x_bcast_rows : (x_bcast_rows < 0 ? rr * -x_bcast_rows : rr);
    const float mu = mean[rr], rs = rstd[rr];
    float xh[NV][CH], gg[NV][CH];
    float c1 = 0.f, c2 = 0.f;
    // the residual-stream gradient is only needed after the row reduction: fetch it now, with x and dy, so one
    // memory round trip per row instead of two
    typename ChunkT<CH>::bt rraw[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int j = 0; j < CH; ++j) rraw[i][j] = (bf16)0.f;
      if (gres && ok) rraw[i] = *(const typename ChunkT<CH>::bt*)(gres + (size_t)rr * E + CH * (sub + LPR * i));
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = CH * (sub + LPR * i);
      float xv[CH], dv[CH], wm[CH], lbi[CH], dwv[CH];
      ld_chunk<CH>(x + (size_t)xrow * E + c, ok, xv);
      ld_chunk<CH>(dy + (size_t)rr * E + c, ok, dv);
      if (SLN) { ld_chunk<CH>(wmod + (size_t)rr * E + c, ok, wm); ld_f32<CH>(lbias + c, lbi); }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float h = ok ? (xv[j] - mu) * rs : 0.f;
        float d = dv[j];
        if (SLN) {
          const float l = h * gam[i][j] + lbi[j];
          dwv[j] = d * (g_s * l + b_s);
          s_gs += d * wm[j] * l;
          s_bs += d * wm[j];
          d *= wm[j] * g_s;
        }
        xh[i][j] = h;
        ag[i][j] += d * h;
        ab[i][j] += d;
        const float g = d * gam[i][j];
        gg[i][j] = g;
        c1 += g;
        c2 += g * h;
      }
      if (SLN && ok) {
        float* dwp = dw_acc + (size_t)rr * E + c;
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          f32x4 w0 = {dwv[4 * q], dwv[4 * q + 1], dwv[4 * q + 2], dwv[4 * q + 3]};
          if (dw_accumulate) w0 += *(const f32x4*)(dwp + 4 * q);
          *(f32x4*)(dwp + 4 * q) = w0;
        }
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
    c1 *= (1.0f / E); c2 *= (1.0f / E);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = CH * (sub + LPR * i);
      typename ChunkT<CH>::bt o;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        float t = rs * (gg[i][j] - c1 - xh[i][j] * c2);
        if (gres) t += vg_bf2f(rraw[i][j]);
        o[j] = vg_f2bf(t);
      }
      if (ok) *(typename ChunkT<CH>::bt*)(dx + (size_t)rr * E + c) = o;
      if (dxm) {  // gradient entering the dropped branch: dx * mask / keep  (same mask as the forward epilogue)
        const unsigned i4 = ((unsigned)(rr * drop_row_mul) * (unsigned)E + (unsigned)c) >> 2;  // (the row of the full tensor this compact row stands for)
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          const unsigned wd = vg_drop_word(dkey, i4 + q);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[4 * q + j] = vg_f2bf(vg_bf2f(o[4 * q + j]) * vg_drop_factor(wd, j, dthr, dscale));
        }
        if (ok) *(typename ChunkT<CH>::bt*)(dxm + (size_t)rr * E + c) = o;
      }
      if (ok) {
#pragma unroll
        for (int j = 0; j < CH; ++j) ac[i][j] += vg_bf2f(o[j]);
      }
    }
  }
  // fold: the RPW row groups of the wave (shuffles), then the 4 waves (LDS), fixed order
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float a = ag[i][j], b = ab[i][j], c = ac[i][j];
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
      if (rg == 0) {
        const int col = CH * (sub + LPR * i) + j;
        red[0][wv][col] = a; red[1][wv][col] = b; red[2][wv][col] = c;
      }
    }
  if (SLN) {
    s_gs = vg_wave_sum(s_gs); s_bs = vg_wave_sum(s_bs);
    if (lane == 0) { reds[wv][0] = s_gs; reds[wv][1] = s_bs; }
  }
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * part_w;
  for (int c = threadIdx.x; c < 3 * E; c += 256) {
    const int which = c / E, col = c - which * E;
    out[c] = (red[which][0][col] + red[which][1][col]) + (red[which][2][col] + red[which][3][col]);
  }
  if (SLN && threadIdx.x < 2) out[3 * E + threadIdx.x] = (reds[0][threadIdx.x] + reds[1][threadIdx.x]) + (reds[2][threadIdx.x] + reds[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// dst_k[c] (+)= sum_r part[r][off_k + c]  for up to 4 consecutive column segments.
struct VgSeg { float* dst; int n; };
#include "vg_fold.h"
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

// Many folds in ONE launch (blockIdx.y = job): the backward passes queue one job per LayerNorm and fold them all
// at the end instead of paying a ~10 us latency-bound launch per LayerNorm.
__global__ __launch_bounds__(256) void vg_colsum_f32_multi_kernel(VgFoldJobs jobs) {
  __shared__ float red[16][17];
  const VgFoldJob& J = jobs.j[blockIdx.y];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int width = J.width, rows = J.rows;
  if (blockIdx.x * 16 >= width) return;
  float a = 0.f;
  if (c < width) {
#pragma unroll 8
    for (int r = rl; r < rows; r += 16) a += J.part[(size_t)r * width + c];  // (same order of additions; the loads of eight rows in flight)
  }
  red[rl][cl] = a;
  __syncthreads();
  if (rl != 0 || c >= width) return;
  a = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) a += red[k][cl];
  int off = 0; float* dst = nullptr;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (c >= off && c < off + J.n[k]) dst = J.dst[k] ? J.dst[k] + (c - off) : nullptr;
    off += J.n[k];
  }
  if (dst) *dst += a;
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
#define NV_SWITCH(E_, CALL)                                                                                 \
  switch ((E_) >> 7) {                                                                                      \
    case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break;         \
    case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break;         \
    default: return -3;                                                                                     \
  }
int vg_ln_fwd_launch(const bf16* x, long long xs, const float* gamma, const float* beta, bf16* y, long long ys,
                     float* mean, float* rstd, int R, int E, float eps, hipStream_t st) {
  if ((E & 127) || E > 1024 || R < 1 || (xs & 7) || (ys & 7)) return -3;
#define LN_FWD(NV_) hipLaunchKernelGGL((vg_ln_fwd_kernel<false, NV_>), dim3((R + 15) / 16), dim3(256), 0, st, x, xs, 0, (const bf16*)nullptr, \
                                       gamma, beta, (const float*)nullptr, (const float*)nullptr, y, ys, mean, rstd, R, eps)
  NV_SWITCH(E, LN_FWD)
#undef LN_FWD
  return (int)hipGetLastError();
}
int vg_sln_fwd_launch(const bf16* h, int h_bcast_rows, const bf16* wmod, const float* lw, const float* lb,
                      const float* gs, const float* bs, bf16* y, float* mean, float* rstd, int R, int E, float eps,
                      hipStream_t st) {
  if ((E & 127) || E > 1024 || R < 1) return -3;
#define SLN_FWD(NV_) hipLaunchKernelGGL((vg_ln_fwd_kernel<true, NV_>), dim3((R + 15) / 16), dim3(256), 0, st, h, (long long)E, h_bcast_rows, wmod, \
                                        lw, lb, gs, bs, y, (long long)E, mean, rstd, R, eps)
  NV_SWITCH(E, SLN_FWD)
#undef SLN_FWD
  return (int)hipGetLastError();
}
// one partial row per workgroup: 16 rows per workgroup pass, at most LN_MAX_PARTS workgroups (bandwidth-bound: wants many)
int vg_ln_bwd_nparts(int R) { const int n = (R + 15) / 16; return n < LN_MAX_PARTS ? n : LN_MAX_PARTS; }
int vg_ln_bwd_launch(const bf16* dy, const bf16* x, const float* mean, const float* rstd, const float* gamma,
                     const bf16* gres, bf16* dx, float* part, int R, int E, bf16* dxm, unsigned dthr, unsigned dkey,
                     float dscale, const unsigned* dstep, hipStream_t st, int x_row_step, int drop_row_mul) {
  if ((E & 127) || E > 1024 || R < 1 || x_row_step < 1 || drop_row_mul < 1) return -3;
  const int xb = x_row_step > 1 ? -x_row_step : 0;  // row r of the problem reads row x_row_step * r of x (the CLS rows of [B, S, E])
#define LN_BWD(NV_) hipLaunchKernelGGL((vg_ln_bwd_kernel<false, NV_, LN_BWD_LPR>), dim3(vg_ln_bwd_nparts(R)), dim3(256), 0, st, dy, x, xb, mean, rstd, gamma, \
                     (const float*)nullptr, gres, dx, part, 3 * E, (const bf16*)nullptr, (const float*)nullptr,                                      \
                     (const float*)nullptr, (float*)nullptr, 0, R, dxm, dthr, dkey, dscale, dstep, drop_row_mul)
  NV_SWITCH(E, LN_BWD)
#undef LN_BWD
  return (int)hipGetLastError();
}
int vg_sln_bwd_launch(const bf16* dy, const bf16* h, int h_bcast_rows, const bf16* wmod, const float* mean,
                      const float* rstd, const float* lw, const float* lb, const float* gs, const float* bs,
                      const bf16* gres, bf16* dh, float* dw_acc, int dw_accumulate, float* part, int R, int E,
                      bf16* dhm, unsigned dthr, unsigned dkey, float dscale, const unsigned* dstep, hipStream_t st) {
  if ((E & 127) || E > 1024 || R < 1) return -3;
#define SLN_BWD(NV_) hipLaunchKernelGGL((vg_ln_bwd_kernel<true, NV_, SLN_BWD_LPR>), dim3(vg_ln_bwd_nparts(R)), dim3(256), 0, st, dy, h, h_bcast_rows, mean, \
                     rstd, lw, lb, gres, dh, part, 3 * E + 64, wmod, gs, bs, dw_acc, dw_accumulate, R, dhm, dthr, dkey, dscale, dstep, 1)
  NV_SWITCH(E, SLN_BWD)
#undef SLN_BWD
  return (int)hipGetLastError();
}
int vg_colsum_f32_launch(const float* part, int rows, int width, float* d0, int n0, float* d1, int n1, float* d2, int n2,
                         float* d3, int n3, int accumulate, hipStream_t st) {
  VgSegs s; s.s[0] = {d0, n0}; s.s[1] = {d1, n1}; s.s[2] = {d2, n2}; s.s[3] = {d3, n3};
  hipLaunchKernelGGL(vg_colsum_f32_kernel, dim3((width + 15) / 16), dim3(256), 0, st, part, rows, width, s, accumulate);
  return (int)hipGetLastError();
}
int vg_colsum_f32_multi_launch(const VgFoldJobs& jobs, hipStream_t st) {
  if (jobs.n < 1) return 0;
  if (jobs.n > VG_MAX_FOLD_JOBS) return -1;
  int wmax = 0;
  for (int i = 0; i < jobs.n; ++i) wmax = jobs.j[i].width > wmax ? jobs.j[i].width : wmax;
  hipLaunchKernelGGL(vg_colsum_f32_multi_kernel, dim3((wmax + 15) / 16, jobs.n), dim3(256), 0, st, jobs);
  return (int)hipGetLastError();
}
int vg_colsum_bf16_nparts(int R) { return (R + CS_ROWS - 1) / CS_ROWS; }
// first stage only: part[chunk][N] partial sums, to be folded later (vg_colsum_f32 / a queued VgFoldJob)
int vg_colsum_bf16_part_launch(const bf16* X, long long ld, int R, int N, float* part, hipStream_t st) {
  if ((N & 7) || (ld & 7) || R < 1) return -3;
  hipLaunchKernelGGL(vg_colsum_bf16_part_kernel, dim3((N + 255) / 256, vg_colsum_bf16_nparts(R)), dim3(256), 0, st, X, ld, R, N, part);
  return (int)hipGetLastError();
}
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
