// bf16 MFMA GEMM family for the ViTGAN hot path (gfx950).
//
// One kernel template, three operand forms (vg_gemm.h).  Tile 128(m) x 128(n) x 64(k),
// 256 threads = 4 waves as 2(m) x 2(n), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
// LDS: two stages x (16 KiB + 16 KiB), register-staged with the loads of step t+1 issued
// before the MFMAs of step t and written to the other stage afterwards (one barrier/step).
//
// The MFMA is issued "swapped": its A operand carries the GEMM's n index and its B operand the
// m index, so a lane's 4 accumulator registers are 4 CONSECUTIVE n of one output row m and the
// epilogue stores 8 B (bf16) / 16 B (fp32) per lane straight from registers.
//
// Operand images in LDS:
//   row form  [128 rows][64 k]  (128-B rows): 16-B chunk c of row r lives at chunk c ^ (r & 7);
//             fragments by ds_read_b128.
//   tr  form  [64 k][128 cols]  (256-B rows): 16-B chunk c of row k lives at c ^ (2*sigma(k)),
//             sigma(k) = (k&3) | ((k>>3)&1)<<2; fragments by two ds_read_b64_tr_b16, whose 32-lane
//             halves then touch all 64 banks exactly once.
#include "vg_gemm.h"

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES 32768
#define TILE_BYTES 16384

__device__ __forceinline__ int tr_sigma(int kk) { return (kk & 3) | (((kk >> 3) & 1) << 2); }

// ---- global -> registers ---------------------------------------------------------------
template <bool TR>
__device__ __forceinline__ void stage_load(u32x4 (&reg)[4], const bf16* __restrict__ X, int ld,
                                           int idx0, int idx_end, int k0, int k_end, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (!TR) {
      const int row = idx0 + (id >> 3), k = k0 + ((id & 7) << 3);
      if (row < idx_end && k < k_end) v = *(const u32x4*)(X + (size_t)row * ld + k);
    } else {
      const int k = k0 + (id >> 4), col = idx0 + ((id & 15) << 3);
      if (k < k_end && col < idx_end) v = *(const u32x4*)(X + (size_t)k * ld + col);
    }
    reg[i] = v;
  }
}

// ---- registers -> LDS --------------------------------------------------------------------
template <bool TR>
__device__ __forceinline__ void stage_write(const u32x4 (&reg)[4], unsigned char* tile, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    int off;
    if (!TR) {
      const int row = id >> 3, c = id & 7;
      off = row * 128 + ((c ^ (row & 7)) << 4);
    } else {
      const int kk = id >> 4, c = id & 15;
      off = kk * 256 + ((c ^ (2 * tr_sigma(kk))) << 4);
    }
    *(u32x4*)(tile + off) = reg[i];
  }
}

// ---- LDS -> MFMA fragment: 16 rows/cols starting at i0, k sub-step ks (32 wide) -------------
template <bool TR>
__device__ __forceinline__ bf16x8 load_frag(const unsigned char* tile, int i0, int ks, int lane) {
  const int g = lane >> 4, li = lane & 15;
  if (!TR) {
    const int row = i0 + li, c = 4 * ks + g;
    return *(const bf16x8*)(tile + row * 128 + ((c ^ (row & 7)) << 4));
  } else {
    const int q = li >> 2, p = li & 3;
    const int c8 = (i0 >> 2) + p;
    const int sw = 2 * (q | ((g & 1) << 2));
    const int kk0 = 32 * ks + 8 * g + q;
    const unsigned char* a0 = tile + kk0 * 256 + ((((c8 >> 1) ^ sw)) << 4) + ((c8 & 1) << 3);
    typedef bf16x4 __attribute__((address_space(3))) * lds4;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(a0));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(a0 + 4 * 256));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

__device__ __forceinline__ float apply_act(int act, float scale, float v) {
  switch (act) {
    case VG_ACT_GELU: return vg_gelu(v);
    case VG_ACT_SIN: return __sinf(scale * v);
    case VG_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void vg_gemm_kernel(const VgGemmGroup grp) {
  constexpr bool A_TR = (MODE == VG_TN);
  constexpr bool B_TR = (MODE != VG_NT);
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];

  // XCD-aware block order: blocks b, b+8, ... share an XCD (L2); give each XCD a contiguous
  // run of tiles so the n-tiles of one m-panel hit the same L2 (bijective for any grid size).
  int bid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
    const int q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < VG_MAX_GROUP; ++i)
    if (i < grp.n && bid >= grp.p[i].tile_start) pi = i;
  const VgGemmProb& P = grp.p[pi];

  const int local = bid - P.tile_start;
  const int tiles_mn = P.tiles_m * P.tiles_n;
  const int split = local / tiles_mn;
  const int t = local - split * tiles_mn;
  const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int k_begin = split * P.k_per_split;
  const int k_end = min(P.K, k_begin + P.k_per_split);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int g = lane >> 4, li = lane & 15;

  f32x4 acc[4][4];  // [nt][mt]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bf16* __restrict__ Ag = P.A;
  const bf16* __restrict__ Bg = P.B;
  const int lda = P.lda, ldb = P.ldb;
  // row form indexes rows (m or n) against M/N; tr form indexes columns against M/N.
  u32x4 ra[4], rb[4];
  const int nsteps = (k_end - k_begin + BK - 1) / BK;
  if (nsteps > 0) {
    stage_load<A_TR>(ra, Ag, lda, m0, P.M, k_begin, k_end, tid);
    stage_load<B_TR>(rb, Bg, ldb, n0, P.N, k_begin, k_end, tid);
    stage_write<A_TR>(ra, smem, tid);
    stage_write<B_TR>(rb, smem + TILE_BYTES, tid);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    unsigned char* cur = smem + (s & 1) * STAGE_BYTES;
    unsigned char* nxt = smem + ((s + 1) & 1) * STAGE_BYTES;
    const bool more = (s + 1 < nsteps);
    if (more) {
      const int k0 = k_begin + (s + 1) * BK;
      stage_load<A_TR>(ra, Ag, lda, m0, P.M, k0, k_end, tid);
      stage_load<B_TR>(rb, Bg, ldb, n0, P.N, k0, k_end, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fm[4], fn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fm[i] = load_frag<A_TR>(cur, wm * 64 + i * 16, ks, lane);
        fn[i] = load_frag<B_TR>(cur + TILE_BYTES, wn * 64 + i * 16, ks, lane);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = vg_mfma(fn[nt], fm[mt], acc[nt][mt]);
    }
    if (more) {
      stage_write<A_TR>(ra, nxt, tid);
      stage_write<B_TR>(rb, nxt + TILE_BYTES, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns row m (per mt) and 4 consecutive n (per nt) ----------------------
  const int act = P.act;
  const float ascale = P.act_scale;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + li;
    if (m >= P.M) continue;
    int mo = m;
    if (P.row_in_per > 0) mo = (m / P.row_in_per) * P.row_out_per + P.row_out_off + (m % P.row_in_per);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + 4 * g;
      if (n >= P.N) continue;
      f32x4 v = acc[nt][mt];
      if (MODE == VG_TN) {
        *(f32x4*)(P.Cf + (size_t)split * P.cf_split_stride + (size_t)m * P.ldcf + n) = v;
        continue;
      }
      if (P.bias) {
        const f32x4 b = *(const f32x4*)(P.bias + n);
        v += b;
      }
      if (P.pre_f32) *(f32x4*)(P.Cf + (size_t)mo * P.ldcf + n) = v;
      if (P.C2) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = vg_f2bf(v[r]);
        *(bf16x4*)(P.C2 + (size_t)mo * P.ldc2 + n) = o;
      }
      if (act == VG_ACT_MUL_GELU_GRAD) {
        const bf16x4 z = *(const bf16x4*)(P.Z + (size_t)m * P.ldz + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= vg_gelu_grad(vg_bf2f(z[r]));
      } else if (act == VG_ACT_MUL_TANH_GRAD) {
        const bf16x4 z = *(const bf16x4*)(P.Z + (size_t)m * P.ldz + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float tv = vg_bf2f(z[r]); v[r] *= 1.f - tv * tv; }
      } else if (act == VG_ACT_MUL_COS) {
        const f32x4 z = *(const f32x4*)(P.Zf + (size_t)m * P.ldzf + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= ascale * __cosf(ascale * z[r]);
      } else if (act != VG_ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(act, ascale, v[r]);
      }
      if (P.res) {
        const bf16x4 rr = *(const bf16x4*)(P.res + (size_t)mo * P.ldr + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += vg_bf2f(rr[r]);
      }
      if (P.resf) {
        const f32x4 rr = *(const f32x4*)(P.resf + (size_t)(m % P.res_period) * P.N + n);
        v += rr;
      }
      if (P.C) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = vg_f2bf(v[r]);
        *(bf16x4*)(P.C + (size_t)mo * P.ldc + n) = o;
      }
    }
  }
}

int vg_gemm_launch(VgGemmProb* probs, int n, int mode, hipStream_t stream) {
  if (n < 1 || n > VG_MAX_GROUP) return -1;
  VgGemmGroup grp;
  grp.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    VgGemmProb& p = probs[i];
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -2;
    // 16-byte vector loads: contiguous extents and leading dims must be multiples of 8 elements
    if ((p.lda & 7) || (p.ldb & 7) || (p.N & 7)) return -3;
    if (mode == VG_NT && (p.K & 7)) return -3;
    if (mode == VG_NN && (p.K & 7)) return -3;
    if (mode == VG_TN && (p.M & 7)) return -3;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    int splits = (mode == VG_TN) ? (p.splits > 0 ? p.splits : 1) : 1;
    int ksteps = (p.K + BK - 1) / BK;
    int per = (ksteps + splits - 1) / splits;
    p.k_per_split = per * BK;
    splits = (ksteps + per - 1) / per;  // drop empty slices
    p.splits = splits;
    p.tile_start = total;
    total += p.tiles_m * p.tiles_n * splits;
    grp.p[i] = p;
  }
  dim3 grid(total), block(256);
  switch (mode) {
    case VG_NT: hipLaunchKernelGGL(vg_gemm_kernel<VG_NT>, grid, block, 0, stream, grp); break;
    case VG_NN: hipLaunchKernelGGL(vg_gemm_kernel<VG_NN>, grid, block, 0, stream, grp); break;
    case VG_TN: hipLaunchKernelGGL(vg_gemm_kernel<VG_TN>, grid, block, 0, stream, grp); break;
    default: return -4;
  }
  return (int)hipGetLastError();
}
