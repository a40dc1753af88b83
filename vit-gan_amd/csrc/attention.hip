// Fused multi-head self-attention (forward and backward) for short sequences (S <= 80): one workgroup
// per (image, head), ONE WAVE PER 16-ROW TILE (5 waves for S = 65).  gfx950 only.
// These kernels are HBM/latency bound (4 % of the step's FLOPs): every operand of a head is staged into LDS
// exactly once by LDS-DMA in a single load phase, everything is computed from LDS, and the outputs leave as
// 16-byte stores; the co-resident workgroups of a CU overlap each other's load / compute / store phases.
//
// The whole S x S score tile lives in MFMA accumulators; softmax runs in registers.
// Layout trick (cdna_hip_programming.md s3, "an accumulator tile as the next MFMA's operand"):
// scores are produced as S^T = K Q^T, so a lane owns ONE query column and 4 consecutive keys per
// tile.  The exponentiated registers are then, unchanged, the k-operand of the P.V product, whose
// other operand (V, keys along k) is read from LDS with ds_read_b64_tr_b16 in the matching key
// order.  No probability ever touches LDS or HBM.
//
// qkv layout: [B*S, 3E] bf16 row-major, columns [0,E) = Q, [E,2E) = K, [2E,3E) = V, head h at
// columns h*HE .. (h+1)*HE of each third.  o / d_o: [B*S, E].  lse: [B, H, S] fp32 (natural log).
#include "vg_common.h"

// LDS image of one head: row-major [rows][HE] bf16 whose 16-B chunks are XOR-swizzled by the row so that BOTH
// access patterns are bank-conflict free:
//   row form   (ds_read_b128, 16 lanes = 16 consecutive rows, same chunk)   and
//   transposed (ds_read_b64_tr_b16, 32 lanes = 8 consecutive rows x one 32-B chunk pair).
// HE = 96 / 32 (row pitch 48 / 16 banks: rows r and r+4 share a bank quadrant): position inside each 64-B window
//   is XORed with F[(r>>2)&3], F = {0,2,1,3} - rows r+4 move to the other pair, rows r+8 / r+12 swap halves.
// HE = 64 (pitch 32 banks: rows r and r+2 collide): pair index ^ (r>>1)&3, half ^ (r>>3)&1.
// The map is an involution on the chunk index, so the DMA applies the same function to its SOURCE chunk.
template <int HE>
__device__ __forceinline__ int swz_chunk(int r, int c) {
  if (HE == 64) return (((c >> 1) ^ ((r >> 1) & 3)) << 1) | ((c & 1) ^ ((r >> 3) & 1));
  const int x = (r >> 2) & 3;
  return (c & ~3) | ((c & 3) ^ (((x & 1) << 1) | (x >> 1)));
}
template <int HE>
__device__ __forceinline__ int lds_off(int r, int d) {
  return r * (HE * 2) + (swz_chunk<HE>(r, d >> 3) << 4) + ((d & 7) << 1);
}

// Stage rows [0, rows_alloc) x HE of one head into an LDS image by LDS-DMA (global_load_lds_dwordx4: no trip
// through registers, every request a whole 16-B chunk of a 64..192-B row segment).  One instruction fills
// 1 KiB lane-linearly, so LDS chunk (row r, position c') = linear chunk 64*piece + lane and the XOR swizzle
// (swz_chunk) is applied to the SOURCE chunk index.  Rows >= S come from a 16-byte zero page (the padded keys'
// V rows multiply p = 0 and must be finite).  rows_alloc * HE / 8 must be a multiple of 64.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int HE, int NW>
__device__ __forceinline__ void dma_head(unsigned char* img, const bf16* __restrict__ src, size_t ld, int S, int rows_alloc,
                                         const void* zeros, int wave, int lane) {
  constexpr int CPR = HE / 8;  // 16-B chunks per row
  const int pieces = rows_alloc * CPR / 64;
  for (int pc = wave; pc < pieces; pc += NW) {
    const int ci = 64 * pc + lane;
    const int r = ci / CPR, cp = ci - r * CPR;
    const int c = swz_chunk<HE>(r, cp);
    const void* p = (r < S) ? (const void*)(src + (size_t)r * ld + 8 * c) : zeros;
    __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(img + 1024 * pc), 16, 0, 0);
  }
}

// Accumulator tiles [dt] (lane = row li, 4 consecutive head-dim columns 16*dt + 4*g ..) -> bf16 row segments.
// v_permlane16_swap between the even and the odd tile of a pair hands every lane 8 CONSECUTIVE columns, so a lane
// stores 16 B and a wave-instruction covers 16 rows x 64 B (same exchange as the GEMM epilogue).
template <int DT>
__device__ __forceinline__ void store_tiles(bf16* __restrict__ rowp, const f32x4 (&acc)[DT], float mul, int g, bool ok) {
#pragma unroll
  for (int pr = 0; pr < DT / 2; ++pr) {
    const f32x4 te = acc[2 * pr], to = acc[2 * pr + 1];
    bf16x8 w;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(te[r] * mul), __float_as_uint(to[r] * mul), false, false);
      w[r] = vg_f2bf(__uint_as_float(sw[0]));
      w[r + 4] = vg_f2bf(__uint_as_float(sw[1]));
    }
    if (ok) *(bf16x8*)(rowp + 32 * pr + ((g & 1) << 4) + ((g & 2) << 2)) = w;
  }
}

// row-form fragment straight from global: rows r0+li, head-dim slice 32*ks + 8*g
__device__ __forceinline__ bf16x8 gfrag(const bf16* __restrict__ src, size_t ld, int r0, int ks, int S, int lane) {
  const int row = r0 + (lane & 15), d = 32 * ks + 8 * (lane >> 4);
  bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  if (row < S) z = *(const bf16x8*)(src + (size_t)row * ld + d);
  return z;
}
template <int HE>
__device__ __forceinline__ bf16x8 lfrag_row(const unsigned char* lds, int r0, int ks, int lane) {
  return *(const bf16x8*)(lds + lds_off<HE>(r0 + (lane & 15), 32 * ks + 8 * (lane >> 4)));
}
// transposed fragment: non-k index = head-dim columns d0..d0+15 (on the lane), k = rows
// (keys or queries) in the accumulator order {32u + 4g + j (j<4), 32u + 16 + 4g + (j-4)}.
template <int HE>
__device__ __forceinline__ bf16x8 lfrag_tr(const unsigned char* lds, int u, int d0, int lane) {
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  typedef bf16x4 __attribute__((address_space(3))) * lds4;
  const int r = 32 * u + 4 * g + q;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(lds + lds_off<HE>(r, d0 + 4 * p)));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(lds + lds_off<HE>(r + 16, d0 + 4 * p)));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}
// ---- fp8 (OCP e4m3) operands for the activation-side products (config C5: "fp8 MFMA attention") ------------------------
// An fp8 16x16x32 fragment is 8 bytes per lane, element j of lane group g pairing with element j of the other operand's
// lane group g exactly like the bf16 fragment's 8 elements - so a bf16 fragment converts element by element.
typedef long fp8x8;
__device__ __forceinline__ fp8x8 to_fp8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(a4, a5, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(a6, a7, hi, true);
  return (fp8x8)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}
__device__ __forceinline__ fp8x8 to_fp8(bf16x8 v) {
  return to_fp8(vg_bf2f(v[0]), vg_bf2f(v[1]), vg_bf2f(v[2]), vg_bf2f(v[3]), vg_bf2f(v[4]), vg_bf2f(v[5]), vg_bf2f(v[6]), vg_bf2f(v[7]));
}
__device__ __forceinline__ fp8x8 pack_pair_fp8(f32x4 a, f32x4 b, float mul) {  // same element order as pack_pair
  return to_fp8(a[0] * mul, a[1] * mul, a[2] * mul, a[3] * mul, b[0] * mul, b[1] * mul, b[2] * mul, b[3] * mul);
}
__device__ __forceinline__ f32x4 vg_mfma_fp8(fp8x8 a, fp8x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0); }
// score product with either operand type (FP8: both operands rounded to e4m3; accumulation stays fp32)
template <bool FP8>
__device__ __forceinline__ f32x4 score_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (FP8) return vg_mfma_fp8(to_fp8(a), to_fp8(b), c);
  else return vg_mfma(a, b, c);
}
#define VG_P_FP8_SCALE 256.0f  // softmax numerators in (0, 1] are scaled into e4m3's normal range before the P.V product

__device__ __forceinline__ bf16x8 pack_pair(f32x4 a, f32x4 b) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) { o[j] = vg_f2bf(a[j]); o[j + 4] = vg_f2bf(b[j]); }
  return o;
}
__device__ __forceinline__ float group_sum(float v) {  // over the 4 lane groups (lane>>4)
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// squared L2 norm of every row of an LDS head image -> out[rows] (rows >= S are zero rows).  One thread per row.
template <int HE>
__device__ __forceinline__ void row_sqnorms(const unsigned char* img, float* out, int rows, int tid, int nthreads) {
  for (int r = tid; r < rows; r += nthreads) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < HE / 8; ++c) {
      const bf16x8 v = *(const bf16x8*)(img + lds_off<HE>(r, 8 * c));
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = vg_bf2f(v[j]); a += f * f; }
    }
    out[r] = a;
  }
}
// v1 attention score (src/v1/attention.py:66-67): the Euclidean distance |q - k| from q.k and the squared norms
__device__ __forceinline__ float l2_dist(float qk, float qn, float kn) { return sqrtf(fmaxf(qn + kn - 2.f * qk, 0.f)); }

// workgroup -> (image, head).  The heads of one image read interleaved 2 HE-byte slices of the same rows of qkv / o / d_o
// (HE = 96: 192-byte segments, 1.5 cache lines - neighbouring heads share a line), and consecutive workgroup ids go round-robin
// over the 8 XCDs, each with its own L2: the heads of an image therefore sit on ONE XCD, as consecutive workgroups of it
// (id = 8 i + x: image 8 (i / H) + x, head i % H), so a shared line is fetched from HBM once.
__device__ __forceinline__ bool attn_block(int B, int H, int& b, int& h) {
#ifdef VG_ATTN_LINEAR_MAP  // A/B builds: the plain mapping
  b = blockIdx.x / H; h = blockIdx.x - b * H;
#else
  const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
  b = (i / H) * 8 + x; h = i % H;
#endif
  return b < B;
}
static inline int attn_grid(int B, int H) { return ((B + 7) / 8) * 8 * H; }

template <int HE, int NT, bool L2, bool FP8>
__global__ __launch_bounds__(64 * NT) void vg_attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                              float* __restrict__ lse, int B, int S, int H, float scale,
                                                              const void* __restrict__ zeros) {
  constexpr int KS = HE / 32, DT = HE / 16, KP = (NT + 1) / 2, RP = KP * 32, RK = 16 * NT;
  __shared__ __attribute__((aligned(16))) unsigned char sm[(RK + RP) * HE * 2 + (L2 ? RK * 4 : 0)];
  unsigned char* kl = sm;                 // K, rows [0, 16 NT): row-form fragments
  unsigned char* vl = sm + RK * HE * 2;   // V, rows [0, 32 KP): transposed fragments
  float* kn = (float*)(sm + (RK + RP) * HE * 2);  // L2 scores: |k|^2 per key
  int b, h;
  if (!attn_block(B, H, b, h)) return;
  const int tid = threadIdx.x, lane = tid & 63, qt = tid >> 6;  // wave qt owns query rows 16*qt .. 16*qt+15
  const int g = lane >> 4, li = lane & 15;
  const int E = H * HE;
  const size_t ld = 3 * (size_t)E;
  const bf16* qb = qkv + (size_t)b * S * ld + h * HE;
  const bf16* kb = qb + E;
  const bf16* vb = qb + 2 * E;

  bf16x8 qf[KS];  // this wave's queries straight from global (nobody else needs them)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = gfrag(qb, ld, 16 * qt, ks, S, lane);
  dma_head<HE, NT>(kl, kb, ld, S, RK, zeros, qt, lane);
  dma_head<HE, NT>(vl, vb, ld, S, RP, zeros, (qt + 2) % NT, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float qn = 0.f;
  if (L2) {
    row_sqnorms<HE>(kl, kn, RK, tid, 64 * NT);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = vg_bf2f(qf[ks][j]); qn += f * f; }
    qn = group_sum(qn);
    __syncthreads();
  }

  f32x4 sc[NT];  // [kt]: rows = keys 16kt+4g+r, col = query 16qt+li
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a = score_mfma<FP8>(lfrag_row<HE>(kl, 16 * kt, ks, lane), qf[ks], a);
    sc[kt] = a;
  }
  const int q = 16 * qt + li;
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * kt + 4 * g + r;
      const float raw = L2 ? l2_dist(sc[kt][r], qn, kn[key]) : sc[kt][r];
      const float sv = (key < S) ? raw * scale : -INFINITY;
      sc[kt][r] = sv;
      m = fmaxf(m, sv);
    }
  m = group_max(m);
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __expf(sc[kt][r] - m);
      sc[kt][r] = p;
      l += p;
    }
  l = group_sum(l);
  const float inv_l = 1.0f / l;
  if (g == 0 && q < S) lse[((size_t)b * H + h) * S + q] = m + __logf(l);

  f32x4 oa[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) oa[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < KP; ++u) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 hi = (2 * u + 1 < NT) ? sc[(2 * u + 1 < NT) ? 2 * u + 1 : 0] : zero;
    if constexpr (FP8) {
      const fp8x8 pf = pack_pair_fp8(sc[2 * u], hi, VG_P_FP8_SCALE);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) oa[dt] = vg_mfma_fp8(to_fp8(lfrag_tr<HE>(vl, u, 16 * dt, lane)), pf, oa[dt]);
    } else {
      const bf16x8 pf = pack_pair(sc[2 * u], hi);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) oa[dt] = vg_mfma(lfrag_tr<HE>(vl, u, 16 * dt, lane), pf, oa[dt]);
    }
  }
  store_tiles<DT>(o + ((size_t)b * S + (q < S ? q : 0)) * E + h * HE, oa, FP8 ? inv_l * (1.0f / VG_P_FP8_SCALE) : inv_l, g, q < S);
}

// Backward.  Phase A works in the S^T orientation (lane = query) and yields dQ; phase B in the
// S orientation (lane = key) and yields dK, dV.  Recomputing the 65x65 tile in both orientations
// costs 2 x 75 extra MFMAs per head and removes every register transpose.
// FP8: the score product is recomputed with the forward's e4m3 operands (so P matches the forward's lse exactly); every
// product that carries a gradient operand (dP, dV, dQ, dK) stays bf16 - gradients need the range.
template <int HE, int NT, bool L2, bool FP8>
__global__ __launch_bounds__(64 * NT, 2) void vg_attn_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                         const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                         bf16* __restrict__ dqkv, int B, int S, int H, float scale,
                                                         const void* __restrict__ zeros) {
  constexpr int KS = HE / 32, DT = HE / 16, KP = (NT + 1) / 2, RP = KP * 32;
  constexpr int IMG = RP * HE * 2;
  // Everything a head needs is staged ONCE: K, V, Q, dO images (LDS-DMA) + lse and delta per query.  One load
  // phase, one compute phase, one store phase per workgroup; the co-resident workgroup overlaps them.
  __shared__ __attribute__((aligned(16))) unsigned char sm[4 * IMG + (L2 ? 4 : 2) * RP * 4];
  unsigned char* lk = sm;
  unsigned char* lv = sm + IMG;
  unsigned char* lq = sm + 2 * IMG;
  unsigned char* ldo = sm + 3 * IMG;
  float* dl = (float*)(sm + 4 * IMG);  // delta[q] = sum_d dO*O
  float* ll = dl + RP;                 // lse[q]
  float* qn_l = ll + RP;               // L2 scores: |q|^2 per query, |k|^2 per key
  float* kn_l = qn_l + RP;
  int b, h;
  if (!attn_block(B, H, b, h)) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;  // wave wv owns query tile wv (phase A) / key tile wv (phase B)
  const int g = lane >> 4, li = lane & 15;
  const int E = H * HE;
  const size_t ld = 3 * (size_t)E;
  const bf16* qb = qkv + (size_t)b * S * ld + h * HE;
  const bf16* kb = qb + E;
  const bf16* vb = qb + 2 * E;
  const bf16* ob = o + (size_t)b * S * E + h * HE;
  const bf16* dob = d_o + (size_t)b * S * E + h * HE;
  const float* lb = lse + ((size_t)b * H + h) * S;
  bf16* dqb = dqkv + (size_t)b * S * ld + h * HE;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  bf16x8 of[KS];  // O is only needed for delta: this wave's 16 query rows, straight from global
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) of[ks] = gfrag(ob, (size_t)E, 16 * wv, ks, S, lane);
  dma_head<HE, NT>(lk, kb, ld, S, RP, zeros, wv, lane);
  dma_head<HE, NT>(lv, vb, ld, S, RP, zeros, (wv + 1) % NT, lane);
  dma_head<HE, NT>(lq, qb, ld, S, RP, zeros, (wv + 2) % NT, lane);
  dma_head<HE, NT>(ldo, dob, (size_t)E, S, RP, zeros, (wv + 3) % NT, lane);
  for (int i = tid; i < RP; i += 64 * NT) ll[i] = (i < S) ? lb[i] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  bf16x8 qf[KS], dof[KS];
  {
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = lfrag_row<HE>(lq, 16 * wv, ks, lane);
      dof[ks] = lfrag_row<HE>(ldo, 16 * wv, ks, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) dpart += vg_bf2f(dof[ks][j]) * vg_bf2f(of[ks][j]);
    }
    const float delta = group_sum(dpart);
    if (g == 0) dl[16 * wv + li] = delta;
  }
  if (L2) { row_sqnorms<HE>(lq, qn_l, RP, tid, 64 * NT); row_sqnorms<HE>(lk, kn_l, RP, tid, 64 * NT); }
  __syncthreads();

  // ---------------- phase A: S^T orientation (lane = query) -> dQ ----------------------
  {
    const int qt = wv;
    const int q = 16 * qt + li;
    const float delta = dl[q];
    const float lse_q = ll[q];
    const float qn = L2 ? qn_l[q] : 0.f;
    float wsum = 0.f;  // L2: sum_k W[q,k], W = dL/d dist / dist
    f32x4 ds[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      f32x4 st = zero, dpt = zero;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        st = score_mfma<FP8>(lfrag_row<HE>(lk, 16 * kt, ks, lane), qf[ks], st);
        dpt = vg_mfma(lfrag_row<HE>(lv, 16 * kt, ks, lane), dof[ks], dpt);
      }
      f32x4 kn4 = zero;
      if (L2) kn4 = *(const f32x4*)(kn_l + 16 * kt + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * g + r;
        const float raw = L2 ? l2_dist(st[r], qn, kn4[r]) : st[r];
        const float p = (key < S && q < S) ? __expf(raw * scale - lse_q) : 0.f;
        float dsv = p * (dpt[r] - delta) * scale;
        if (L2) { dsv = raw > 0.f ? dsv / raw : 0.f; wsum += dsv; }  // d dist/dq = (q - k)/dist
        ds[kt][r] = dsv;
      }
    }
    if (L2) wsum = group_sum(wsum);
    f32x4 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = zero;
#pragma unroll
    for (int u = 0; u < KP; ++u) {
      const bf16x8 dsf = pack_pair(ds[2 * u], (2 * u + 1 < NT) ? ds[(2 * u + 1 < NT) ? 2 * u + 1 : 0] : zero);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dq[dt] = vg_mfma(lfrag_tr<HE>(lk, u, 16 * dt, lane), dsf, dq[dt]);
    }
    if (L2) {  // dQ = rowsum(W) q - W K
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16x4 qv = *(const bf16x4*)(lq + lds_off<HE>(q, 16 * dt + 4 * g));
#pragma unroll
        for (int r = 0; r < 4; ++r) dq[dt][r] = wsum * vg_bf2f(qv[r]) - dq[dt][r];
      }
    }
    store_tiles<DT>(dqb + (size_t)(q < S ? q : 0) * ld, dq, 1.0f, g, q < S);
  }
  // ---------------- phase B: S orientation (lane = key) -> dK, dV ---------------------
  {
    const int kt = wv;
    const int key = 16 * kt + li;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = lfrag_row<HE>(lk, 16 * kt, ks, lane);
      vf[ks] = lfrag_row<HE>(lv, 16 * kt, ks, lane);
    }
    f32x4 pr[NT], ds[NT];
    const float kn = L2 ? kn_l[key] : 0.f;
    float wsum = 0.f;  // L2: sum_q W[q,key]
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      f32x4 s = zero, dp = zero;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s = score_mfma<FP8>(lfrag_row<HE>(lq, 16 * qt, ks, lane), kf[ks], s);
        dp = vg_mfma(lfrag_row<HE>(ldo, 16 * qt, ks, lane), vf[ks], dp);
      }
      const f32x4 lq4 = *(const f32x4*)(ll + 16 * qt + 4 * g);
      const f32x4 dl4 = *(const f32x4*)(dl + 16 * qt + 4 * g);
      f32x4 qn4 = zero;
      if (L2) qn4 = *(const f32x4*)(qn_l + 16 * qt + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 16 * qt + 4 * g + r;
        const bool ok = (q < S) && (key < S);
        const float raw = L2 ? l2_dist(s[r], qn4[r], kn) : s[r];
        const float p = ok ? __expf(raw * scale - lq4[r]) : 0.f;
        pr[qt][r] = p;
        float dsv = p * (dp[r] - dl4[r]) * scale;
        if (L2) { dsv = raw > 0.f ? dsv / raw : 0.f; wsum += dsv; }  // d dist/dk = (k - q)/dist
        ds[qt][r] = dsv;
      }
    }
    if (L2) wsum = group_sum(wsum);
    f32x4 dv[DT], dk[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dv[dt] = zero; dk[dt] = zero; }
#pragma unroll
    for (int u = 0; u < KP; ++u) {
      const int hi = (2 * u + 1 < NT) ? 2 * u + 1 : 0;
      const bf16x8 pf = pack_pair(pr[2 * u], (2 * u + 1 < NT) ? pr[hi] : zero);
      const bf16x8 dsf = pack_pair(ds[2 * u], (2 * u + 1 < NT) ? ds[hi] : zero);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dv[dt] = vg_mfma(lfrag_tr<HE>(ldo, u, 16 * dt, lane), pf, dv[dt]);
        dk[dt] = vg_mfma(lfrag_tr<HE>(lq, u, 16 * dt, lane), dsf, dk[dt]);
      }
    }
    if (L2) {  // dK = colsum(W) k - W^T Q
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16x4 kv = *(const bf16x4*)(lk + lds_off<HE>(key, 16 * dt + 4 * g));
#pragma unroll
        for (int r = 0; r < 4; ++r) dk[dt][r] = wsum * vg_bf2f(kv[r]) - dk[dt][r];
      }
    }
    bf16* rowp = dqb + (size_t)(key < S ? key : 0) * ld;
    store_tiles<DT>(rowp + E, dk, 1.0f, g, key < S);
    store_tiles<DT>(rowp + 2 * E, dv, 1.0f, g, key < S);
  }
}

// The same backward with TWO LDS images instead of four (dot-product scores, S > 32): phase A needs K and V whole and only this
// wave's 16 rows of Q and dO (read straight from global as fragments), phase B needs Q and dO whole and only this wave's rows of K
// and V (fragments taken from the images before they are overwritten) - so Q and dO are staged into the SAME two images between
// the phases.  37 KB instead of 74 KB of LDS per workgroup: four workgroups (20 waves) per CU instead of two, and it is the
// co-resident workgroups that overlap one another's load / compute / store phases (the kernel is latency-bound: 4.0 TB/s of its
// 205 MB at two workgroups per CU).  HBM bytes are unchanged (the fragment reads of the own rows hit L2 or are the first touch of
// lines the second staging then finds there).  Arithmetic and operand values are those of vg_attn_bwd_kernel: results are bit-equal.
#ifndef VG_ATTN_BWD2_WPE
#define VG_ATTN_BWD2_WPE 5  // waves per SIMD the register budget is set for: 5 = four 5-wave workgroups per CU (96 VGPRs)
#endif
template <int HE, int NT, bool FP8>
__global__ __launch_bounds__(64 * NT, VG_ATTN_BWD2_WPE) void vg_attn_bwd2_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                                  const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                                  bf16* __restrict__ dqkv, int B, int S, int H, float scale,
                                                                  const void* __restrict__ zeros) {
  constexpr int KS = HE / 32, DT = HE / 16, KP = (NT + 1) / 2, RP = KP * 32;
  constexpr int IMG = RP * HE * 2;
  __shared__ __attribute__((aligned(16))) unsigned char sm[2 * IMG + 2 * RP * 4];
  unsigned char* s0 = sm;        // K, then Q
  unsigned char* s1 = sm + IMG;  // V, then dO
  float* dl = (float*)(sm + 2 * IMG);  // delta[q] = sum_d dO*O
  float* ll = dl + RP;                 // lse[q]
  int b, h;
  if (!attn_block(B, H, b, h)) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;  // wave wv owns query tile wv (phase A) / key tile wv (phase B)
  const int g = lane >> 4, li = lane & 15;
  const int E = H * HE;
  const size_t ld = 3 * (size_t)E;
  const bf16* qb = qkv + (size_t)b * S * ld + h * HE;
  const bf16* kb = qb + E;
  const bf16* vb = qb + 2 * E;
  const bf16* ob = o + (size_t)b * S * E + h * HE;
  const bf16* dob = d_o + (size_t)b * S * E + h * HE;
  const float* lb = lse + ((size_t)b * H + h) * S;
  bf16* dqb = dqkv + (size_t)b * S * ld + h * HE;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  dma_head<HE, NT>(s0, kb, ld, S, RP, zeros, wv, lane);
  dma_head<HE, NT>(s1, vb, ld, S, RP, zeros, (wv + 1) % NT, lane);
  bf16x8 qf[KS], dof[KS];
  {
    bf16x8 of[KS];  // this wave's 16 query rows of O, Q, dO straight from global
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      of[ks] = gfrag(ob, (size_t)E, 16 * wv, ks, S, lane);
      dof[ks] = gfrag(dob, (size_t)E, 16 * wv, ks, S, lane);
      qf[ks] = gfrag(qb, ld, 16 * wv, ks, S, lane);
    }
    for (int i = tid; i < RP; i += 64 * NT) ll[i] = (i < S) ? lb[i] : 0.f;
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dpart += vg_bf2f(dof[ks][j]) * vg_bf2f(of[ks][j]);
    }
    const float delta = group_sum(dpart);
    if (g == 0) dl[16 * wv + li] = delta;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------- phase A: S^T orientation (lane = query) -> dQ ----------------------
  {
    const int q = 16 * wv + li;
    const float delta = dl[q];
    const float lse_q = ll[q];
    f32x4 ds[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      f32x4 st = zero, dpt = zero;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        st = score_mfma<FP8>(lfrag_row<HE>(s0, 16 * kt, ks, lane), qf[ks], st);
        dpt = vg_mfma(lfrag_row<HE>(s1, 16 * kt, ks, lane), dof[ks], dpt);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * g + r;
        const float p = (key < S && q < S) ? __expf(st[r] * scale - lse_q) : 0.f;
        ds[kt][r] = p * (dpt[r] - delta) * scale;
      }
    }
    bf16x8 dsf[KP];
#pragma unroll
    for (int u = 0; u < KP; ++u) dsf[u] = pack_pair(ds[2 * u], (2 * u + 1 < NT) ? ds[(2 * u + 1 < NT) ? 2 * u + 1 : 0] : zero);
    bf16* rowq = dqb + (size_t)(q < S ? q : 0) * ld;
#pragma unroll
    for (int pp = 0; pp < DT / 2; ++pp) {  // two head-dim tiles (32 columns = one 16-byte store per lane) at a time: registers
      f32x4 dq[2] = {zero, zero};
#pragma unroll
      for (int u = 0; u < KP; ++u) {
        dq[0] = vg_mfma(lfrag_tr<HE>(s0, u, 32 * pp, lane), dsf[u], dq[0]);
        dq[1] = vg_mfma(lfrag_tr<HE>(s0, u, 32 * pp + 16, lane), dsf[u], dq[1]);
      }
      store_tiles<2>(rowq + 32 * pp, dq, 1.0f, g, q < S);
    }
  }
  // this wave's key tile for phase B, before Q and dO take the images over
  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = lfrag_row<HE>(s0, 16 * wv, ks, lane);
    vf[ks] = lfrag_row<HE>(s1, 16 * wv, ks, lane);
  }
  __syncthreads();  // (waits for the LDS reads above: every wave is done with K and V)
  dma_head<HE, NT>(s0, qb, ld, S, RP, zeros, wv, lane);
  dma_head<HE, NT>(s1, dob, (size_t)E, S, RP, zeros, (wv + 1) % NT, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---------------- phase B: S orientation (lane = key) -> dK, dV ---------------------
  {
    const int key = 16 * wv + li;
    f32x4 pr[NT], ds[NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      f32x4 s = zero, dp = zero;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s = score_mfma<FP8>(lfrag_row<HE>(s0, 16 * qt, ks, lane), kf[ks], s);
        dp = vg_mfma(lfrag_row<HE>(s1, 16 * qt, ks, lane), vf[ks], dp);
      }
      const f32x4 lq4 = *(const f32x4*)(ll + 16 * qt + 4 * g);
      const f32x4 dl4 = *(const f32x4*)(dl + 16 * qt + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 16 * qt + 4 * g + r;
        const float p = ((q < S) && (key < S)) ? __expf(s[r] * scale - lq4[r]) : 0.f;
        pr[qt][r] = p;
        ds[qt][r] = p * (dp[r] - dl4[r]) * scale;
      }
    }
    bf16x8 pf[KP], dsf[KP];
#pragma unroll
    for (int u = 0; u < KP; ++u) {
      const int hi = (2 * u + 1 < NT) ? 2 * u + 1 : 0;
      pf[u] = pack_pair(pr[2 * u], (2 * u + 1 < NT) ? pr[hi] : zero);
      dsf[u] = pack_pair(ds[2 * u], (2 * u + 1 < NT) ? ds[hi] : zero);
    }
    bf16* rowp = dqb + (size_t)(key < S ? key : 0) * ld;
#pragma unroll
    for (int pp = 0; pp < DT / 2; ++pp) {
      f32x4 dv[2] = {zero, zero}, dk[2] = {zero, zero};
#pragma unroll
      for (int u = 0; u < KP; ++u) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          dv[t] = vg_mfma(lfrag_tr<HE>(s1, u, 32 * pp + 16 * t, lane), pf[u], dv[t]);
          dk[t] = vg_mfma(lfrag_tr<HE>(s0, u, 32 * pp + 16 * t, lane), dsf[u], dk[t]);
        }
      }
      store_tiles<2>(rowp + E + 32 * pp, dk, 1.0f, g, key < S);
      store_tiles<2>(rowp + 2 * E + 32 * pp, dv, 1.0f, g, key < S);
    }
  }
}

// ---- the backward OF the attention backward (gradient penalty, src/v2/utils.py:124-144) on the MFMA pipe -------------------
// Forward: P = softmax(s Q K^T), O = P V.  Backward: dV = P^T dO ; dP = dO V^T ; delta_i = sum_j P_ij dP_ij ; dS = P (dP - delta) ;
// dQ = s dS K ; dK = s dS^T Q.  Given (uQ, uK, uV) = dL/d(dQ, dK, dV):
//   G = s (uQ K^T + Q uK^T) ; gam_i = sum_j G_ij P_ij ; H = P (G - gam)                       [dL/d dP]
//   Pi = dO uV^T + G (dP - delta) - gam dP ; pi_i = sum_j P_ij Pi_ij ; Sg = P (Pi - pi)       [dL/d S]
//   d(dO) = P uV + H V ;  d(Q) = s (dS uK + Sg K) ;  d(K) = s (dS^T uQ + Sg^T Q) ;  d(V) = H^T dO.
// Same two-orientation scheme as the first-order backward: phase A (lane = query; K, V, uK, uV as LDS images, this wave's 16 rows of
// Q, dO, uQ as fragments straight from global) computes the five S x S products of a query tile against every key tile in MFMA
// accumulators, the three row sums (delta, gam, pi; left in LDS for phase B) and the query-side outputs d(dO), d(Q); phase B
// (lane = key; Q, dO, uQ staged into the same images, this wave's rows of K, V, uK, uV kept as fragments) recomputes the products in
// the other orientation and yields d(K), d(V).  The S x S matrices never leave registers (round 2's kernel kept four of them as fp32
// in LDS and ran plain FMA loops: 1.2 ms per launch); they enter the output products as bf16, like P and dS in the first-order
// backward.  One workgroup per (image, head), five waves, 74 KB of LDS.
template <int HE>
__global__ __launch_bounds__(320, 2) void vg_attn_bwd_bwd2_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                                  const float* __restrict__ lse, const bf16* __restrict__ uqkv,
                                                                  bf16* __restrict__ d_do, bf16* __restrict__ d_qkv, int B, int S, int H,
                                                                  float scale, const void* __restrict__ zeros) {
  constexpr int NT = 5, KS = HE / 32, DT = HE / 16, KP = 3, RP = 96;
  constexpr int IMG = RP * HE * 2;
  __shared__ __attribute__((aligned(16))) unsigned char sm[4 * IMG + 4 * RP * 4];
  unsigned char* i0 = sm;            // K,  then Q
  unsigned char* i1 = sm + IMG;      // V,  then dO
  unsigned char* i2 = sm + 2 * IMG;  // uK, then uQ
  unsigned char* i3 = sm + 3 * IMG;  // uV
  float* dl = (float*)(sm + 4 * IMG);  // delta[q]
  float* gm = dl + RP;                 // gam[q]
  float* pl = gm + RP;                 // pi[q]
  float* ll = pl + RP;                 // lse[q]
  int b, h;
  if (!attn_block(B, H, b, h)) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int E = H * HE;
  const size_t ld = 3 * (size_t)E;
  const bf16* qb = qkv + (size_t)b * S * ld + h * HE;
  const bf16* kb = qb + E;
  const bf16* vb = qb + 2 * E;
  const bf16* uqb = uqkv + (size_t)b * S * ld + h * HE;
  const bf16* ukb = uqb + E;
  const bf16* uvb = uqb + 2 * E;
  const bf16* dob = d_o + (size_t)b * S * E + h * HE;
  const float* lb = lse + ((size_t)b * H + h) * S;
  bf16* ddob = d_do + (size_t)b * S * E + h * HE;
  bf16* dqb = d_qkv + (size_t)b * S * ld + h * HE;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  dma_head<HE, NT>(i0, kb, ld, S, RP, zeros, wv, lane);
  dma_head<HE, NT>(i1, vb, ld, S, RP, zeros, (wv + 1) % NT, lane);
  dma_head<HE, NT>(i2, ukb, ld, S, RP, zeros, (wv + 2) % NT, lane);
  dma_head<HE, NT>(i3, uvb, ld, S, RP, zeros, (wv + 3) % NT, lane);
  bf16x8 qf[KS], dof[KS], uqf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = gfrag(qb, ld, 16 * wv, ks, S, lane);
    dof[ks] = gfrag(dob, (size_t)E, 16 * wv, ks, S, lane);
    uqf[ks] = gfrag(uqb, ld, 16 * wv, ks, S, lane);
  }
  for (int i = tid; i < RP; i += 64 * NT) ll[i] = (i < S) ? lb[i] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------- phase A: lane = query li of tile wv; keys 16 kt + 4 g + r -------------------------------------------
  {
    const int q = 16 * wv + li;
    const float lse_q = ll[q];
    f32x4 Pm[NT], Dm[NT], Gm[NT], Tm[NT];
    float delta = 0.f, gam = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      f32x4 st = zero, dpt = zero, gt = zero, tt = zero;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kr = lfrag_row<HE>(i0, 16 * kt, ks, lane);
        st = vg_mfma(kr, qf[ks], st);
        gt = vg_mfma(kr, uqf[ks], gt);
        gt = vg_mfma(lfrag_row<HE>(i2, 16 * kt, ks, lane), qf[ks], gt);
        dpt = vg_mfma(lfrag_row<HE>(i1, 16 * kt, ks, lane), dof[ks], dpt);
        tt = vg_mfma(lfrag_row<HE>(i3, 16 * kt, ks, lane), dof[ks], tt);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * g + r;
        const float p = (key < S && q < S) ? __expf(st[r] * scale - lse_q) : 0.f;
        const float gv = gt[r] * scale;
        Pm[kt][r] = p; Dm[kt][r] = dpt[r]; Gm[kt][r] = gv; Tm[kt][r] = tt[r];
        delta += p * dpt[r];
        gam += p * gv;
      }
    }
    delta = group_sum(delta);
    gam = group_sum(gam);
    float pis = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = Dm[kt][r] - delta;
        const float pie = Tm[kt][r] + Gm[kt][r] * a - gam * Dm[kt][r];
        Tm[kt][r] = pie;                 // Pi
        Dm[kt][r] = a;                   // dP - delta
        pis += Pm[kt][r] * pie;
      }
    pis = group_sum(pis);
    if (g == 0) { dl[q] = delta; gm[q] = gam; pl[q] = pis; }
    // H = P (G - gam) -> Gm ;  dS = P (dP - delta) -> Dm ;  Sg = P (Pi - pi) -> Tm
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = Pm[kt][r];
        Gm[kt][r] = p * (Gm[kt][r] - gam);
        Dm[kt][r] = p * Dm[kt][r];
        Tm[kt][r] = p * (Tm[kt][r] - pis);
      }
    bf16x8 pf[KP], hf[KP], dsf[KP], sgf[KP];
#pragma unroll
    for (int u = 0; u < KP; ++u) {
      const bool two = 2 * u + 1 < NT;
      const int hi = two ? 2 * u + 1 : 0;
      pf[u] = pack_pair(Pm[2 * u], two ? Pm[hi] : zero);
      hf[u] = pack_pair(Gm[2 * u], two ? Gm[hi] : zero);
      dsf[u] = pack_pair(Dm[2 * u], two ? Dm[hi] : zero);
      sgf[u] = pack_pair(Tm[2 * u], two ? Tm[hi] : zero);
    }
    bf16* rowo = ddob + (size_t)(q < S ? q : 0) * E;
    bf16* rowq = dqb + (size_t)(q < S ? q : 0) * ld;
#pragma unroll
    for (int pp = 0; pp < DT / 2; ++pp) {  // two head-dim tiles at a time (one 16-byte store per lane)
      f32x4 o1[2] = {zero, zero}, o2[2] = {zero, zero};
#pragma unroll
      for (int u = 0; u < KP; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int d0 = 32 * pp + 16 * t;
          o1[t] = vg_mfma(lfrag_tr<HE>(i3, u, d0, lane), pf[u], o1[t]);   // P uV
          o1[t] = vg_mfma(lfrag_tr<HE>(i1, u, d0, lane), hf[u], o1[t]);   // + H V
          o2[t] = vg_mfma(lfrag_tr<HE>(i2, u, d0, lane), dsf[u], o2[t]);  // dS uK
          o2[t] = vg_mfma(lfrag_tr<HE>(i0, u, d0, lane), sgf[u], o2[t]);  // + Sg K
        }
      store_tiles<2>(rowo + 32 * pp, o1, 1.0f, g, q < S);
      store_tiles<2>(rowq + 32 * pp, o2, scale, g, q < S);
    }
  }
  // this wave's key tile for phase B, before Q, dO, uQ take the images over
  bf16x8 kf[KS], vf[KS], ukf[KS], uvf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = lfrag_row<HE>(i0, 16 * wv, ks, lane);
    vf[ks] = lfrag_row<HE>(i1, 16 * wv, ks, lane);
    ukf[ks] = lfrag_row<HE>(i2, 16 * wv, ks, lane);
    uvf[ks] = lfrag_row<HE>(i3, 16 * wv, ks, lane);
  }
  __syncthreads();  // every wave is done with K, V, uK, uV (and delta / gam / pi of every query are in LDS)
  dma_head<HE, NT>(i0, qb, ld, S, RP, zeros, wv, lane);
  dma_head<HE, NT>(i1, dob, (size_t)E, S, RP, zeros, (wv + 1) % NT, lane);
  dma_head<HE, NT>(i2, uqb, ld, S, RP, zeros, (wv + 2) % NT, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---------------- phase B: lane = key li of tile wv; queries 16 qt + 4 g + r -------------------------------------------
  {
    const int key = 16 * wv + li;
    f32x4 Hm[NT], Dm[NT], Sm[NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      f32x4 st = zero, dpt = zero, gt = zero, tt = zero;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 qr = lfrag_row<HE>(i0, 16 * qt, ks, lane);
        const bf16x8 dr = lfrag_row<HE>(i1, 16 * qt, ks, lane);
        st = vg_mfma(qr, kf[ks], st);
        gt = vg_mfma(qr, ukf[ks], gt);
        gt = vg_mfma(lfrag_row<HE>(i2, 16 * qt, ks, lane), kf[ks], gt);
        dpt = vg_mfma(dr, vf[ks], dpt);
        tt = vg_mfma(dr, uvf[ks], tt);
      }
      const f32x4 lq4 = *(const f32x4*)(ll + 16 * qt + 4 * g);
      const f32x4 dl4 = *(const f32x4*)(dl + 16 * qt + 4 * g);
      const f32x4 gm4 = *(const f32x4*)(gm + 16 * qt + 4 * g);
      const f32x4 pl4 = *(const f32x4*)(pl + 16 * qt + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 16 * qt + 4 * g + r;
        const float p = ((q < S) && (key < S)) ? __expf(st[r] * scale - lq4[r]) : 0.f;
        const float gv = gt[r] * scale, a = dpt[r] - dl4[r];
        const float pie = tt[r] + gv * a - gm4[r] * dpt[r];
        Hm[qt][r] = p * (gv - gm4[r]);
        Dm[qt][r] = p * a;
        Sm[qt][r] = p * (pie - pl4[r]);
      }
    }
    bf16x8 hf[KP], dsf[KP], sgf[KP];
#pragma unroll
    for (int u = 0; u < KP; ++u) {
      const bool two = 2 * u + 1 < NT;
      const int hi = two ? 2 * u + 1 : 0;
      hf[u] = pack_pair(Hm[2 * u], two ? Hm[hi] : zero);
      dsf[u] = pack_pair(Dm[2 * u], two ? Dm[hi] : zero);
      sgf[u] = pack_pair(Sm[2 * u], two ? Sm[hi] : zero);
    }
    bf16* rowp = dqb + (size_t)(key < S ? key : 0) * ld;
#pragma unroll
    for (int pp = 0; pp < DT / 2; ++pp) {
      f32x4 dk[2] = {zero, zero}, dv[2] = {zero, zero};
#pragma unroll
      for (int u = 0; u < KP; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int d0 = 32 * pp + 16 * t;
          dk[t] = vg_mfma(lfrag_tr<HE>(i2, u, d0, lane), dsf[u], dk[t]);  // dS^T uQ
          dk[t] = vg_mfma(lfrag_tr<HE>(i0, u, d0, lane), sgf[u], dk[t]);  // + Sg^T Q
          dv[t] = vg_mfma(lfrag_tr<HE>(i1, u, d0, lane), hf[u], dv[t]);   // H^T dO
        }
      store_tiles<2>(rowp + E + 32 * pp, dk, scale, g, key < S);
      store_tiles<2>(rowp + 2 * E + 32 * pp, dv, 1.0f, g, key < S);
    }
  }
}

__device__ __attribute__((aligned(16))) unsigned int vg_attn_zero_page[4] = {0u, 0u, 0u, 0u};
static const void* attn_zeros() {
  static void* zp = nullptr;  // one device per process
  if (!zp && hipGetSymbolAddress(&zp, HIP_SYMBOL(vg_attn_zero_page)) != hipSuccess) zp = nullptr;
  return zp;
}

template <int HE, int NT, int MODE>  // MODE: 0 dot-product bf16, 1 L2-distance scores, 2 dot-product with fp8 activation products
static int launch_fwd(const bf16* qkv, bf16* o, float* lse, int B, int H, int S, float scale, hipStream_t st) {
  const void* z = attn_zeros();
  if (!z) return -5;
  hipLaunchKernelGGL((vg_attn_fwd_kernel<HE, NT, MODE == 1, MODE == 2>), dim3(attn_grid(B, H)), dim3(64 * NT), 0, st, qkv, o, lse, B, S, H, scale, z);
  return (int)hipGetLastError();
}
template <int HE, int NT, int MODE>
static int launch_bwd(const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse, bf16* dqkv, int B, int H,
                      int S, float scale, hipStream_t st) {
  const void* z = attn_zeros();
  if (!z) return -5;
#ifndef VG_ATTN_BWD_4IMG  // A/B builds (make var DEFS=-DVG_ATTN_BWD_4IMG): the four-image kernel everywhere
  if constexpr (MODE != 1 && NT == 5) {
    hipLaunchKernelGGL((vg_attn_bwd2_kernel<HE, NT, MODE == 2>), dim3(attn_grid(B, H)), dim3(64 * NT), 0, st, qkv, o, d_o, lse, dqkv, B, S, H, scale, z);
    return (int)hipGetLastError();
  }
#endif
  hipLaunchKernelGGL((vg_attn_bwd_kernel<HE, NT, MODE == 1, MODE == 2>), dim3(attn_grid(B, H)), dim3(64 * NT), 0, st, qkv, o, d_o, lse, dqkv, B, S, H, scale, z);
  return (int)hipGetLastError();
}

// Supported shapes: head dim 32, 64 or 96; S <= 32 (2 tiles) or S <= 80 (5 tiles).
// mode = 0: dot-product scores (src/v2/modules.py:142-155, v1 lp = 1); 1: Euclidean-distance scores (v1 lp = 2);
// 2: dot-product scores with fp8 (e4m3) operands for Q.K^T (forward and recompute) and P.V (config C5).
#define VG_ATTN_BY_NT(FN, HE_, MODE_, ...) return small ? FN<HE_, 2, MODE_>(__VA_ARGS__) : FN<HE_, 5, MODE_>(__VA_ARGS__)
#define VG_ATTN_BY_MODE(FN, HE_, ...)                                  \
  do {                                                                 \
    if (mode == 0) { VG_ATTN_BY_NT(FN, HE_, 0, __VA_ARGS__); }         \
    if (mode == 1) { VG_ATTN_BY_NT(FN, HE_, 1, __VA_ARGS__); }         \
    VG_ATTN_BY_NT(FN, HE_, 2, __VA_ARGS__);                            \
  } while (0)
#define VG_ATTN_DISPATCH(FN, ...)                                      \
  do {                                                                 \
    if (S < 1 || S > 80 || B < 1 || H < 1) return -2;                  \
    if (mode < 0 || mode > 2) return -4;                               \
    const bool small = (S <= 32);                                      \
    if (HE == 96) VG_ATTN_BY_MODE(FN, 96, __VA_ARGS__);                \
    if (HE == 64) VG_ATTN_BY_MODE(FN, 64, __VA_ARGS__);                \
    if (HE == 32) VG_ATTN_BY_MODE(FN, 32, __VA_ARGS__);                \
    return -3;                                                         \
  } while (0)
int vg_attn_fwd_launch(const bf16* qkv, bf16* o, float* lse, int B, int H, int S, int HE, float scale, int mode, hipStream_t st) {
  VG_ATTN_DISPATCH(launch_fwd, qkv, o, lse, B, H, S, scale, st);
}
int vg_attn_bwd_launch(const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse, bf16* dqkv, int B, int H,
                       int S, int HE, float scale, int mode, hipStream_t st) {
  VG_ATTN_DISPATCH(launch_bwd, qkv, o, d_o, lse, dqkv, B, H, S, scale, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention of the TOP encoder block as the classifier sees it (src/v2/modules.py:195 reads the CLS row only): ONE query - row 0 of
// every image - against all S keys.  Forward: o_cls[b, h*HE ..] = softmax(s q0 K^T) V; backward: dO is nonzero for that query
// only, so dK_j = ds_j q0, dV_j = p_j dO_0 are rank-one and dQ is zero off row 0.  Same arithmetic as the full kernels at that
// row (fp32 scores of bf16 operands, p and ds rounded to bf16 where the full kernels make them MFMA operands, 1 / l applied to
// the fp32 sum), 1/65 of their products and half their bytes (K and V in, dK and dV out).  One wave per (image, head).  (A first
// version with lanes over the KEYS - a lane reading and writing its own rows in 16-byte chunks, 64 row segments per instruction -
// took 65 us for the 2B backward, more than the full kernel's 57.)
// ---------------------------------------------------------------------------------------------------------------------
// Work layout of both kernels: a wave walks the head's K / V rows RPI at a time, lane = (row rr = lane / CPR, 16-byte chunk c = lane % CPR),
// so every load and store of a wave-instruction covers RPI whole 2 HE-byte row segments (HE = 96: 60 of the 64 lanes, 5 rows).
template <int HE>
__global__ __launch_bounds__(64) void vg_attn_cls_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o_cls, float* __restrict__ lse_cls,
                                                             int B, int S, int H, float scale) {
  constexpr int CPR = HE / 8, RPI = 64 / CPR;
  __shared__ float q0l[HE];
  __shared__ float part[128 * CPR];  // per (key, chunk) partial dot products; then the per-lane partial outputs
  __shared__ float pl[128];
  int b, h;
  if (!attn_block(B, H, b, h)) return;
  const int lane = threadIdx.x;
  const int rr = lane / CPR, c = lane - rr * CPR;
  const bool act = rr < RPI;
  const int E = H * HE;
  const size_t ld = 3 * (size_t)E;
  const bf16* qb = qkv + (size_t)b * S * ld + h * HE;
  const bf16* kb = qb + E;
  const bf16* vb = qb + 2 * E;
  for (int d = lane; d < HE; d += 64) q0l[d] = vg_bf2f(qb[d]);
  __syncthreads();
  float qc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) qc[t] = q0l[8 * (act ? c : 0) + t];
  for (int r0 = 0; r0 < S; r0 += RPI) {
    const int r = r0 + rr;
    if (act && r < S) {
      const bf16x8 k8 = *(const bf16x8*)(kb + (size_t)r * ld + 8 * c);
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) a = fmaf(qc[t], vg_bf2f(k8[t]), a);
      part[r * CPR + c] = a;
    }
  }
  __syncthreads();
  float sv[2];
#pragma unroll
  for (int rnd = 0; rnd < 2; ++rnd) {
    const int j = lane + 64 * rnd;
    float a = 0.f;
    if (j < S) {
#pragma unroll
      for (int cc = 0; cc < CPR; ++cc) a += part[j * CPR + cc];
    }
    sv[rnd] = (j < S) ? a * scale : -INFINITY;
  }
  const float m = vg_wave_max(fmaxf(sv[0], sv[1]));
  const float p0 = __expf(sv[0] - m), p1 = __expf(sv[1] - m);  // exp(-inf) = 0 for the padded keys
  const float l = vg_wave_sum(p0 + p1);
  pl[lane] = vg_bf2f(vg_f2bf(p0));       // the full kernel multiplies V by bf16(p) (an MFMA operand) and divides the fp32 sum by l
  pl[lane + 64] = vg_bf2f(vg_f2bf(p1));
  if (lane == 0) lse_cls[(size_t)b * H + h] = m + __logf(l);
  __syncthreads();
  // o[d] = sum_j p_j V[j][d] / l: every lane sums its chunk over its rows, then the RPI row slots are added up through LDS
  float oa[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) oa[t] = 0.f;
  for (int r0 = 0; r0 < S; r0 += RPI) {
    const int r = r0 + rr;
    if (act && r < S) {
      const bf16x8 v8 = *(const bf16x8*)(vb + (size_t)r * ld + 8 * c);
      const float pj = pl[r];
#pragma unroll
      for (int t = 0; t < 8; ++t) oa[t] = fmaf(pj, vg_bf2f(v8[t]), oa[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) part[lane * 8 + t] = oa[t];
  __syncthreads();
  const float inv_l = 1.0f / l;
  for (int d = lane; d < HE; d += 64) {
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < RPI; ++q) a += part[(q * CPR + (d >> 3)) * 8 + (d & 7)];
    o_cls[(size_t)b * E + h * HE + d] = vg_f2bf(a * inv_l);
  }
}

template <int HE>
__global__ __launch_bounds__(64) void vg_attn_cls_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o_cls, const bf16* __restrict__ do_cls,
                                                             const float* __restrict__ lse_cls, bf16* __restrict__ dqkv, int B, int S, int H,
                                                             float scale) {
  constexpr int CPR = HE / 8, RPI = 64 / CPR;
  __shared__ float q0l[HE], d0l[HE];
  __shared__ float ps[128 * CPR], pd[128 * CPR];  // per (key, chunk) partial dot products q0.K and dO0.V; ps then holds the partial dQ
  __shared__ float dsl[128], pbl[128];
  int b, h;
  if (!attn_block(B, H, b, h)) return;
  const int lane = threadIdx.x;
  const int rr = lane / CPR, c = lane - rr * CPR;
  const bool act = rr < RPI;
  const int E = H * HE;
  const size_t ld = 3 * (size_t)E;
  const bf16* qb = qkv + (size_t)b * S * ld + h * HE;
  const bf16* kb = qb + E;
  const bf16* vb = qb + 2 * E;
  bf16* dqb = dqkv + (size_t)b * S * ld + h * HE;
  const float lse0 = lse_cls[(size_t)b * H + h];
  float dpart = 0.f;  // delta = sum_d dO_0[d] O_0[d]
  for (int d = lane; d < HE; d += 64) {
    q0l[d] = vg_bf2f(qb[d]);
    const float dv = vg_bf2f(do_cls[(size_t)b * E + h * HE + d]);
    d0l[d] = dv;
    dpart = fmaf(dv, vg_bf2f(o_cls[(size_t)b * E + h * HE + d]), dpart);
  }
  const float delta = vg_wave_sum(dpart);
  __syncthreads();
  float qc[8], dc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) { qc[t] = q0l[8 * (act ? c : 0) + t]; dc[t] = d0l[8 * (act ? c : 0) + t]; }
  for (int r0 = 0; r0 < S; r0 += RPI) {
    const int r = r0 + rr;
    if (act && r < S) {
      const bf16x8 k8 = *(const bf16x8*)(kb + (size_t)r * ld + 8 * c);
      const bf16x8 v8 = *(const bf16x8*)(vb + (size_t)r * ld + 8 * c);
      float a = 0.f, e = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) { a = fmaf(qc[t], vg_bf2f(k8[t]), a); e = fmaf(dc[t], vg_bf2f(v8[t]), e); }
      ps[r * CPR + c] = a;
      pd[r * CPR + c] = e;
    }
  }
  __syncthreads();
#pragma unroll
  for (int rnd = 0; rnd < 2; ++rnd) {
    const int j = lane + 64 * rnd;
    float dsv = 0.f, pb = 0.f;
    if (j < S) {
      float sc = 0.f, dp = 0.f;
#pragma unroll
      for (int cc = 0; cc < CPR; ++cc) { sc += ps[j * CPR + cc]; dp += pd[j * CPR + cc]; }
      const float p = __expf(sc * scale - lse0);
      dsv = vg_bf2f(vg_f2bf(p * (dp - delta) * scale));  // bf16: MFMA operands in the full kernel
      pb = vg_bf2f(vg_f2bf(p));
    }
    dsl[j] = dsv;
    pbl[j] = pb;
  }
  __syncthreads();
  // dK_j = ds_j q0, dV_j = p_j dO_0 (rank one in the CLS query); dQ_0 = sum_j ds_j K_j
  float qa[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) qa[t] = 0.f;
  for (int r0 = 0; r0 < S; r0 += RPI) {
    const int r = r0 + rr;
    if (act && r < S) {
      const bf16x8 k8 = *(const bf16x8*)(kb + (size_t)r * ld + 8 * c);
      const float dj = dsl[r], pj = pbl[r];
      bf16x8 dk8, dv8;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        dk8[t] = vg_f2bf(dj * qc[t]);
        dv8[t] = vg_f2bf(pj * dc[t]);
        qa[t] = fmaf(dj, vg_bf2f(k8[t]), qa[t]);
      }
      bf16* rowp = dqb + (size_t)r * ld + 8 * c;
      *(bf16x8*)(rowp + E) = dk8;
      *(bf16x8*)(rowp + 2 * E) = dv8;
      if (r > 0) *(u32x4*)rowp = (u32x4){0u, 0u, 0u, 0u};  // dQ is zero off the CLS row
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) ps[lane * 8 + t] = qa[t];
  __syncthreads();
  for (int d = lane; d < HE; d += 64) {
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < RPI; ++q) a += ps[(q * CPR + (d >> 3)) * 8 + (d & 7)];
    dqb[d] = vg_f2bf(a);
  }
}

// o_cls / do_cls: [B, E] (the CLS rows, compact); lse_cls: [B, H]; dqkv: the full [B*S, 3E] gradient.  Dot-product scores only
// (the engine keeps the full kernels for the top block when the fp8 mode is on); -3: head dim not 32 / 64 / 96.
int vg_attn_cls_fwd_launch(const bf16* qkv, bf16* o_cls, float* lse_cls, int B, int H, int S, int HE, float scale, hipStream_t st) {
  if (S < 1 || S > 128 || B < 1 || H < 1) return -2;
#define VG_ACLS(HE_) hipLaunchKernelGGL((vg_attn_cls_fwd_kernel<HE_>), dim3(attn_grid(B, H)), dim3(64), 0, st, qkv, o_cls, lse_cls, B, S, H, scale)
  if (HE == 96) VG_ACLS(96); else if (HE == 64) VG_ACLS(64); else if (HE == 32) VG_ACLS(32); else return -3;
#undef VG_ACLS
  return (int)hipGetLastError();
}
int vg_attn_cls_bwd_launch(const bf16* qkv, const bf16* o_cls, const bf16* do_cls, const float* lse_cls, bf16* dqkv, int B, int H, int S, int HE,
                           float scale, hipStream_t st) {
  if (S < 1 || S > 128 || B < 1 || H < 1) return -2;
#define VG_ACLS(HE_) hipLaunchKernelGGL((vg_attn_cls_bwd_kernel<HE_>), dim3(attn_grid(B, H)), dim3(64), 0, st, qkv, o_cls, do_cls, lse_cls, dqkv, B, S, H, scale)
  if (HE == 96) VG_ACLS(96); else if (HE == 64) VG_ACLS(64); else if (HE == 32) VG_ACLS(32); else return -3;
#undef VG_ACLS
  return (int)hipGetLastError();
}

// the second-order kernel above (S <= 80: padded to five 16-row tiles); -3: head dim not 32 / 64 / 96
int vg_attn_bwd_bwd_mfma_launch(const bf16* qkv, const bf16* d_o, const float* lse, const bf16* uqkv, bf16* d_do, bf16* d_qkv, int B, int H,
                                int S, int HE, float scale, hipStream_t st) {
  if (S < 1 || S > 80 || B < 1 || H < 1) return -2;
  const void* z = attn_zeros();
  if (!z) return -5;
#define VG_ABB2(HE_) hipLaunchKernelGGL((vg_attn_bwd_bwd2_kernel<HE_>), dim3(attn_grid(B, H)), dim3(320), 0, st, qkv, d_o, lse, uqkv, d_do, d_qkv, B, S, H, scale, z)
  if (HE == 96) VG_ABB2(96); else if (HE == 64) VG_ABB2(64); else if (HE == 32) VG_ABB2(32); else return -3;
#undef VG_ABB2
  return (int)hipGetLastError();
}
