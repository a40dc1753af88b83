// Fused multi-head self-attention (forward and backward) for short sequences (S <= 80): one workgroup
// per (image, head), ONE WAVE PER 16-ROW TILE (5 waves for S = 65) so that a CU holds 15-20 waves -
// these kernels are latency bound and occupancy is what hides it.  gfx950 only.
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

template <int HE>
__device__ __forceinline__ int lds_off(int r, int d) {
  // row-major [rows][HE] bf16 with the 32-B chunk index XOR-swizzled by the row so that the
  // 8 consecutive rows one half-wave touches in a transposed read fall on 64 distinct banks.
  const int sw = (HE == 64) ? ((r >> 1) & 3) : ((r >> 2) & 1);  // HE = 96 / 32: rows realign every 4
  return r * (HE * 2) + ((((d >> 4) ^ sw)) << 5) + ((d & 15) << 1);
}

// stage rows [0, rows_alloc) x HE of one head into LDS; rows >= S are zero-filled
template <int HE>
__device__ __forceinline__ void stage_head(unsigned char* lds, const bf16* __restrict__ src, size_t ld,
                                           int S, int rows_alloc, int tid, int nthreads) {
  constexpr int CPR = HE / 8;  // 16-B chunks per row
  for (int idx = tid; idx < rows_alloc * CPR; idx += nthreads) {
    const int r = idx / CPR, c = idx - r * CPR;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < S) v = *(const u32x4*)(src + (size_t)r * ld + 8 * c);
    *(u32x4*)(lds + lds_off<HE>(r, 8 * c)) = v;
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

template <int HE, int NT>
__global__ __launch_bounds__(64 * NT) void vg_attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                              float* __restrict__ lse, int S, int H, float scale) {
  constexpr int KS = HE / 32, DT = HE / 16, KP = (NT + 1) / 2, RP = KP * 32;
  __shared__ __attribute__((aligned(16))) unsigned char vl[RP * HE * 2];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, lane = tid & 63, qt = tid >> 6;  // wave qt owns query rows 16*qt .. 16*qt+15
  const int g = lane >> 4, li = lane & 15;
  const int E = H * HE;
  const size_t ld = 3 * (size_t)E;
  const bf16* qb = qkv + (size_t)b * S * ld + h * HE;
  const bf16* kb = qb + E;
  const bf16* vb = qb + 2 * E;

  stage_head<HE>(vl, vb, ld, S, RP, tid, 64 * NT);

  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = gfrag(qb, ld, 16 * qt, ks, S, lane);

  f32x4 sc[NT];  // [kt]: rows = keys 16kt+4g+r, col = query 16qt+li
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a = vg_mfma(gfrag(kb, ld, 16 * kt, ks, S, lane), qf[ks], a);
    sc[kt] = a;
  }
  const int q = 16 * qt + li;
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * kt + 4 * g + r;
      const float sv = (key < S) ? sc[kt][r] * scale : -INFINITY;
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

  __syncthreads();  // V image complete
  f32x4 oa[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) oa[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < KP; ++u) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 pf = pack_pair(sc[2 * u], (2 * u + 1 < NT) ? sc[(2 * u + 1 < NT) ? 2 * u + 1 : 0] : zero);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oa[dt] = vg_mfma(lfrag_tr<HE>(vl, u, 16 * dt, lane), pf, oa[dt]);
  }
  if (q < S) {
    bf16* op = o + ((size_t)b * S + q) * E + h * HE + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      bf16x4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = vg_f2bf(oa[dt][r] * inv_l);
      *(bf16x4*)(op + 16 * dt) = w;
    }
  }
}

// Backward.  Phase A works in the S^T orientation (lane = query) and yields dQ; phase B in the
// S orientation (lane = key) and yields dK, dV.  Recomputing the 65x65 tile in both orientations
// costs 2 x 75 extra MFMAs per head and removes every register transpose.
template <int HE, int NT>
__global__ __launch_bounds__(64 * NT, 4) void vg_attn_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                         const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                         bf16* __restrict__ dqkv, int S, int H, float scale) {
  constexpr int KS = HE / 32, DT = HE / 16, KP = (NT + 1) / 2, RP = KP * 32;
  constexpr int IMG = RP * HE * 2;
  __shared__ __attribute__((aligned(16))) unsigned char sm[2 * IMG + RP * 4];
  unsigned char* l0 = sm;
  unsigned char* l1 = sm + IMG;
  float* dl = (float*)(sm + 2 * IMG);  // delta[q] = sum_d dO*O
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
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

  // ---------------- phase A: K, V in LDS; loop over query tiles -> dQ ----------------------
  stage_head<HE>(l0, kb, ld, S, RP, tid, 64 * NT);
  stage_head<HE>(l1, vb, ld, S, RP, tid, 64 * NT);
  for (int i = tid; i < RP; i += 64 * NT) dl[i] = 0.f;
  __syncthreads();
  {
    const int qt = wv;
    const int q = 16 * qt + li;
    bf16x8 qf[KS], dof[KS];
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = gfrag(qb, ld, 16 * qt, ks, S, lane);
      dof[ks] = gfrag(dob, (size_t)E, 16 * qt, ks, S, lane);
      const bf16x8 of = gfrag(ob, (size_t)E, 16 * qt, ks, S, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) dpart += vg_bf2f(dof[ks][j]) * vg_bf2f(of[j]);
    }
    const float delta = group_sum(dpart);
    if (g == 0) dl[q] = delta;
    const float lq = (q < S) ? lb[q] : 0.f;
    f32x4 ds[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      f32x4 st = zero, dpt = zero;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        st = vg_mfma(lfrag_row<HE>(l0, 16 * kt, ks, lane), qf[ks], st);
        dpt = vg_mfma(lfrag_row<HE>(l1, 16 * kt, ks, lane), dof[ks], dpt);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * g + r;
        const float p = (key < S && q < S) ? __expf(st[r] * scale - lq) : 0.f;
        ds[kt][r] = p * (dpt[r] - delta) * scale;
      }
    }
    f32x4 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = zero;
#pragma unroll
    for (int u = 0; u < KP; ++u) {
      const bf16x8 dsf = pack_pair(ds[2 * u], (2 * u + 1 < NT) ? ds[(2 * u + 1 < NT) ? 2 * u + 1 : 0] : zero);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dq[dt] = vg_mfma(lfrag_tr<HE>(l0, u, 16 * dt, lane), dsf, dq[dt]);
    }
    if (q < S) {
      bf16* p = dqb + (size_t)q * ld + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16x4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = vg_f2bf(dq[dt][r]);
        *(bf16x4*)(p + 16 * dt) = w;
      }
    }
  }
  __syncthreads();
  // ---------------- phase B: Q, dO in LDS; loop over key tiles -> dK, dV ---------------------
  stage_head<HE>(l0, qb, ld, S, RP, tid, 64 * NT);
  stage_head<HE>(l1, dob, (size_t)E, S, RP, tid, 64 * NT);
  __syncthreads();
  {
    const int kt = wv;
    const int key = 16 * kt + li;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = gfrag(kb, ld, 16 * kt, ks, S, lane);
      vf[ks] = gfrag(vb, ld, 16 * kt, ks, S, lane);
    }
    f32x4 pr[NT], ds[NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      f32x4 s = zero, dp = zero;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s = vg_mfma(lfrag_row<HE>(l0, 16 * qt, ks, lane), kf[ks], s);
        dp = vg_mfma(lfrag_row<HE>(l1, 16 * qt, ks, lane), vf[ks], dp);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 16 * qt + 4 * g + r;
        const bool ok = (q < S) && (key < S);
        const float lq = (q < S) ? lb[q] : 0.f;
        const float p = ok ? __expf(s[r] * scale - lq) : 0.f;
        pr[qt][r] = p;
        ds[qt][r] = p * (dp[r] - dl[q]) * scale;
      }
    }
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
        dv[dt] = vg_mfma(lfrag_tr<HE>(l1, u, 16 * dt, lane), pf, dv[dt]);
        dk[dt] = vg_mfma(lfrag_tr<HE>(l0, u, 16 * dt, lane), dsf, dk[dt]);
      }
    }
    if (key < S) {
      bf16* pk = dqb + (size_t)key * ld + E + 4 * g;
      bf16* pv = dqb + (size_t)key * ld + 2 * E + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16x4 wk, wv;
#pragma unroll
        for (int r = 0; r < 4; ++r) { wk[r] = vg_f2bf(dk[dt][r]); wv[r] = vg_f2bf(dv[dt][r]); }
        *(bf16x4*)(pk + 16 * dt) = wk;
        *(bf16x4*)(pv + 16 * dt) = wv;
      }
    }
  }
}

template <int HE, int NT>
static int launch_fwd(const bf16* qkv, bf16* o, float* lse, int B, int H, int S, float scale, hipStream_t st) {
  hipLaunchKernelGGL((vg_attn_fwd_kernel<HE, NT>), dim3(B * H), dim3(64 * NT), 0, st, qkv, o, lse, S, H, scale);
  return (int)hipGetLastError();
}
template <int HE, int NT>
static int launch_bwd(const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse, bf16* dqkv, int B, int H,
                      int S, float scale, hipStream_t st) {
  hipLaunchKernelGGL((vg_attn_bwd_kernel<HE, NT>), dim3(B * H), dim3(64 * NT), 0, st, qkv, o, d_o, lse, dqkv, S, H, scale);
  return (int)hipGetLastError();
}

// Supported shapes: head dim 64 or 96; S <= 32 (2 tiles) or S <= 80 (5 tiles).
int vg_attn_fwd_launch(const bf16* qkv, bf16* o, float* lse, int B, int H, int S, int HE, float scale, hipStream_t st) {
  if (S < 1 || S > 80 || B < 1 || H < 1) return -2;
  const bool small = (S <= 32);
  if (HE == 96) return small ? launch_fwd<96, 2>(qkv, o, lse, B, H, S, scale, st) : launch_fwd<96, 5>(qkv, o, lse, B, H, S, scale, st);
  if (HE == 64) return small ? launch_fwd<64, 2>(qkv, o, lse, B, H, S, scale, st) : launch_fwd<64, 5>(qkv, o, lse, B, H, S, scale, st);
  if (HE == 32) return small ? launch_fwd<32, 2>(qkv, o, lse, B, H, S, scale, st) : launch_fwd<32, 5>(qkv, o, lse, B, H, S, scale, st);
  return -3;
}
int vg_attn_bwd_launch(const bf16* qkv, const bf16* o, const bf16* d_o, const float* lse, bf16* dqkv, int B, int H,
                       int S, int HE, float scale, hipStream_t st) {
  if (S < 1 || S > 80 || B < 1 || H < 1) return -2;
  const bool small = (S <= 32);
  if (HE == 96) return small ? launch_bwd<96, 2>(qkv, o, d_o, lse, dqkv, B, H, S, scale, st) : launch_bwd<96, 5>(qkv, o, d_o, lse, dqkv, B, H, S, scale, st);
  if (HE == 64) return small ? launch_bwd<64, 2>(qkv, o, d_o, lse, dqkv, B, H, S, scale, st) : launch_bwd<64, 5>(qkv, o, d_o, lse, dqkv, B, H, S, scale, st);
  if (HE == 32) return small ? launch_bwd<32, 2>(qkv, o, d_o, lse, dqkv, B, H, S, scale, st) : launch_bwd<32, 5>(qkv, o, d_o, lse, dqkv, B, H, S, scale, st);
  return -3;
}
