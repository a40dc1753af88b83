// bf16 MFMA GEMM family for the ViTGAN hot path (gfx950).
//
// One kernel template, three operand forms (vg_gemm.h).  Tile 64*WM (m) x 128 (n) x 32 (k): WM = 2 -> 128 x 128,
// 4 waves as 2(m) x 2(n); WM = 4 -> 256 x 128, 8 waves as 4(m) x 2(n); each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
// LDS: a ring of NSTAGE stages filled by LDS-DMA (global_load_lds_dwordx4); NSTAGE-1 k-steps stay in flight across a
// raw s_barrier behind a COUNTED s_waitcnt vmcnt.  Fragment reads are inline asm so that the compiler's conservative
// `s_waitcnt vmcnt(0)` between an LDS-DMA and the next ds_read (it cannot tell the stages apart) does not drain the
// ring at every step.  What bounds the family is the L2 -> LDS staging rate (~10 TB/s chip-wide):
// FLOP/s follow flop per staged byte, i.e. the tile size.
//
// The MFMA is issued "swapped": its A operand carries the GEMM's n index and its B operand the
// m index, so a lane's 4 accumulator registers are 4 CONSECUTIVE n of one output row m; the epilogue pairs
// n-tiles with v_permlane16_swap and stores 16 B (bf16) / 32 B (fp32) per lane straight from registers.
//
// Operand images in LDS:
//   row form  [rows][32 k]  (64-B rows): 16-B chunk c of row r lives at chunk c ^ F[(r>>2)&3],
//             F = {0,2,3,1}: the four 16-lane groups of a ds_read_b128 then each cover all 16
//             slots of a 256-B bank row (conflict-free).
//   tr  form  [32 k][128 cols]  (256-B rows): 16-B chunk c of row k lives at c ^ (2*sigma(k)),
//             sigma(k) = (k&3) | ((k>>3)&1)<<2; fragments by two ds_read_b64_tr_b16, whose 32-lane
//             halves then touch all 64 banks exactly once.
#include "vg_gemm.h"
#include <stdlib.h>

#define BM (64 * WM)   // WM wave-rows: 2 -> 128 x 128 tile (256 threads), 4 -> 256 x 128 tile (512 threads)
#define BN 128
#define BK 32           // k per LDS stage = one MFMA k-step
#ifndef NSTAGE_WM2
#define NSTAGE_WM2 2
#endif
#ifndef NSTAGE_WM4
#define NSTAGE_WM4 3
#endif
#ifndef OCC_WM4
#define OCC_WM4 4
#endif
#ifndef NSTAGE_TN
#define NSTAGE_TN 3      // weight gradients: long k loops at 2-3 workgroups per CU - a deeper ring instead of occupancy
#endif
#define NSTAGE (MODE == VG_TN ? NSTAGE_TN : (WM == 2 ? NSTAGE_WM2 : NSTAGE_WM4))   // LDS ring: NSTAGE-1 k-steps in flight + 1 being read
#ifndef OCC_WM2
#define OCC_WM2 4
#endif
#define A_TILE_BYTES (BM * BK * 2)
#define B_TILE_BYTES (BN * BK * 2)
#define STAGE_BYTES (A_TILE_BYTES + B_TILE_BYTES)

__device__ __forceinline__ int tr_sigma(int kk) { return (kk & 3) | (((kk >> 3) & 1) << 2); }

// ---- global -> LDS, direct (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per wave-instruction) ----
// The LDS destination of one instruction is wave-uniform base + lane*16, so the tile images are
// lane-linear per 1-KiB piece and the XOR swizzle is applied to the per-lane SOURCE address
// (cdna_hip_programming.md rule 21).  A tile is 16 pieces; wave w issues pieces w, w+4, w+8, w+12.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// row-form chunk swizzle: {0,2,3,1}[(r>>2)&3]
__device__ __forceinline__ int row_f(int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; }

// PIECES 1-KiB pieces per tile (8 per 128 rows/cols), dealt round-robin to the NW waves.
// Addressing is split into a wave-UNIFORM base (tile origin at the current k, kept in SGPRs and advanced by one
// scalar add per k-step) and a per-lane 32-bit byte offset per piece that never changes: in the steady state a
// piece costs its DMA instruction and nothing else.  Lanes whose row / column lies beyond the matrix are CLAMPED
// onto its last row / 8-column group: what they fetch only reaches accumulator rows / columns the epilogue never
// stores.  Only a k tail (k_end not a multiple of the stage depth: patch-embed K = 48, ragged wgrad splits) needs
// real zeros; that one step takes the per-lane select path (`issue_tail`, 16 zero bytes from `zeros`).
template <bool TR, int PIECES, int NW>
struct Stager {
  static constexpr int PER = PIECES / NW;
  const char* base;       // uniform: &X[k_cur * kstride] as bytes (row form: + k_cur elements; tr form: + k_cur rows)
  long long step_bytes;   // uniform: bytes per k-step
  unsigned voff[PER];     // per lane: byte offset of its 16-B chunk of piece i from `base`
  int k_cur;              // uniform: k of the next stage to issue

  // k offset (elements) of this lane's chunk inside a stage, piece j
  static __device__ __forceinline__ int lane_k(int j, int lane) {
    if (!TR) {
      constexpr int LPR = 4;
      const int rl = (64 / LPR) * j + lane / LPR;
      return ((lane & (LPR - 1)) ^ row_f(rl)) << 3;
    }
    return 4 * (j % 8) + (lane >> 4);
  }
  __device__ __forceinline__ void setup(const bf16* __restrict__ X, int ld, int idx0, int idx_end, int k_begin, int wid, int lane) {
    base = (const char*)(X + (size_t)k_begin * (TR ? ld : 1));
    step_bytes = (long long)BK * (TR ? ld : 1) * 2;
    k_cur = k_begin;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int j = wid + NW * i;
      if (!TR) {  // piece = 16 rows x 64 B
        constexpr int LPR = 4;  // lanes (16-B chunks) per row
        const int rl = (64 / LPR) * j + lane / LPR;
        const int row = min(idx0 + rl, idx_end - 1);
        voff[i] = ((unsigned)row * (unsigned)ld + (unsigned)lane_k(j, lane)) * 2u;
      } else {    // piece = 4 k-rows x 256 B of one 128-column sub-tile (8 pieces per sub-tile)
        const int sub = j / 8;
        const int kk = lane_k(j, lane);
        const int c = (lane & 15) ^ (2 * tr_sigma(kk));
        const int col = min(idx0 + 128 * sub + (c << 3), idx_end - 8);
        voff[i] = ((unsigned)kk * (unsigned)ld + (unsigned)col) * 2u;
      }
    }
  }
  // full stage (k_cur + BK <= k_end)
  __device__ __forceinline__ void issue(unsigned char* tile, int wid) {
#pragma unroll
    for (int i = 0; i < PER; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + voff[i]), (lptr_t)(tile + 1024 * (wid + NW * i)), 16, 0, 0);
    base += step_bytes;
    k_cur += BK;
  }
  // last, partial stage: chunks at k >= k_end come from the zero page
  __device__ __forceinline__ void issue_tail(unsigned char* tile, int k_end, const void* zeros, int wid, int lane) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int j = wid + NW * i;
      const void* p = (k_cur + lane_k(j, lane) < k_end) ? (const void*)(base + voff[i]) : zeros;
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(tile + 1024 * j), 16, 0, 0);
    }
    base += step_bytes;
    k_cur += BK;
  }
};

// ---- LDS -> MFMA fragments, by inline asm ------------------------------------------------------------------
// The compiler cannot tell an LDS-DMA's destination stage from the stage being read and puts `s_waitcnt vmcnt(0)`
// in front of every ds_read that follows a global_load_lds - which drains the whole prefetch ring at every k-step
// (a deeper ring then buys nothing).  Reads issued from inline asm carry no such dependence; the counted
// `s_waitcnt vmcnt(N)` + `s_barrier` at the top of the step is the only synchronisation, as intended.
// One asm block per step: all ds_reads of both operands, then s_waitcnt lgkmcnt(0).
// loop-invariant LDS byte offsets (inside a stage) of a wave's fragments
template <bool TR>
struct FragAddr {
  unsigned a[TR ? 4 : 1];
  __device__ __forceinline__ void setup(int tile_off, int w0, int lane) {
    const int g = lane >> 4, li = lane & 15;
    if (!TR) {
      a[0] = tile_off + (w0 + li) * 64 + ((g ^ row_f(w0 + li)) << 4);   // fragment i at + 1024 * i
    } else {
      const int q = li >> 2, p = li & 3, sw = 2 * (q | ((g & 1) << 2)), kk0 = 8 * g + q;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int i0 = w0 + 16 * i, c8 = ((i0 & 127) >> 2) + p;
        a[i] = tile_off + (i0 >> 7) * 8192 + kk0 * 256 + ((((c8 >> 1) ^ sw)) << 4) + ((c8 & 1) << 3);  // hi half at + 1024
      }
    }
  }
};
__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 as_frag(u32x2 lo, u32x2 hi) { return __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]}); }

#define VG_RD128_4(o0, o1, o2, o3, ad) \
  "ds_read_b128 %" #o0 ", %" #ad "\n\tds_read_b128 %" #o1 ", %" #ad " offset:1024\n\t" \
  "ds_read_b128 %" #o2 ", %" #ad " offset:2048\n\tds_read_b128 %" #o3 ", %" #ad " offset:3072\n\t"
#define VG_RDTR_2(o0, o1, ad) "ds_read_b64_tr_b16 %" #o0 ", %" #ad "\n\tds_read_b64_tr_b16 %" #o1 ", %" #ad " offset:1024\n\t"

template <bool A_TR, bool B_TR>
__device__ __forceinline__ void load_frags_asm(unsigned sb, const FragAddr<A_TR>& fa, const FragAddr<B_TR>& fb, bf16x8 (&fm)[4], bf16x8 (&fn)[4]) {
  if constexpr (!A_TR && !B_TR) {
    u32x4 m0, m1, m2, m3, n0, n1, n2, n3;
    asm volatile(VG_RD128_4(0, 1, 2, 3, 8) VG_RD128_4(4, 5, 6, 7, 9) "s_waitcnt lgkmcnt(0)"
                 : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)
                 : "v"(sb + fa.a[0]), "v"(sb + fb.a[0]) : "memory");
    fm[0] = as_frag(m0); fm[1] = as_frag(m1); fm[2] = as_frag(m2); fm[3] = as_frag(m3);
    fn[0] = as_frag(n0); fn[1] = as_frag(n1); fn[2] = as_frag(n2); fn[3] = as_frag(n3);
  } else if constexpr (!A_TR && B_TR) {
    u32x4 m0, m1, m2, m3;
    u32x2 l0, h0, l1, h1, l2, h2, l3, h3;
    asm volatile(VG_RD128_4(0, 1, 2, 3, 12) VG_RDTR_2(4, 5, 13) VG_RDTR_2(6, 7, 14) VG_RDTR_2(8, 9, 15) VG_RDTR_2(10, 11, 16) "s_waitcnt lgkmcnt(0)"
                 : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3), "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
                 : "v"(sb + fa.a[0]), "v"(sb + fb.a[0]), "v"(sb + fb.a[1]), "v"(sb + fb.a[2]), "v"(sb + fb.a[3]) : "memory");
    fm[0] = as_frag(m0); fm[1] = as_frag(m1); fm[2] = as_frag(m2); fm[3] = as_frag(m3);
    fn[0] = as_frag(l0, h0); fn[1] = as_frag(l1, h1); fn[2] = as_frag(l2, h2); fn[3] = as_frag(l3, h3);
  } else {
    static_assert(A_TR && B_TR, "operand forms: NT, NN, TN");
    u32x2 al0, ah0, al1, ah1, al2, ah2, al3, ah3, l0, h0, l1, h1, l2, h2, l3, h3;
    asm volatile(VG_RDTR_2(0, 1, 16) VG_RDTR_2(2, 3, 17) VG_RDTR_2(4, 5, 18) VG_RDTR_2(6, 7, 19)
                 VG_RDTR_2(8, 9, 20) VG_RDTR_2(10, 11, 21) VG_RDTR_2(12, 13, 22) VG_RDTR_2(14, 15, 23) "s_waitcnt lgkmcnt(0)"
                 : "=&v"(al0), "=&v"(ah0), "=&v"(al1), "=&v"(ah1), "=&v"(al2), "=&v"(ah2), "=&v"(al3), "=&v"(ah3),
                   "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
                 : "v"(sb + fa.a[0]), "v"(sb + fa.a[1]), "v"(sb + fa.a[2]), "v"(sb + fa.a[3]),
                   "v"(sb + fb.a[0]), "v"(sb + fb.a[1]), "v"(sb + fb.a[2]), "v"(sb + fb.a[3]) : "memory");
    fm[0] = as_frag(al0, ah0); fm[1] = as_frag(al1, ah1); fm[2] = as_frag(al2, ah2); fm[3] = as_frag(al3, ah3);
    fn[0] = as_frag(l0, h0); fn[1] = as_frag(l1, h1); fn[2] = as_frag(l2, h2); fn[3] = as_frag(l3, h3);
  }
}

template <int ACT>
__device__ __forceinline__ float apply_act(float scale, float v) {
  if (ACT == VG_ACT_GELU) return vg_gelu(v);
  if (ACT == VG_ACT_SIN) return __sinf(scale * v);
  if (ACT == VG_ACT_TANH) return vg_tanh(v);
  return v;
}

// FEAT: epilogue features COMPILED IN (each still tests its runtime pointer); the launcher picks the
// smallest compiled superset so the common epilogues carry no dead address arithmetic.
enum { F_RES = 1, F_RESF = 2, F_REMAP = 4, F_C2 = 8, F_PREF32 = 16, F_ALL = 31, F_DROP = 32 };

// Occupancy is the lever on MI355X for these short-K GEMMs (measured: 16 waves/CU beats a deeper DMA ring at
// 8-12 waves/CU by 25-40 %): WM=2 -> 2-stage ring (32 KiB) x 4 workgroups/CU, WM=4 -> 3-stage ring (72 KiB) x 2
// workgroups/CU; both 4 waves/SIMD, so at most 128 registers per lane.
// Waves per SIMD the kernel is compiled for: 4 (<= 128 registers) for the hot instantiations, 3 for weight gradients
// (LDS allows 3 workgroups/CU anyway), 2 (<= 256 registers) for the feature-laden epilogues of the rarely launched
// embedding / generator-entry / SIREN GEMMs, which otherwise spill 50-120 registers to scratch.
template <int MODE, int WM, int ACT, int FEAT>
constexpr int vg_gemm_waves() {
  if (MODE == VG_TN) return 3;
  if ((FEAT & (F_RESF | F_REMAP | F_PREF32)) || ACT == VG_ACT_MUL_COS) return 2;
  return WM == 2 ? OCC_WM2 : OCC_WM4;
}
// A workgroup processes grp.tpw CONSECUTIVE tiles (same m-panel first: the A panel stays in its XCD's L2).  Between two
// tiles the DMA of the next tile's first NSTAGE-1 stages is issued BEFORE the current tile's epilogue, so the epilogue's
// loads and 16-byte stores run under that latency and the next main loop starts on landed data.  (The launcher picks
// tpw = 1 for this model's shapes - measurements in vg_gemm_launch - so the loop below normally runs once.)
// vmcnt protocol at the seam: one `s_waitcnt vmcnt(0)` after the
// epilogue's LOADS and before its first STORE retires the prefetched stages (and those loads); the stores are never
// waited for by name - the first counted wait that covers them is NSTAGE-1 k-steps into the next tile.
template <int MODE, int WM, int ACT, int FEAT>
__global__ __launch_bounds__(128 * WM, (vg_gemm_waves<MODE, WM, ACT, FEAT>())) void vg_gemm_kernel(const VgGemmGroup grp) {
  constexpr int NW = 2 * WM;
  constexpr bool A_TR = (MODE == VG_TN);
  constexpr bool B_TR = (MODE != VG_NT);
  constexpr int SMEM_BYTES = NSTAGE * STAGE_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

  // XCD-aware block order: blocks b, b+8, ... share an XCD (L2); give each XCD a contiguous
  // run of tiles so the n-tiles of one m-panel hit the same L2 (bijective for any grid size).
  int wg;
  {
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
    const int q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int bid_first = wg * grp.tpw;
  const int bid_end = min(bid_first + grp.tpw, grp.total);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index as a scalar: per-wave LDS bases stay in SGPRs
  const int wm = wid >> 1, wn = wid & 1;  // wm in [0, WM)
  const void* zeros = grp.zeros;
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;

  // tile -> (problem, origin, K slice)
  struct Tile { int pi, m0, n0, tn, split, k_begin, k_end; };
  auto locate = [&](int bid) {
    Tile t;
    t.pi = 0;
#pragma unroll
    for (int i = 1; i < VG_MAX_GROUP; ++i)
      if (i < grp.n && bid >= grp.p[i].tile_start) t.pi = i;
    const VgGemmProb& Q = grp.p[t.pi];
    const int local = bid - Q.tile_start;
    const int tiles_mn = Q.tiles_m * Q.tiles_n;
    t.split = local / tiles_mn;
    const int r = local - t.split * tiles_mn;
    const int tm = r / Q.tiles_n;
    t.tn = r - tm * Q.tiles_n;
    t.m0 = tm * BM; t.n0 = t.tn * BN;
    t.k_begin = t.split * Q.k_per_split;
    t.k_end = min(Q.K, t.k_begin + Q.k_per_split);
    return t;
  };
  Stager<A_TR, 4 * WM, NW> sa;
  Stager<B_TR, 8, NW> sb;
  // stages are issued strictly in order (prologue, then one per k-step), so the stagers keep a running k
#define ISSUE(step, kend)                                                                                \
  do {                                                                                                   \
    unsigned char* _b = smem + ((step) % NSTAGE) * STAGE_BYTES;                                          \
    if (sa.k_cur + BK <= (kend)) {                                                                      \
      sa.issue(_b, wid);                                                                                 \
      sb.issue(_b + A_TILE_BYTES, wid);                                                                  \
    } else {                                                                                             \
      sa.issue_tail(_b, (kend), zeros, wid, lane);                                                       \
      sb.issue_tail(_b + A_TILE_BYTES, (kend), zeros, wid, lane);                                        \
    }                                                                                                    \
  } while (0)
  // row form indexes rows (m or n) against M/N; tr form indexes columns against M/N.
  auto prime = [&](const Tile& t) {  // stagers at the tile's origin + the DMA of its first NSTAGE-1 stages
    const VgGemmProb& Q = grp.p[t.pi];
    // lane-derived address parts are recomputed per tile from an opaque copy of the lane id: hoisted out of the tile loop
    // they would be spilled around it, and a spill reload drags a compiler `s_waitcnt vmcnt(0)` into the seam
    int ln = lane;
    asm volatile("" : "+v"(ln));
    sa.setup(Q.A, Q.lda, t.m0, Q.M, t.k_begin, wid, ln);
    sb.setup(Q.B, Q.ldb, t.n0, Q.N, t.k_begin, wid, ln);
    const int n = (t.k_end - t.k_begin + BK - 1) / BK;
    for (int s = 0; s < NSTAGE - 1 && s < n; ++s) ISSUE(s, t.k_end);
  };
  const unsigned smem_base = (unsigned)(unsigned long)(lptr_t)smem;
  FragAddr<A_TR> fra;
  FragAddr<B_TR> frb;
  fra.setup(0, wm * 64, lane);
  frb.setup(A_TILE_BYTES, wn * 64, lane);
  constexpr int DPS = (4 * WM + 8) / NW;  // LDS-DMA instructions per wave per stage: 4 (WM=2) or 3 (WM=4)

  Tile cur = locate(bid_first);
  prime(cur);
  bool landed = false;  // the prologue stages of `cur` were retired by the previous tile's epilogue wait
  for (int bid = bid_first; bid < bid_end; ++bid) {
  const VgGemmProb& P = grp.p[cur.pi];
  // epilogue operands, read from kernarg memory up front (overlaps the DMA latency)
  const int eM = P.M, eN = P.N;
  bf16* const eC = P.C; const int eldc = P.ldc;
  bf16* const eC2 = P.C2; const int eldc2 = P.ldc2; const int ec2g = P.c2_gelu_grad;
  float* const eCf = P.Cf; const int eldcf = P.ldcf; const long long ecfs = P.cf_split_stride;
  const float* const ebias = P.bias;
  const bf16* const eres = P.res; const int eldr = P.ldr;
  const float* const eresf = P.resf; const int eper = P.res_period;
  const bf16* const eZ = P.Z; const int eldz = P.ldz;
  const float* const eZf = P.Zf; const int eldzf = P.ldzf;
  const float ascale = P.act_scale;
  const int epre = P.pre_f32, rip = P.row_in_per, rop = P.row_out_per, roo = P.row_out_off;
  const unsigned dthr = P.drop_thresh, dkey = vg_drop_key(P.drop_key, P.drop_step); const float dscale = P.drop_scale; const int dpost = P.drop_post;
  const int drm = P.drop_row_mul > 1 ? P.drop_row_mul : 1;
  const int m0 = cur.m0, n0 = cur.n0, split = cur.split, k_end = cur.k_end;
  const int nsteps = (k_end - cur.k_begin + BK - 1) / BK;

  // TN: bias-gradient column sums ride along (first n-tile, wn == 0 waves)
  const bool do_cs = (MODE == VG_TN) && P.colsum != nullptr && cur.tn == 0 && wn == 0;
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 acc[4][4];  // [nt][mt]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Per k-step: counted vmcnt (stage s landed, later stages may still fly) -> s_barrier -> DMA for stage s+NSTAGE-1
  // -> ds_read fragments -> 16 MFMAs.  The co-resident workgroups' waves fill the SIMD while this one waits.
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    const int ahead = nsteps - 1 - s;  // stages issued after s (at most NSTAGE-2 of them are in flight here)
    if (landed && s < NSTAGE - 1) asm volatile("s_barrier" ::: "memory");  // retired at the seam; epilogue stores may still fly
    else if (NSTAGE >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * DPS) : "memory");
    else if (NSTAGE >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (s + NSTAGE - 1 < nsteps) ISSUE(s + NSTAGE - 1, k_end);
    bf16x8 fm[4], fn[4];
    load_frags_asm<A_TR, B_TR>(smem_base + (s % NSTAGE) * STAGE_BYTES, fra, frb, fm, fn);
    {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = vg_mfma(fn[nt], fm[mt], acc[nt][mt]);
      if (MODE == VG_TN && do_cs) {  // rows of the result are all equal: sum_k A[k, m] lands on lane li = m, any register
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) accb[mt] = vg_mfma(ones, fm[mt], accb[mt]);
      }
    }
  }
  // ---- seam: next tile's first stages go out before this tile's epilogue ----------------------
  const bool has_next = bid + 1 < bid_end;
  Tile nxt = cur;
  if (has_next) {
    nxt = locate(bid + 1);
    asm volatile("s_barrier" ::: "memory");  // every wave has finished its fragment reads of the last step (WAR on the ring)
    prime(nxt);
  }

  // ---- epilogue ---------------------------------------------------------------------------
  // Problem fields were copied to registers up front (a reference into kernarg memory is re-read after every
  // store), and every global LOAD of the epilogue is issued before the first STORE: vmcnt retires in order, so a
  // load issued behind stores would wait for them.
  // lane coordinates from an opaque copy of the lane id, per tile (see prime(): nothing lane-derived is carried - and
  // spilled - across the tile loop except the fragment addresses the main loop itself needs)
  int lne = lane;
  asm volatile("" : "+v"(lne));
  const int g = lne >> 4, li = lne & 15;
  if (MODE == VG_TN && do_cs && g == 0) {
    float* cs = P.colsum + (size_t)split * P.colsum_split_stride;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wm * 64 + mt * 16 + li;
      if (m < P.M) cs[m] = accb[mt][0];
    }
  }
  // ---- epilogue: registers only ---------------------------------------------------------------
  // A lane holds, per (n-tile, m-tile), 4 consecutive n of row li.  v_permlane16_swap between the even and the
  // odd n-tile of a pair hands every lane 8 CONSECUTIVE n of one tile (even lane-groups keep the even tile,
  // odd lane-groups the odd one), so each lane loads/stores 16 B (bf16) or 32 B (fp32) per slot with no trip
  // through LDS (the LDS transpose of the first version cost ~1.5k LDS cycles per workgroup and stalled the
  // co-resident workgroup's main loop).  8 slots per lane: q = 2*mt + pair.
  constexpr bool NEED_ZBF = (ACT == VG_ACT_MUL_GELU_GRAD || ACT == VG_ACT_MUL_TANH_GRAD || ACT == VG_ACT_MUL_Z || ACT == VG_ACT_MUL_Z8);
  constexpr bool Z8 = (ACT == VG_ACT_MUL_Z8);
  constexpr bool NEED_ZF = (ACT == VG_ACT_MUL_COS);
  constexpr bool HAS_RES = (FEAT & F_RES) != 0, HAS_RESF = (FEAT & F_RESF) != 0, HAS_REMAP = (FEAT & F_REMAP) != 0;
  constexpr bool HAS_C2 = (FEAT & F_C2) != 0, HAS_PREF32 = (FEAT & F_PREF32) != 0, HAS_DROP = (FEAT & F_DROP) != 0;
  constexpr bool PRE_BF = NEED_ZBF || HAS_RES, PRE_F = NEED_ZF || HAS_RESF;
  const int mrow0 = m0 + wm * 64 + li;                               // slot (mt, .) covers row mrow0 + 16*mt
  const int ncol0 = n0 + wn * 64 + ((g & 1) << 4) + ((g & 2) << 2);  // slot (., pair) covers columns ncol0 + 32*pair .. +7
  auto out_row = [&](int m) -> int {
    if (HAS_REMAP && rip > 0) return (m / rip) * rop + roo + (m % rip);
    return m;
  };
  if (MODE != VG_TN && ebias) {
    // bias: added to the accumulators in their own layout (a lane's 4 consecutive n of tile nt) right away, so its
    // registers are dead again before the slot loop needs its temporaries
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + 16 * nt + 4 * g;
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if (n < eN) b4 = *(const f32x4*)(ebias + n);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[nt][mt] += b4;
    }
  }
  // The 8 slots are processed in groups of 4 (m-tiles 2h, 2h+1) or 2: the epilogue operands (residual, stored
  // derivative, fp32 addend) of a group are all loaded before its first store; a group's registers instead of a whole
  // tile's keep every instantiation inside its register budget (no scratch) now that the stagers and fragment addresses
  // stay live across the epilogue for the next tile.
  constexpr int NG = HAS_DROP ? 4 : 2, GS = 8 / NG;  // slot groups; the dropout epilogue (hash temporaries) takes quarters
#pragma unroll
  for (int hf = 0; hf < NG; ++hf) {
  bf16x8 pre_bf[PRE_BF ? GS : 1];
  f32x4 pre_f0[PRE_F ? GS : 1], pre_f1[PRE_F ? GS : 1];
  if (MODE != VG_TN && (PRE_BF || PRE_F)) {
#pragma unroll
    for (int qq = 0; qq < GS; ++qq) {
      const int q = GS * hf + qq;
      const int m = mrow0 + 16 * (q >> 1), n = ncol0 + 32 * (q & 1);
      const bool ok = n < eN && m < eM;
      if (PRE_BF) {
        bf16x8 zb = {0, 0, 0, 0, 0, 0, 0, 0};
        if (Z8) {  // 8 bytes per lane: the codes of its 8 columns, carried in the first half of the slot
          u32x2 c8 = {0u, 0u};
          if (ok) c8 = *(const u32x2*)((const unsigned char*)eZ + (unsigned)(m * eldz + n));
          union { u32x4 u; bf16x8 b; } cv; cv.u = (u32x4){c8[0], c8[1], 0u, 0u};
          zb = cv.b;
        } else if (NEED_ZBF) { if (ok) zb = *(const bf16x8*)(eZ + (unsigned)(m * eldz + n)); }
        else if (eres && ok) zb = *(const bf16x8*)(eres + (unsigned)(out_row(m) * eldr + n));
        pre_bf[qq] = zb;
      }
      if (PRE_F) {
        f32x4 zf0 = {0.f, 0.f, 0.f, 0.f}, zf1 = zf0;
        if (NEED_ZF) { if (ok) { const float* zp = eZf + (unsigned)(m * eldzf + n); zf0 = *(const f32x4*)zp; zf1 = *(const f32x4*)(zp + 4); } }
        else if (eresf && ok) { const float* rp = eresf + (unsigned)((m % eper) * eN + n); zf0 = *(const f32x4*)rp; zf1 = *(const f32x4*)(rp + 4); }
        pre_f0[qq] = zf0; pre_f1[qq] = zf1;
      }
    }
  }
  // seam wait: the next tile's prefetched stages (and the loads above) - everything issued so far - have landed; the
  // stores below are not waited for here
  if (hf == 0 && has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int qq = 0; qq < GS; ++qq) {
    {
      const int q = GS * hf + qq;
      const int mt = q >> 1, pr = q & 1;
      const int m = mrow0 + 16 * mt, n = ncol0 + 32 * pr;
      const f32x4 te = acc[2 * pr][mt], to = acc[2 * pr + 1][mt];
      f32x4 lo, hi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // rows 1,3 of the even tile's register <-> rows 0,2 of the odd tile's
        const unsigned ue = __float_as_uint(te[r]), uo = __float_as_uint(to[r]);
        const auto sw = __builtin_amdgcn_permlane16_swap(ue, uo, false, false);
        lo[r] = __uint_as_float(sw[0]);
        hi[r] = __uint_as_float(sw[1]);
      }
      const bool ncol_ok = n < eN;
      if (m >= eM || !ncol_ok) continue;
      if (MODE == VG_TN) {
        float* dst = eCf + (size_t)split * ecfs + (unsigned)(m * eldcf + n);
        if (P.cf_accumulate) { lo += *(const f32x4*)dst; hi += *(const f32x4*)(dst + 4); }
        *(f32x4*)dst = lo;
        *(f32x4*)(dst + 4) = hi;
        continue;
      }
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      const int mo = out_row(m);
      if (HAS_PREF32 && epre) {
        float* dst = eCf + (unsigned)(mo * eldcf + n);
        *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
        *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
      }
      float gact[8];  // GELU: activation and derivative share one exp / rcp / polynomial
      if (ACT == VG_ACT_GELU) {
        float gd[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) vg_gelu_both(v[r], gact[r], gd[r]);
        if (HAS_C2 && eC2) {
          if (ec2g == 2) {
            const u32x2 c8 = {vg_g8_pack4(gd[0], gd[1], gd[2], gd[3]), vg_g8_pack4(gd[4], gd[5], gd[6], gd[7])};
            *(u32x2*)((unsigned char*)eC2 + (unsigned)(mo * eldc2 + n)) = c8;
          } else {
            bf16x8 o;
#pragma unroll
            for (int r = 0; r < 8; ++r) o[r] = vg_f2bf(ec2g ? gd[r] : v[r]);
            *(bf16x8*)(eC2 + (unsigned)(mo * eldc2 + n)) = o;
          }
        }
      } else if (HAS_C2 && eC2) {
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = vg_f2bf(v[r]);
        *(bf16x8*)(eC2 + (unsigned)(mo * eldc2 + n)) = o;
      }
      if (Z8) {
        union { bf16x8 b; u32x4 u; } cv; cv.b = pre_bf[PRE_BF ? qq : 0];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] *= vg_g8_value(cv.u[0], r); v[r + 4] *= vg_g8_value(cv.u[1], r); }
      } else if (NEED_ZBF) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float zv = vg_bf2f(pre_bf[PRE_BF ? qq : 0][r]);
          v[r] *= (ACT == VG_ACT_MUL_GELU_GRAD) ? vg_gelu_grad(zv) : ((ACT == VG_ACT_MUL_Z) ? zv : (1.f - zv * zv));
        }
      } else if (NEED_ZF) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] *= ascale * __cosf(ascale * pre_f0[PRE_F ? qq : 0][r]);
          v[r + 4] *= ascale * __cosf(ascale * pre_f1[PRE_F ? qq : 0][r]);
        }
      } else if (ACT == VG_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gact[r];
      } else if (ACT != VG_ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = apply_act<ACT>(ascale, v[r]);
      }
      unsigned dw0 = 0, dw1 = 0;
      if (HAS_DROP && dthr) {
        const unsigned i4 = (unsigned)(mo * drm * eN + n) >> 2;
        dw0 = vg_drop_word(dkey, i4); dw1 = vg_drop_word(dkey, i4 + 1);
        if (!dpost) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] *= vg_drop_factor(dw0, r, dthr, dscale); v[r + 4] *= vg_drop_factor(dw1, r, dthr, dscale); }
        }
      }
      if (HAS_RES && !NEED_ZBF && eres) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += vg_bf2f(pre_bf[PRE_BF ? qq : 0][r]);
      }
      if (HAS_RESF && !NEED_ZF && eresf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += pre_f0[PRE_F ? qq : 0][r]; v[r + 4] += pre_f1[PRE_F ? qq : 0][r]; }
      }
      if (HAS_DROP && dthr && dpost) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] *= vg_drop_factor(dw0, r, dthr, dscale); v[r + 4] *= vg_drop_factor(dw1, r, dthr, dscale); }
      }
      if (eC) {
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = vg_f2bf(v[r]);
        *(bf16x8*)(eC + (unsigned)(mo * eldc + n)) = o;
      }
    }
  }
  }  // slot groups
  cur = nxt;
  landed = has_next;
  }  // tiles of this workgroup
#undef ISSUE
}

__device__ __attribute__((aligned(16))) unsigned int vg_zero_page[4] = {0u, 0u, 0u, 0u};

int vg_gemm_launch(VgGemmProb* probs, int n, int mode, hipStream_t stream) {
  if (n < 1 || n > VG_MAX_GROUP) return -1;
#if !defined(VG_TN384_OFF)  // A/B builds (make var DEFS=-DVG_TN384_OFF): weight gradients on the tiled kernel
  if (mode == VG_TN) {  // weight gradients whose n extent is a multiple of 384: 128 x 384 tiles (gemm_tn.hip)
    const int r = vg_gemm_tn384_try(probs, n, stream);
    if (r > 0) return 0;
    if (r < 0) return -r;
  }
#endif
  if (n == 1) {  // the K = 384 Linears go to the weights-in-registers kernel (gemm_wr.hip) when it covers them
#if defined(VG_WR_OFF)  // A/B builds (make var DEFS=-DVG_WR_OFF): everything on the tiled kernel
    constexpr int wr_on = 0;
#elif defined(VG_TUNING)
    static const int wr_on = getenv("VG_GEMM_WR") ? atoi(getenv("VG_GEMM_WR")) : 1;
#else
    constexpr int wr_on = 1;
#endif
    if (wr_on) {
      const int r = vg_gemm_wr_try(probs[0], mode, stream);
      if (r > 0) return 0;
      if (r < 0) return -r;
    }
  }
  // tile height: 256 rows (8 waves) when every problem is tall enough to fill the chip that way
  // 256-row tiles (8 waves, 2 workgroups/CU) only when they still give every CU its two workgroups (>= 512 tiles)
  // and the epilogue is light; otherwise 128-row tiles (4 workgroups/CU): small problems (generator, M = 8192)
  // and transcendental epilogues (GELU, gelu', sin, cos) need the extra workgroups to fill / overlap.
  int wm4 = 1;
  for (int i = 0; i < n; ++i) {
    const long long t4 = (long long)((probs[i].M + 255) / 256) * ((probs[i].N + 127) / 128);
#ifdef VG_TUNING  // experimental builds only (make var): the product library reads no environment
    static const long long t4min = getenv("VG_GEMM_T4MIN") ? atoll(getenv("VG_GEMM_T4MIN")) : 96;
#else
    constexpr long long t4min = 96;
#endif
    // epilogues that fit the 128-register budget of 8-wave workgroups (sin / cos / tanh variants do not)
    const bool light = probs[i].act == VG_ACT_NONE || probs[i].act == VG_ACT_MUL_Z || probs[i].act == VG_ACT_MUL_Z8 || probs[i].act == VG_ACT_GELU;
    if (t4 < t4min || mode == VG_TN || !light) wm4 = 0;  // (128-row tiles for the generator's mapping Linear - M = 256, 96 -> 192 workgroups - measured 23.5 vs 24.3 us: not kept)
  }
#ifdef VG_TUNING
  static const int wm_env = getenv("VG_GEMM_WM") ? atoi(getenv("VG_GEMM_WM")) : 0;  // force the tile height
  if (wm_env) wm4 = (wm_env == 4) && mode != VG_TN;
#endif
  const int bm = wm4 ? 256 : 128;
  VgGemmGroup grp;
  grp.n = n;
  {
    static void* zp = nullptr;  // one device per process (one process per GPU)
    if (!zp && hipGetSymbolAddress(&zp, HIP_SYMBOL(vg_zero_page)) != hipSuccess) return -5;
    grp.zeros = zp;
  }
  int total = 0;
  for (int i = 0; i < n; ++i) {
    VgGemmProb& p = probs[i];
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -2;
    // 16-byte vector loads: contiguous extents and leading dims must be multiples of 8 elements
    if ((p.lda & 7) || (p.ldb & 7) || (p.N & 7)) return -3;
    // epilogue addresses are base + 32-bit element offsets
    if ((long long)(p.M + 256) * (long long)(p.N > p.ldc ? p.N : p.ldc) >= (1LL << 31)) return -3;
    if (mode == VG_NT && (p.K & 7)) return -3;
    if (mode == VG_NN && (p.K & 7)) return -3;
    if (mode == VG_TN && (p.M & 7)) return -3;
    // one-byte derivative codes: 8-byte accesses per lane
    if (p.c2_gelu_grad == 2 && p.C2 && (p.act != VG_ACT_GELU || (p.ldc2 & 7))) return -3;
    if (p.act == VG_ACT_MUL_Z8 && (!p.Z || (p.ldz & 7))) return -3;
    p.tiles_m = (p.M + bm - 1) / bm;
    p.tiles_n = (p.N + BN - 1) / BN;
    int splits = (mode == VG_TN) ? (p.splits > 0 ? p.splits : 1) : 1;
    int ksteps = (p.K + BK - 1) / BK;
    int per = (ksteps + splits - 1) / splits;
    p.k_per_split = per * BK;
    splits = (ksteps + per - 1) / per;  // drop empty slices
    p.splits = splits;
    if (p.cf_accumulate && (mode != VG_TN || splits != 1)) return -4;  // accumulation needs a single writer per element
    p.tile_start = total;
    total += p.tiles_m * p.tiles_n * splits;
    grp.p[i] = p;
  }
  // Tiles per workgroup (see the kernel's header comment).  Measured on the C2 step (round 2, one MI355X): one tile per
  // workgroup 6.80 ms/step; two tiles wherever the launch has >= 2 tiles per CU slot (the QKV forward) 6.86; two / three /
  // four everywhere 8.3 / 10.2 / 11.8 - the hardware's own refill of CU slots already staggers the workgroups of a
  // multi-round launch, and in the single-round launches of this model (390-780 tiles) fewer, longer workgroups only cost
  // occupancy.  (Also measured and dropped: delaying the second workgroup of every CU by 1-6 us at the start so that the two
  // are in different phases - 6.83 -> 6.83 / 6.87 / 6.91 / 6.98 ms with the delay - and running the step as two concurrent
  // half-batch chains, engine.py two_stream: +2 %.)  The product therefore runs one tile per workgroup; the seam is exercised by the tests through the tuning
  // build (VG_GEMM_TPW) and is the hook for shapes with many rounds of tiles.
  int tpw = 1;
#ifdef VG_TUNING
  {
    static const int tpw_env = getenv("VG_GEMM_TPW") ? atoi(getenv("VG_GEMM_TPW")) : 0;
    if (tpw_env > 0) tpw = tpw_env;
  }
#endif
  grp.tpw = tpw;

  grp.total = total;
  dim3 grid((total + tpw - 1) / tpw);
  const int act = probs[0].act;
  for (int i = 1; i < n; ++i)
    if (probs[i].act != act) return -4;  // one epilogue per launch
#define VG_LAUNCH(MODE_, WM_, ACT_, FEAT_) \
  hipLaunchKernelGGL((vg_gemm_kernel<MODE_, WM_, ACT_, FEAT_>), grid, dim3(128 * WM_), 0, stream, grp)
#define VG_BY_WM(MODE_, ACT_, FEAT_) do { if (wm4) VG_LAUNCH(MODE_, 4, ACT_, FEAT_); else VG_LAUNCH(MODE_, 2, ACT_, FEAT_); } while (0)
  // features any problem of the group needs
  int feat = 0;
  for (int i = 0; i < n; ++i) {
    const VgGemmProb& q = probs[i];
    if (q.res) feat |= F_RES;
    if (q.resf) feat |= F_RESF;
    if (q.row_in_per > 0) feat |= F_REMAP;
    if (q.C2) feat |= F_C2;
    if (q.pre_f32) feat |= F_PREF32;
    if (q.drop_thresh) feat |= F_DROP;
  }
  if ((feat & F_DROP) && ((feat & (F_C2 | F_PREF32)) || mode != VG_NT)) return -4;
  if (mode == VG_NT) {
    if (act == VG_ACT_NONE) {
      if (feat == 0) VG_BY_WM(VG_NT, VG_ACT_NONE, 0);
      else if (feat == F_RES) VG_BY_WM(VG_NT, VG_ACT_NONE, F_RES);
      else if (feat == (F_DROP | F_RES)) VG_BY_WM(VG_NT, VG_ACT_NONE, F_DROP | F_RES);        // x + drop(Linear(.)): block sites
      else if (feat & F_DROP) VG_BY_WM(VG_NT, VG_ACT_NONE, F_DROP | F_RES | F_RESF | F_REMAP);  // embedding / generator block 0
      else if ((feat & ~(F_RESF | F_REMAP)) == 0) VG_BY_WM(VG_NT, VG_ACT_NONE, F_RESF | F_REMAP);
      else VG_BY_WM(VG_NT, VG_ACT_NONE, F_ALL);
    } else if (feat & F_DROP) {
      return -4;  // dropout is only fused behind a plain Linear (the reference's three sites)
    } else if (act == VG_ACT_GELU) {
      if ((feat & ~F_C2) == 0) VG_BY_WM(VG_NT, VG_ACT_GELU, F_C2);
      else VG_BY_WM(VG_NT, VG_ACT_GELU, F_ALL);
    } else if (act == VG_ACT_SIN) {
      if ((feat & ~F_PREF32) == 0) VG_BY_WM(VG_NT, VG_ACT_SIN, F_PREF32);
      else VG_BY_WM(VG_NT, VG_ACT_SIN, F_ALL);
    } else if (act == VG_ACT_TANH) {
      VG_BY_WM(VG_NT, VG_ACT_TANH, F_ALL);
    } else {
      return -4;
    }
  } else if (mode == VG_NN) {
    if (feat != 0) return -4;  // dgrad epilogues take no residual / second output / dropout
    switch (act) {
      case VG_ACT_NONE: VG_BY_WM(VG_NN, VG_ACT_NONE, 0); break;
      case VG_ACT_MUL_GELU_GRAD: VG_BY_WM(VG_NN, VG_ACT_MUL_GELU_GRAD, 0); break;
      case VG_ACT_MUL_COS: VG_BY_WM(VG_NN, VG_ACT_MUL_COS, 0); break;
      case VG_ACT_MUL_TANH_GRAD: VG_BY_WM(VG_NN, VG_ACT_MUL_TANH_GRAD, 0); break;
      case VG_ACT_MUL_Z: VG_BY_WM(VG_NN, VG_ACT_MUL_Z, 0); break;
      case VG_ACT_MUL_Z8: VG_BY_WM(VG_NN, VG_ACT_MUL_Z8, 0); break;
      default: return -4;
    }
  } else if (mode == VG_TN) {
    if (act != VG_ACT_NONE) return -4;
    VG_LAUNCH(VG_TN, 2, VG_ACT_NONE, 0);
  } else {
    return -4;
  }
#undef VG_BY_WM
#undef VG_LAUNCH
  return (int)hipGetLastError();
}
