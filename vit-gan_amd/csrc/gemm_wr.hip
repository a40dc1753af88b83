// "Weights in registers" GEMM for the Linears whose contraction is the embedding width (K = 384): QKV / out-proj / fc1
// forward (NT: C = A W^T, W [N,K]) and out-proj / fc2 input gradients (NN: C = A W, W [K,N]).  gfx950 only.
//
// Why a second kernel (measured, round 2): the tiled kernel of gemm.hip re-stages the weight tile with every
// k-step of every m-tile, and what it is short of is LDS-DMA issue slots (a 1-KiB piece costs its wave 60-180 cycles, a
// piece of 64-byte row segments twice a piece of full 128-byte lines), not HBM (aliasing A and C onto L2-resident rows
// changes its time by < 5 %).  Here a workgroup (4 waves) owns ONE 128-column n-tile for its whole life: wave w keeps
// W[n0 + 32 w .. + 32][0 .. 384) as 24 MFMA fragments in 96 VGPRs (loaded once, through LDS in full lines), and walks a run
// of 128-row m-tiles.  Only A passes through LDS, as [128 rows][64 k] stages of full 128-byte lines (8 rows x 128 B per
// wave-instruction), in a 4-slot ring with two stages in flight; one piece per 8 MFMAs instead of one per 5.3 half-line
// pieces.  Fragment reads are a rolling queue of 12 register quads, requested 11 MFMA-pairs ahead and waited for with a
// counted lgkmcnt; the step (wait, 2 MFMAs, next request) is one asm statement, so the schedule is the source order.
// Two workgroups per CU (256 registers per lane each): one's epilogue runs under the other's MFMAs.
//
// Rows are dealt to the workgroups in units of 32 (a run = full 128-row tiles + one shorter tile), the n-tiles of one
// run sit in one XCD so the A rows are fetched from HBM once.
#include "vg_gemm.h"
#include <type_traits>

namespace {
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int WR_KS = 6;          // 64-deep stages per tile (K = 384)
constexpr int WR_STAGE = 16384;   // 128 rows x 128 B
constexpr int WR_NSLOT = 4;
constexpr int WR_NS = 12;         // fragment queue slots

enum { WF_RES = 1, WF_C2 = 2, WF_DROP = 4 };

struct VgWrArgs {
  VgGemmProb p;
  int units;    // 32-row units of A
  int n_tiles;  // 128-column tiles
  int gpx;      // runs (row groups) per XCD
};

__device__ __forceinline__ bf16x8 wr_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 wr_frag(u32x2 lo, u32x2 hi) { return __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]}); }
__device__ __forceinline__ int wr_sigma(int kk) { return (kk & 3) | (((kk >> 3) & 1) << 2); }

}  // namespace

template <int WTR, int ACT, int FEAT>
__global__ __launch_bounds__(256, 2) void vg_gemm_wr_kernel(const VgWrArgs args) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[WR_NSLOT * WR_STAGE];
  const VgGemmProb& P = args.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (run, n-tile): the n-tiles of one run share an XCD (blocks b, b + 8, ... share an L2)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  if (idx >= args.gpx * args.n_tiles) return;
  const int run = xcd * args.gpx + idx / args.n_tiles, ntile = idx % args.n_tiles, nruns = 8 * args.gpx;
  const int u0 = (int)((long long)run * args.units / nruns), u1 = (int)((long long)(run + 1) * args.units / nruns);
  if (u0 >= u1) return;
  const int n0 = ntile * 128;
  const unsigned sbase = (unsigned)(unsigned long)(lptr_t)smem;
  const int eM = P.M, eN = P.N;

  // fragment read address inside a stage: row li (+16 per m-tile by immediate), 16-B chunk (g + 4 sub) ^ f(li)
  unsigned fa0;
  {
    const int g = lane >> 4, li = lane & 15, f = (li >> 1) & 7;
    fa0 = (unsigned)(li * 128 + (((g ^ (f & 3)) | (((f >> 2) & 1) << 2)) << 4));  // sub = 1: ^ 64
  }
  // the uniform base of a DMA is made opaque: folded with the lane offset into 64-bit per-lane addresses outside the
  // loops it costs two registers per piece
  auto issue4 = [&](const char* base, const unsigned (&voff)[4], int slot) {
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + voff[i]), (lptr_t)(smem + slot * WR_STAGE + 1024 * (wid + 4 * i)), 16, 0, 0);
  };

  // ---- A addressing (its first two stages are issued together with the second round of the W load below) ----------
  // per-lane DMA source offsets (bytes) of this wave's 4 pieces of a stage: piece p = wid + 4 i covers rows 8p .. 8p+7
  unsigned voffA[4];
  auto set_voff = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = wid + 4 * i, r = 8 * p + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      voffA[i] = ((unsigned)min(r, eM - 1 - m0) * (unsigned)P.lda + (unsigned)c * 8u) * 2u;
    }
  };
  const char* Ab = (const char*)P.A;
  const long long row_bytes = (long long)P.lda * 2;
  auto stage_src = [&](int m0, int ks) { return Ab + (long long)m0 * row_bytes + 128 * ks; };
  const int m_begin = u0 * 32, nfull = (u1 - u0) >> 2, rem = (u1 - u0) & 3;
  const int ntiles = nfull + (rem ? 1 : 0);
  set_voff(m_begin);  // rows beyond M are clamped onto M - 1 (only the last tile of A can overhang)
  auto issue_a01 = [&]() { issue4(stage_src(m_begin, 0), voffA, 2); issue4(stage_src(m_begin, 1), voffA, 3); };

  // ---- W -> registers, once, through LDS in full lines ------------------------------------------------------------
  bf16x8 wf[2][12];
  if (WTR == 0) {
    // W [N, K]: the six [128 n][64 k] blocks of the n-tile, staged exactly like A stages; fragment (nt, 2 s + sub)
    unsigned voffW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = wid + 4 * i, r = 8 * p + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      voffW[i] = ((unsigned)min(n0 + r, eN - 1) * (unsigned)P.ldb + (unsigned)c * 8u) * 2u;
    }
    const char* wb = (const char*)P.B;
    const unsigned wa = sbase + fa0 + (unsigned)(wid * 32 * 128);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      const int s0 = round ? 4 : 0, ns = round ? 2 : 4;
#pragma unroll
      for (int s = 0; s < ns; ++s) issue4(wb + 128 * (s0 + s), voffW, s);
      if (round == 1) issue_a01();  // slots 2, 3 are free in the second round: A's first stages ride on the same wait
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int s = 0; s < ns; ++s) {
        u32x4 a, b, c, d;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:2048\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                     : "v"(wa + s * WR_STAGE), "v"((wa + s * WR_STAGE) ^ 64u)
                     : "memory");
        wf[0][2 * (s0 + s)] = wr_frag(a); wf[1][2 * (s0 + s)] = wr_frag(b);
        wf[0][2 * (s0 + s) + 1] = wr_frag(c); wf[1][2 * (s0 + s) + 1] = wr_frag(d);
      }
      asm volatile("s_barrier" ::: "memory");
    }
  } else {
    // W [K, N]: twelve [32 k][128 cols] blocks (8 KiB, 4 k-rows x 256 B per piece, chunk c of row kk at c ^ 2 sigma(kk));
    // fragments by two ds_read_b64_tr_b16 (k rows +0..3 / +4..7 of the lane group's 8)
    unsigned voffT[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = wid + 4 * i, kk = 4 * j + (lane >> 4);
      const int c = (lane & 15) ^ (2 * wr_sigma(kk));
      voffT[i] = ((unsigned)kk * (unsigned)P.ldb + (unsigned)min(n0 + c * 8, eN - 8)) * 2u;
    }
    unsigned ta[2];
    {
      const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, sw = 2 * (q | ((g & 1) << 2)), kk0 = 8 * g + q;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int c8 = ((32 * wid + 16 * nt) >> 2) + p;
        ta[nt] = sbase + (unsigned)(kk0 * 256 + (((c8 >> 1) ^ sw) << 4) + ((c8 & 1) << 3));
      }
    }
    const char* wb = (const char*)P.B;
    const long long kstep_bytes = (long long)32 * P.ldb * 2;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      const int s0 = round ? 8 : 0, ns = round ? 4 : 8;
#pragma unroll
      for (int s = 0; s < ns; ++s) {
        const char* base = wb + (long long)(s0 + s) * kstep_bytes;
        asm volatile("" : "+s"(base));
#pragma unroll
        for (int i = 0; i < 2; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(base + voffT[i]), (lptr_t)(smem + s * 8192 + 1024 * (wid + 4 * i)), 16, 0, 0);
      }
      if (round == 1) issue_a01();  // the second round fills 32 KiB (slots 0, 1): A's first stages ride on the same wait
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int s = 0; s < ns; ++s) {
        u32x2 l0, h0, l1, h1;
        asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:1024\n\t"
                     "ds_read_b64_tr_b16 %2, %5\n\tds_read_b64_tr_b16 %3, %5 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1)
                     : "v"(ta[0] + s * 8192), "v"(ta[1] + s * 8192)
                     : "memory");
        wf[0][s0 + s] = wr_frag(l0, h0);
        wf[1][s0 + s] = wr_frag(l1, h1);
      }
      asm volatile("s_barrier" ::: "memory");
    }
  }

  // ---- epilogue operands, read from kernarg memory once -----------------------------------------------------------
  constexpr bool HAS_RES = (FEAT & WF_RES) != 0, HAS_C2 = (FEAT & WF_C2) != 0, HAS_DROP = (FEAT & WF_DROP) != 0;
  constexpr bool NEED_Z = (ACT == VG_ACT_MUL_Z || ACT == VG_ACT_MUL_Z8), Z8 = (ACT == VG_ACT_MUL_Z8);
  bf16* const eC = P.C; const int eldc = P.ldc;
  bf16* const eC2 = P.C2; const int eldc2 = P.ldc2; const int ec2g = P.c2_gelu_grad;
  const float* const ebias = P.bias;
  const bf16* const eres = NEED_Z ? P.Z : P.res; const int eldr = NEED_Z ? P.ldz : P.ldr;
  const unsigned dthr = P.drop_thresh, dkey = vg_drop_key(P.drop_key, P.drop_step); const float dscale = P.drop_scale;
  const int drm = P.drop_row_mul > 1 ? P.drop_row_mul : 1;

  // ---- A pipeline: stages 0, 1 are in slots 2, 3 (landed with the W load), stage 2 goes to slot 0 now ---------------
  issue4(stage_src(m_begin, 2), voffA, 0);

  u32x4 F[WR_NS];
  f32x4 acc[2][8];
  const unsigned fa1 = fa0 ^ 64u;
  int base_slot = 2;  // slot of (tile, ks) = (base_slot + ks) & 3; six stages per tile

  // One tile of MT m-tiles (16 MT rows) at row m0.  `first`: no epilogue stores precede it; `has_next`: another tile
  // (its stages 0..2 are issued from here) follows at m0 + 128.
  auto tile = [&](auto mt_c, int m0, bool first, bool has_next) {
    constexpr int MT = decltype(mt_c)::value;
    constexpr int SPS = 2 * MT, STEPS = WR_KS * SPS, LA = SPS < 11 ? SPS : 11;
    constexpr int EST = 8 * (HAS_C2 ? 2 : 1);  // stores of the previous (always full) tile's epilogue
    auto frag_addr = [&](int q) {
      const int ks = q / SPS, sub = (q % SPS) / MT;
      return sbase + (unsigned)(((base_slot + ks) & 3) * WR_STAGE) + (sub ? fa1 : fa0);
    };
    // the tile's first LA fragments (its stage 0 has landed: previous tile's last barrier, or the prologue's)
#pragma unroll
    for (int q = 0; q < LA; ++q)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(F[q % WR_NS]) : "v"(frag_addr(q)), "n"(2048 * ((q % SPS) % MT)) : "memory");
    __builtin_amdgcn_s_setprio(1);  // the MFMA stream outranks the co-resident workgroup's epilogue / DMA issue (step -0.3 %)
#pragma unroll
    for (int ks = 0; ks < WR_KS; ++ks) {
      // B(g): my pieces of stage g+1 have landed (stage g+2's and, right after an epilogue, its stores may still fly);
      // behind the barrier stage g+1 is complete in LDS and nobody reads stage g-1 any more
      if (!has_next && ks == 5) {
      } else if (!has_next && ks == 4) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      } else if (!first && ks <= 1) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 + EST) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      }
      {  // stage g+3 -> the slot stage g-1 has left
        const int ks3 = ks + 3;
        if (ks == 3 && has_next && m0 + 256 > eM) set_voff(m0 + 128);  // from here on the pieces are the next tile's
        if (ks3 < WR_KS) issue4(stage_src(m0, ks3), voffA, (base_slot + ks3) & 3);
        else if (has_next) issue4(stage_src(m0 + 128, ks3 - WR_KS), voffA, (base_slot + ks3) & 3);
      }
#pragma unroll
      for (int j = 0; j < SPS; ++j) {
        const int q = ks * SPS + j, sub = j / MT, mt = j % MT, qn = q + LA;
        const bool rd = qn < STEPS;
        const int cnt = rd ? LA - 1 : STEPS - 1 - q;
        const unsigned ad = frag_addr(rd ? qn : q);
        const int off = 2048 * ((qn % SPS) % MT);
        if (ks == 0 && sub == 0) {
          if (rd)
            asm volatile("s_waitcnt lgkmcnt(%8)\n\tv_mfma_f32_16x16x32_bf16 %0, %3, %5, 0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %5, 0\n\t"
                         "ds_read_b128 %2, %6 offset:%7"
                         : "=&v"(acc[0][mt]), "=&v"(acc[1][mt]), "=&v"(F[qn % WR_NS])
                         : "v"(wf[0][0]), "v"(wf[1][0]), "v"(F[q % WR_NS]), "v"(ad), "n"(off), "n"(cnt)
                         : "memory");
          else
            asm volatile("s_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %4, 0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %4, 0"
                         : "=&v"(acc[0][mt]), "=&v"(acc[1][mt])
                         : "v"(wf[0][0]), "v"(wf[1][0]), "v"(F[q % WR_NS]), "n"(cnt)
                         : "memory");
        } else {
          if (rd)
            asm volatile("s_waitcnt lgkmcnt(%8)\n\tv_mfma_f32_16x16x32_bf16 %0, %3, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t"
                         "ds_read_b128 %2, %6 offset:%7"
                         : "+v"(acc[0][mt]), "+v"(acc[1][mt]), "=&v"(F[qn % WR_NS])
                         : "v"(wf[0][2 * ks + sub]), "v"(wf[1][2 * ks + sub]), "v"(F[q % WR_NS]), "v"(ad), "n"(off), "n"(cnt)
                         : "memory");
          else
            asm volatile("s_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %4, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %4, %1"
                         : "+v"(acc[0][mt]), "+v"(acc[1][mt])
                         : "v"(wf[0][2 * ks + sub]), "v"(wf[1][2 * ks + sub]), "v"(F[q % WR_NS]), "n"(cnt)
                         : "memory");
        }
      }
    }
    base_slot = (base_slot + WR_KS) & 3;
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMAs' results, before the VALU reads them

    // ---- epilogue: a lane holds, per (n-tile, m-tile), 4 consecutive n of row li; v_permlane16_swap between the two n-tiles
    // gives it 8 consecutive n of one tile -> 16-byte loads / stores straight from registers, MT slots per lane
    int ln = lane;
    asm volatile("" : "+v"(ln));  // addresses are recomputed per tile (hoisted they would be live across the main loop)
    const int g = ln >> 4, li = ln & 15;
    if (ebias) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 b4 = *(const f32x4*)(ebias + n0 + 32 * wid + 16 * nt + 4 * g);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] += b4;
      }
    }
    const int ncol = n0 + 32 * wid + ((g & 1) << 4) + ((g & 2) << 2);
    constexpr int GS = (MT % 4 == 0) ? 4 : 2;  // slot groups: a group's loads all precede its first store
#pragma unroll
    for (int h = 0; h < MT / GS; ++h) {
      bf16x8 pre[(HAS_RES || NEED_Z) ? GS : 1];
      if (HAS_RES || NEED_Z) {
#pragma unroll
        for (int qq = 0; qq < GS; ++qq) {
          const int m = m0 + 16 * (GS * h + qq) + li;
          bf16x8 zb = {0, 0, 0, 0, 0, 0, 0, 0};
          if (Z8) {  // 8 bytes per lane: the codes of its 8 columns, carried in the first half of the slot
            u32x2 c8 = {0u, 0u};
            if (eres && m < eM) c8 = *(const u32x2*)((const unsigned char*)eres + (unsigned)(m * eldr + ncol));
            union { u32x4 u; bf16x8 b; } cv; cv.u = (u32x4){c8[0], c8[1], 0u, 0u};
            zb = cv.b;
          } else if (eres && m < eM) zb = *(const bf16x8*)(eres + (unsigned)(m * eldr + ncol));
          pre[qq] = zb;
        }
      }
#pragma unroll
      for (int qq = 0; qq < GS; ++qq) {
        const int mt = GS * h + qq;
        const int m = m0 + 16 * mt + li;
        const f32x4 te = acc[0][mt], to = acc[1][mt];
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // rows 1,3 of the even tile's register <-> rows 0,2 of the odd tile's
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(te[r]), __float_as_uint(to[r]), false, false);
          v[r] = __uint_as_float(sw[0]);
          v[r + 4] = __uint_as_float(sw[1]);
        }
        if (m >= eM) continue;
        if (ACT == VG_ACT_GELU) {
          float ga[8], gd[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) vg_gelu_both(v[r], ga[r], gd[r]);
          if (HAS_C2 && eC2) {
            if (ec2g == 2) {
              const u32x2 c8 = {vg_g8_pack4(gd[0], gd[1], gd[2], gd[3]), vg_g8_pack4(gd[4], gd[5], gd[6], gd[7])};
              *(u32x2*)((unsigned char*)eC2 + (unsigned)(m * eldc2 + ncol)) = c8;
            } else {
              bf16x8 o;
#pragma unroll
              for (int r = 0; r < 8; ++r) o[r] = vg_f2bf(ec2g ? gd[r] : v[r]);
              *(bf16x8*)(eC2 + (unsigned)(m * eldc2 + ncol)) = o;
            }
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = ga[r];
        } else if (HAS_C2 && eC2) {
          bf16x8 o;
#pragma unroll
          for (int r = 0; r < 8; ++r) o[r] = vg_f2bf(v[r]);
          *(bf16x8*)(eC2 + (unsigned)(m * eldc2 + ncol)) = o;
        }
        if (ACT == VG_ACT_TANH) {  // the classifier's fc1 (modules.py:196-197)
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = vg_tanh(v[r]);
        }
        if (Z8) {
          union { bf16x8 b; u32x4 u; } cv; cv.b = pre[qq];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] *= vg_g8_value(cv.u[0], r); v[r + 4] *= vg_g8_value(cv.u[1], r); }
        } else if (NEED_Z) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] *= vg_bf2f(pre[qq][r]);
        }
        if (HAS_DROP && dthr) {
          const unsigned i4 = (unsigned)(m * drm * eN + ncol) >> 2;
          const unsigned dw0 = vg_drop_word(dkey, i4), dw1 = vg_drop_word(dkey, i4 + 1);
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] *= vg_drop_factor(dw0, r, dthr, dscale); v[r + 4] *= vg_drop_factor(dw1, r, dthr, dscale); }
        }
        if (HAS_RES && !NEED_Z && eres) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += vg_bf2f(pre[qq][r]);
        }
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = vg_f2bf(v[r]);
        *(bf16x8*)(eC + (unsigned)(m * eldc + ncol)) = o;
      }
    }
  };

  int m0 = m_begin;
#pragma unroll 1
  for (int i = 0; i < nfull; ++i, m0 += 128) {
    const bool has_next = i + 1 < ntiles;
    tile(std::integral_constant<int, 8>{}, m0, i == 0, has_next);
  }
  if (rem == 1) tile(std::integral_constant<int, 2>{}, m0, nfull == 0, false);
  else if (rem == 2) tile(std::integral_constant<int, 4>{}, m0, nfull == 0, false);
  else if (rem == 3) tile(std::integral_constant<int, 6>{}, m0, nfull == 0, false);
}

// Launch when the problem is of the kind this kernel covers; returns 1 if enqueued, 0 if not eligible (the caller falls
// back to the tiled kernel), < 0 on a launch error.
int vg_gemm_wr_try(const VgGemmProb& p, int mode, hipStream_t stream) {
  if (mode != VG_NT && mode != VG_NN) return 0;
  if (p.K != 384 || (p.N & 127) || p.M < 256 || (p.M & 31)) return 0;
  if ((p.lda & 7) || (p.ldb & 7) || (p.ldc & 7)) return 0;
  if (p.resf || p.row_in_per > 0 || p.pre_f32 || p.Cf || !p.C) return 0;
  // 16-byte epilogue accesses at base + 32-bit element offsets, for every operand the epilogue touches
  int ldmax = p.N > p.ldc ? p.N : p.ldc;
  if (p.res) { if (p.ldr & 7) return 0; if (p.ldr > ldmax) ldmax = p.ldr; }
  if (p.C2) { if (p.ldc2 & 7) return 0; if (p.ldc2 > ldmax) ldmax = p.ldc2; }
  if (p.act == VG_ACT_MUL_Z || p.act == VG_ACT_MUL_Z8) { if (!p.Z || (p.ldz & 7)) return 0; if (p.ldz > ldmax) ldmax = p.ldz; }
  if ((long long)(p.M + 128) * (long long)ldmax >= (1LL << 31)) return 0;
  if ((long long)p.M * p.lda * 2 >= (1LL << 32)) return 0;  // 32-bit lane offsets inside a tile only, but keep A itself addressable
  int feat = 0;
  if (p.res) feat |= WF_RES;
  if (p.C2) feat |= WF_C2;
  if (p.drop_thresh) feat |= WF_DROP;
  if (p.drop_thresh && p.drop_post) return 0;
  const int n_tiles = p.N / 128;
  if (n_tiles > 64) return 0;
  VgWrArgs a;
  a.p = p;
  a.units = p.M / 32;
  a.n_tiles = n_tiles;
  a.gpx = 64 / n_tiles;  // two workgroups per CU, 32 CUs per XCD
  const dim3 grid(8 * 64), block(256);
#define WR_LAUNCH(WTR_, ACT_, FEAT_) hipLaunchKernelGGL((vg_gemm_wr_kernel<WTR_, ACT_, FEAT_>), grid, block, 0, stream, a)
  if (mode == VG_NT) {
    if (p.act == VG_ACT_NONE) {
      if (feat == 0) WR_LAUNCH(0, VG_ACT_NONE, 0);
      else if (feat == WF_RES) WR_LAUNCH(0, VG_ACT_NONE, WF_RES);
      else if (feat == (WF_RES | WF_DROP)) WR_LAUNCH(0, VG_ACT_NONE, WF_RES | WF_DROP);
      else return 0;
    } else if (p.act == VG_ACT_GELU && feat == WF_C2) {
      WR_LAUNCH(0, VG_ACT_GELU, WF_C2);
    } else if (p.act == VG_ACT_TANH && feat == 0) {  // classifier fc1 at M = B >= 256: 16 us on the tiled kernel's 12 workgroups
      WR_LAUNCH(0, VG_ACT_TANH, 0);
    } else {
      return 0;
    }
  } else {
    if (feat != 0) return 0;
    if (p.act == VG_ACT_NONE) WR_LAUNCH(1, VG_ACT_NONE, 0);  // Z / Zf are only read by the activations that name them
    else if (p.act == VG_ACT_MUL_Z && p.Z) WR_LAUNCH(1, VG_ACT_MUL_Z, 0);
    else if (p.act == VG_ACT_MUL_Z8 && p.Z) WR_LAUNCH(1, VG_ACT_MUL_Z8, 0);
    else return 0;
  }
#undef WR_LAUNCH
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 1 : -(int)e;
}
