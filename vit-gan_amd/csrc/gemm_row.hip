// Full-row GEMM for the Linears whose OUTPUT is the embedding width (N = 384), with the LayerNorm next to them in the
// epilogue.  gfx950 only.
//
//   VG_ROW_LNFWD   y = res + drop(A W^T + b);  yn = LayerNorm(y)          out-projection (K = 384) and fc2 (K = 768) of a block
//                  (src/v2/modules.py:179-183: the residual adds, and norm2 / the next block's norm1 at :168,172)
//   VG_ROW_LNBWD   dx = gres + LayerNorm'(A W);  dxm = dx * mask           fc1 (K = 768) and QKV (K = 1152) input gradients +
//                  the backward of the LayerNorm that fed them (autograd of :178-181)
//
// Why: the standalone LayerNorm kernels were 70 launches and 15 % of the step, and each moved a [M, 384] tensor through HBM
// only to re-read it in the next launch.  A LayerNorm needs whole rows, so a workgroup here owns whole rows: 8 waves side by
// side along n (48 columns = 3 MFMA tiles each) over MT <= 9 m-tiles (16 MT rows), one workgroup per CU.  Rows are dealt in
// units of 16: M = 65 * 2^k never divides into 128-row tiles over 256 CUs (260 tiles = two rounds, the second one empty), 2080
// units over 256 workgroups are 8 or 9 each - one tile per workgroup at 89 % balance.
//
// Operands: the weight comes PACKED (vg_pack_rows_kernel): one 24-KiB image per 32-deep k-stage, [384 n][32 k] with the
// row-form XOR swizzle already applied, transposed on the way for the input gradients - so a W stage is 24 fully contiguous
// 1-KiB LDS-DMA pieces (full 128-byte lines: half-line pieces cost a whole request each) and both passes are the
// NT form with ds_read_b128 fragments (10-12 reads per 24-27 MFMAs; the transposing reads of the NN form were what limited
// gemm_tn.hip).  A is staged per stage as [16 MT rows][32 k].  4-slot ring, two stages in flight, fragments of stage s+1
// requested before the MFMAs of stage s, the two waves of a SIMD in opposite order - the protocol of gemm_tn.hip.
//
// Epilogue: the accumulators go to LDS once as a bf16 tile [16 MT][384] (the ring is free by then) - exactly the rounding
// the unfused pair had at its HBM round trip - and are re-read row-wise, 16 lanes per row, by the LayerNorm code of norm.hip:
// every global access of the epilogue is then a whole 768-byte row in 16-byte pieces.
#include "vg_row.h"
#include <type_traits>
#include <stdlib.h>

// The epilogue arithmetic is instantiated once per tile height (MT = 1..9), and a row lands in tiles of different heights
// depending on the batch it is part of: with the compiler free to contract a*b+c here and not there, the same row came out
// one ulp different in a batch of 64 and of 256 (tests/test_fullsize_gpu.py: per-sample bit-independence).  Contraction is
// therefore OFF in this file and every fused multiply-add is written as fmaf().
#pragma clang fp contract(off)

namespace {
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#ifndef RW_MAXMT_
#define RW_MAXMT_ 9
#endif
constexpr int RW_NSLOT = 4;
// Everything that depends on the output width N (= the embedding width: 384 for C1-C3, 512 for C4; round 4).  8 waves side by side own
// N / 8 columns each = NT MFMA n-tiles; the accumulators (4 NT MT registers) cap the tile height: 9 m-tiles at NT = 3, 6 at NT = 4.
// N = 768 (C5) does not fit: one 32-deep W stage is 48 KiB, a ring of 3 with its A stages and the LayerNorm vectors 162 KiB.
template <int N> struct RwCfg {
  static_assert(N == 384 || N == 512, "widths this kernel is built for");
  static constexpr int NT = N / 128;                      // n-tiles per wave
  static constexpr int WIMG = N * 64;                     // one W stage image: N rows x 64 B
  static constexpr int MAXMT = N == 384 ? RW_MAXMT_ : 6;  // m-tiles per tile (tuning builds: make var DEFS=-DRW_MAXMT_=5)
  static constexpr int STAGE = WIMG + MAXMT * 1024;       // + A stage image: 16 MT rows x 64 B
  static constexpr int RING = RW_NSLOT * STAGE;           // 135 168 B / 155 648 B
  // LNBWD column-sum fold over the ring: [3 sums][32 row-group slots][N] fp32 (+ the SLN scalars) at once where it fits (N = 384:
  // round 3's fold, its association unchanged), one sum at a time otherwise
  static constexpr bool FOLD3 = (3 * 32 * N + 16) * 4 <= 160 * 1024 - 2 * N * 4;
  static constexpr int RED = ((FOLD3 ? 3 : 1) * 32 * N + 16) * 4;
  static constexpr int BODY = RED > RING ? RED : RING;
  static constexpr int GAM = 2 * N * 4;                   // behind both: gamma (and the SLN's bias) in fp32 - the LNBWD epilogue has no registers for them
  static constexpr int TS = 2 * N + 16;                   // row stride of the bf16 epilogue tile (LNBWD): + 16 (ds_write_b64 2-way at worst)
  static constexpr int TSF = 4 * N + 16;                  // row stride of the fp32 epilogue tile (LNFWD, HM m-tiles at a time)
  static constexpr int HM = N == 384 ? 5 : 4;
  static constexpr int CPL = N / 128;                     // chunks per lane and row of the row-wise epilogue passes (128 columns per round of the lanes)
  static_assert(MAXMT * 16 * TS <= RING, "epilogue tile must fit the ring");
  static_assert(HM * 16 * TSF <= RING, "fp32 half tile must fit the ring");
  static_assert(BODY + GAM <= 160 * 1024, "LDS");
};

// row-form chunk swizzle of gemm.hip: 16-B chunk c of row r lives at position c ^ {0,2,3,1}[(r>>2)&3]
__device__ __host__ __forceinline__ int rw_row_f(int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; }

__device__ __forceinline__ void rw_wait_vm(int n) {  // n is wave-uniform (scalar branch)
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
// Sum over the 16 lanes of a row group, in every lane, by DPP row rotations (8, 4, 2, 1): the association tree - and so the
// result, bit for bit, in every lane - is that of the xor butterfly of norm.hip, without its four trips through the LDS
// crossbar (ds_bpermute + lgkmcnt), which nothing hides at two waves per SIMD.
template <int N>
__device__ __forceinline__ float rw_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float rw_row16_sum(float v) {
  v += rw_ror<8>(v); v += rw_ror<4>(v); v += rw_ror<2>(v); v += rw_ror<1>(v);
  return v;
}
}  // namespace

#ifdef VG_TUNING  // diagnostic builds: a.dbg bits 8 = no epilogue stores, 16 = no phase-2 loads, 32 = no phase-1 loads, 64 = no epilogue at all
#define RW_DBG(bit) (a.dbg & (bit))
#else
#define RW_DBG(bit) 0
#endif
template <int EPI, bool SLN, int N>
__global__ __launch_bounds__(512, 2) void vg_gemm_row_kernel(const VgRowArgs a) {
  using Cf = RwCfg<N>;
  constexpr int RW_E = N, NT = Cf::NT, RW_WIMG = Cf::WIMG, RW_MAXMT = Cf::MAXMT, RW_STAGE = Cf::STAGE, RW_BODY = Cf::BODY, RW_TS = Cf::TS,
                RW_TSF = Cf::TSF, CPL = Cf::CPL;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RW_BODY + Cf::GAM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wid >> 2;  // the two waves of a SIMD are w and w + 4
  const int u0 = (int)((long long)blockIdx.x * a.units / a.nwg), u1 = (int)((long long)(blockIdx.x + 1) * a.units / a.nwg);
  if (u0 >= u1) return;  // never: the launcher keeps nwg <= units
  const int nsteps = a.K >> 5;
  const unsigned sbase = (unsigned)(unsigned long)(lptr_t)smem;
  const long long a8 = (long long)128 * a.lda * 2;  // bytes from piece 0 to piece 8

  // N = 512, forward: gamma / beta live behind the ring too - 64 registers of them on top of 4 chunks of residual, modulation and row values
  // spilled 92 registers in the self-modulated form
  constexpr bool GB_LDS = N > 384;
  constexpr bool PEN = (EPI == VG_ROW_LNBWD_PEN);  // the gradient penalty's variant of the LayerNorm-backward epilogue: + gres2, + dy_out
  if (EPI != VG_ROW_LNFWD && tid < RW_E / 4) {  // visible to everyone behind the first tile's barriers
    *(f32x4*)(smem + RW_BODY + 16 * tid) = *(const f32x4*)(a.gamma + 4 * tid);
    if (SLN) *(f32x4*)(smem + RW_BODY + RW_E * 4 + 16 * tid) = *(const f32x4*)(a.lbias + 4 * tid);
  }
  if (EPI == VG_ROW_LNFWD && GB_LDS && a.Yn && tid < RW_E / 4) {
    *(f32x4*)(smem + RW_BODY + 16 * tid) = *(const f32x4*)(a.gamma + 4 * tid);
    *(f32x4*)(smem + RW_BODY + RW_E * 4 + 16 * tid) = *(const f32x4*)(a.beta + 4 * tid);
  }
  float cum_s = 0.f;               // LNBWD + SLN: threads 0 / 1 carry d gs / d bs
  float cum[3] = {0.f, 0.f, 0.f};  // LNBWD: this thread's columns tid, tid + 512, tid + 1024 of the workgroup's partial row

  auto tile = [&](auto mt_c, const int m0) {
    constexpr int MT = decltype(mt_c)::value;
    // lane-derived addresses are recomputed per tile from an opaque copy of the lane id: carried across the tile loop their
    // live ranges would span the epilogue, and the allocator spills them INTO the k loop (a reload there drains the DMA ring)
    int lnm = lane;
    asm volatile("" : "+v"(lnm));
    // fragment addresses inside a stage: W rows 48 wid + 16 nt + li, A rows 16 mt + li; the swizzle only sees li
    const unsigned fsw = (unsigned)((((lnm >> 4) ^ rw_row_f(lnm & 15))) << 4);
    const unsigned fw = sbase + (unsigned)((16 * NT * wid + (lnm & 15)) * 64) + fsw;  // n-tile nt at + 1024 nt
    const unsigned fa = sbase + (unsigned)(RW_WIMG + (lnm & 15) * 64) + fsw;        // m-tile mt at + 1024 mt
    // LDS-DMA lane offsets: W pieces are contiguous; an A piece is 16 rows x 64 B, position lane&3 of row lane>>2 holds
    // chunk (lane&3) ^ f(row)
    const unsigned offW = (unsigned)lnm * 16u;
    const unsigned offA = ((unsigned)(lnm >> 2) * (unsigned)a.lda + (unsigned)(((lnm & 3) ^ rw_row_f(lnm >> 2)) << 3)) * 2u;
    const int pps = NT + (wid < MT ? 1 : 0) + ((MT == 9 && wid == 0) ? 1 : 0);  // LDS-DMA pieces of this wave per stage
    const char* wptr = (const char*)a.Wp + 1024 * wid;
    const char* aptr = (const char*)a.A + ((long long)(m0 + 16 * wid) * a.lda) * 2;
    auto issue = [&](int slot) {
      asm volatile("" : "+s"(wptr), "+s"(aptr));
      unsigned char* d = smem + slot * RW_STAGE + 1024 * wid;
#ifdef VG_TUNING  // diagnostic builds (make var SRC=gemm_row): a.dbg bit 2 = W pieces re-read stage 0, bit 4 = A pieces re-read stage 0
      const char* wsrc = (a.dbg & 2) ? (const char*)a.Wp + 1024 * wid : wptr;
      const char* asrc = (a.dbg & 4) ? (const char*)a.A + ((long long)(m0 + 16 * wid) * a.lda) * 2 : aptr;
#else
      const char* wsrc = wptr; const char* asrc = aptr;
#endif
#pragma unroll
      for (int i = 0; i < NT; ++i)  // piece wid + 8 i of the W stage's 8 NT contiguous 1-KiB pieces
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + 8192 * i + offW), (lptr_t)(d + 8192 * i), 16, 0, 0);
      if (wid < MT) __builtin_amdgcn_global_load_lds((gptr_t)(asrc + offA), (lptr_t)(d + RW_WIMG), 16, 0, 0);
      if (MT == 9 && wid == 0) __builtin_amdgcn_global_load_lds((gptr_t)(asrc + a8 + offA), (lptr_t)(d + RW_WIMG + 8192), 16, 0, 0);
      wptr += RW_WIMG; aptr += 64;
    };
    struct Frags { u32x4 w[NT]; u32x4 m[MT]; };
    auto read_frags = [&](Frags& f, int slot) {
      const unsigned so = (unsigned)(slot * RW_STAGE);
      const unsigned w0 = fw + so, a0 = fa + so;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.w[nt]) : "v"(w0), "n"(1024 * nt) : "memory");
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.m[mt]) : "v"(a0), "n"(1024 * mt) : "memory");
    };
    auto wait_frags = [&](Frags& f) {  // the registers are tied behind the wait: no use of them can be scheduled above it
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2])::"memory");
      if (NT > 3) asm volatile("" : "+v"(f.w[NT - 1])::"memory");
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(f.m[mt])::"memory");
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: three stages in flight, fragments of stage 0 ----
    issue(0); issue(1); issue(2);
    rw_wait_vm(2 * pps);
    asm volatile("s_barrier" ::: "memory");
    Frags f0, f1;
    read_frags(f0, 0);

    // One stage (gemm_tn.hip): B(s) = stage s+1 landed and nobody reads stage s-1 any more; DMA of stage s+3 into the slot
    // stage s-1 left; fragment reads of stage s+1 into the other register set; this stage's MFMAs.
    auto stage = [&](Frags& cur, Frags& nxt, int s) {
      if (s + 1 < nsteps) {
        // my pieces of stage s+1 have landed; those of stage s+2 may still be in flight
        if (s + 2 < nsteps) rw_wait_vm(pps); else rw_wait_vm(0);
        asm volatile("s_barrier" ::: "memory");
      }
      wait_frags(cur);  // requested a stage ago
      if (half == 0) {
        if (s + 3 < nsteps) issue((s + 3) & 3);
        if (s + 1 < nsteps) read_frags(nxt, (s + 1) & 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      {
        bf16x8 fm[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fm[mt] = __builtin_bit_cast(bf16x8, cur.m[mt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const bf16x8 fn = __builtin_bit_cast(bf16x8, cur.w[nt]);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = vg_mfma(fn, fm[mt], acc[nt][mt]);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (half != 0) {
        if (s + 3 < nsteps) issue((s + 3) & 3);
        if (s + 1 < nsteps) read_frags(nxt, (s + 1) & 3);
      }
    };
#pragma unroll 1
    for (int s = 0; s < nsteps; s += 2) {
      stage(f0, f1, s);
      stage(f1, f0, s + 1);
    }
    // every wave has its last fragments in registers (lgkmcnt(0) above) and no DMA is in flight: the ring is free
    asm volatile("s_barrier" ::: "memory");

    if (RW_DBG(64)) { if (acc[0][0][0] == 12345.f) a.Y[0] = (bf16)1.f; return; }
    // ======================================================= epilogue =======================================================
    // Phase 1: a lane holds C[m = 16 mt + li][n = 48 wid + 16 nt + 4 g + r], r = 0..3, and writes it to a tile in LDS (the ring
    // is free).  Phase 2: the tile is re-read row-wise, 16 lanes per row (norm.hip's layout: lane `sub` owns the 16-byte chunks
    // sub, sub + 16, sub + 32; wave w takes rows 32 ps + 4 w + rg of pass ps), so every global access is a whole row in 16-byte
    // pieces.  With two waves per SIMD nothing hides a load's latency: the global loads of pass ps+1 are issued before pass ps is
    // computed (those of pass 0 before phase 1), and the row sums go over DPP rotations, not the LDS crossbar.
    // Everything the epilogue reads through is made opaque HERE: otherwise the loads of gamma / beta / bias (invariant over
    // the tiles of a workgroup) are hoisted above the k loop, stay live across it, and the main loop - at 250 registers on its
    // own - spills a fragment address and reloads it behind a vmcnt(0) every stage, draining the DMA ring.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int sub = ln & 15, rg = ln >> 4, g = ln >> 4, li = ln & 15;
    VgRowArgs e = a;
    asm volatile("" : "+s"(e.bias), "+s"(e.res), "+s"(e.Y), "+s"(e.Yn), "+s"(e.mean_out), "+s"(e.rstd_out), "+s"(e.beta), "+s"(e.gamma));
    asm volatile("" : "+s"(e.x), "+s"(e.mean), "+s"(e.rstd), "+s"(e.gres), "+s"(e.dx), "+s"(e.dxm));
    asm volatile("" : "+s"(e.wmod), "+s"(e.gs), "+s"(e.bs), "+s"(e.dw_acc), "+s"(e.resf));
    if (PEN) asm volatile("" : "+s"(e.gres2), "+s"(e.dy_out));
    const float g_s = SLN ? e.gs[0] : 1.f, b_s = SLN ? e.bs[0] : 0.f;
    if (EPI == VG_ROW_LNFWD) {
      // LNFWD keeps the sum in fp32 until the residual is added (ONE rounding, as the unfused epilogue had): fp32 tile rows of
      // 1552 B, 5 m-tiles (80 rows) at a time
      constexpr int HM = Cf::HM, NH = (MT + HM - 1) / HM;
      const unsigned dthr = a.drop_thresh, dkey = vg_drop_key(a.drop_key, a.drop_step);
      const float dscale = a.drop_scale;
      const unsigned drm = a.drop_row_mul > 1 ? (unsigned)a.drop_row_mul : 1u;
      const size_t ldres = a.ldr > 0 ? (size_t)a.ldr : (size_t)RW_E;
      f32x4 b4[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        b4[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (e.bias) b4[nt] = *(const f32x4*)(e.bias + 16 * NT * wid + 16 * nt + 4 * g);
      }
      float gam[GB_LDS ? 1 : CPL][8], bet[GB_LDS ? 1 : CPL][8];
      if (e.Yn && !GB_LDS) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int c = 8 * (sub + 16 * i);
          const f32x4 g0 = *(const f32x4*)(e.gamma + c), g1 = *(const f32x4*)(e.gamma + c + 4);
          const f32x4 b0 = *(const f32x4*)(e.beta + c), b1 = *(const f32x4*)(e.beta + c + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { gam[GB_LDS ? 0 : i][j] = g0[j]; gam[GB_LDS ? 0 : i][j + 4] = g1[j]; bet[GB_LDS ? 0 : i][j] = b0[j]; bet[GB_LDS ? 0 : i][j + 4] = b1[j]; }
        }
      }
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        constexpr int dummy = 0; (void)dummy;
        const int mt0 = h * HM;
        const int hm = (MT - mt0) < HM ? (MT - mt0) : HM;  // m-tiles of this half (compile-time after unrolling)
        const int rows = 16 * hm, passes = (rows + 31) / 32;
        // residual rows of pass 0 (phase-2 layout): in flight across the dump
        bf16x8 rn[CPL];  // residual rows of the NEXT pass
        bf16x8 wn[SLN ? CPL : 1];  // SLN: modulation rows of the next pass
        auto ld_res = [&](bf16x8 (&dst)[CPL], int ps) {
          const int rl = 32 * ps + 4 * wid + rg;
          const size_t row = (size_t)(m0 + 16 * mt0 + (rl < rows ? rl : 0));
#pragma unroll
          for (int i = 0; i < CPL; ++i) {
            dst[i] = (bf16x8){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
            if (e.res && !RW_DBG(32)) dst[i] = *(const bf16x8*)(e.res + row * ldres + 8 * (sub + 16 * i));
            if (SLN) wn[SLN ? i : 0] = *(const bf16x8*)(e.wmod + row * RW_E + 8 * (sub + 16 * i));
          }
        };
        ld_res(rn, 0);
        if (h > 0) __syncthreads();  // the previous half's rows have been read
#pragma unroll
        for (int mt = 0; mt < HM; ++mt) {
          if (mt0 + mt < MT) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int n = 16 * NT * wid + 16 * nt + 4 * g;
              f32x4 v = acc[nt][mt0 + mt < MT ? mt0 + mt : 0] + b4[nt];
              if (dthr) {
                const unsigned wd = vg_drop_word(dkey, ((unsigned)(m0 + 16 * (mt0 + mt) + li) * drm * (unsigned)RW_E + (unsigned)n) >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= vg_drop_factor(wd, r, dthr, dscale);
              }
              *(f32x4*)(smem + (16 * mt + li) * RW_TSF + n * 4) = v;
            }
          }
        }
        __syncthreads();
#pragma unroll 1
        for (int ps = 0; ps < passes; ++ps) {
          {
            const int rem = rows - 32 * ps;  // 16 or >= 32: with 16 only waves 0..3 have rows
            bf16x8 rc[CPL], wc[SLN ? CPL : 1];
#pragma unroll
            for (int i = 0; i < CPL; ++i) { rc[i] = rn[i]; if (SLN) wc[SLN ? i : 0] = wn[SLN ? i : 0]; }
            if (ps + 1 < passes) ld_res(rn, ps + 1);
            if (rem >= 32 || wid < 4) {
              const int rl = 32 * ps + 4 * wid + rg;
              const size_t row = (size_t)(m0 + 16 * mt0 + rl);
              float v[CPL][8];
              float sm = 0.f;
#pragma unroll
              for (int i = 0; i < CPL; ++i) {
                const f32x4 t0 = *(const f32x4*)(smem + rl * RW_TSF + 32 * (sub + 16 * i));
                const f32x4 t1 = *(const f32x4*)(smem + rl * RW_TSF + 32 * (sub + 16 * i) + 16);
                bf16x8 o;
                if (e.resf) {  // residual from an fp32 table, broadcast over the batch (block 0 of the generator: the learned embedding)
                  const float* rf = e.resf + (size_t)((int)(row % (size_t)a.res_period)) * RW_E + 8 * (sub + 16 * i);
                  const f32x4 r0 = *(const f32x4*)rf, r1 = *(const f32x4*)(rf + 4);
#pragma unroll
                  for (int j = 0; j < 4; ++j) { o[j] = vg_f2bf(t0[j] + r0[j]); o[j + 4] = vg_f2bf(t1[j] + r1[j]); }
                } else {
#pragma unroll
                  for (int j = 0; j < 4; ++j) {
                    o[j] = vg_f2bf(t0[j] + vg_bf2f(rc[i][j]));
                    o[j + 4] = vg_f2bf(t1[j] + vg_bf2f(rc[i][j + 4]));
                  }
                }
                if (!RW_DBG(8)) *(bf16x8*)(e.Y + row * RW_E + 8 * (sub + 16 * i)) = o;
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[i][j] = vg_bf2f(o[j]); sm += v[i][j]; }
              }
              if (e.Yn) {
                const float mu = rw_row16_sum(sm) * (1.0f / RW_E);
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < CPL; ++i)
#pragma unroll
                  for (int j = 0; j < 8; ++j) { const float c = v[i][j] - mu; q += c * c; }  // unfused, as norm.hip's build has it
                const float rs = rsqrtf(fmaf(rw_row16_sum(q), 1.0f / RW_E, a.eps));  // norm.hip's contracted form: bit-identical statistics
                if (sub == 0) { e.mean_out[row] = mu; e.rstd_out[row] = rs; }
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                  bf16x8 o;
                  float gl[8], bl[8];
                  if (GB_LDS) {
                    const unsigned char* gp = smem + RW_BODY + 32 * (sub + 16 * i);
                    const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 16);
                    const f32x4 b0 = *(const f32x4*)(gp + RW_E * 4), b1 = *(const f32x4*)(gp + RW_E * 4 + 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { gl[j] = g0[j]; gl[j + 4] = g1[j]; bl[j] = b0[j]; bl[j + 4] = b1[j]; }
                  }
#pragma unroll
                  for (int j = 0; j < 8; ++j) {
                    float r = fmaf((v[i][j] - mu) * rs, GB_LDS ? gl[j] : gam[GB_LDS ? 0 : i][j], GB_LDS ? bl[j] : bet[GB_LDS ? 0 : i][j]);
                    if (SLN) r = vg_bf2f(wc[SLN ? i : 0][j]) * fmaf(g_s, r, b_s);
                    o[j] = vg_f2bf(r);
                  }
                  if (!RW_DBG(8)) *(bf16x8*)(e.Yn + row * RW_E + 8 * (sub + 16 * i)) = o;
                }
              }
            }
          }
        }
      }
    } else {
      // LPR lanes per row: 16 (16-byte chunks, 4 rows per wave and pass) for the plain LayerNorm; 32 (8-byte chunks, 2 rows) for the
      // self-modulated one, whose extra operands (w, d w, the LayerNorm's bias) would not leave room for 72 column accumulators
      // (N = 512: 32 lanes per row for the plain LayerNorm too - 16 lanes would hold 96 column accumulators)
      constexpr int LPR = (SLN || N > 384) ? 32 : 16, CH = 128 / LPR, RPW = 64 / LPR, RPP = 8 * RPW;
      constexpr int ROWS = 16 * MT, PASSES = (ROWS + RPP - 1) / RPP;
      typedef typename std::conditional<CH == 8, bf16x8, bf16x4>::type chunk_t;
      const int subl = ln % LPR, rgl = ln / LPR;
      const unsigned dthr = a.drop_thresh, dkey = vg_drop_key(a.drop_key, a.drop_step);
      const float dscale = a.drop_scale;
      const unsigned drm = a.drop_row_mul > 1 ? (unsigned)a.drop_row_mul : 1u;
      auto zero_chunk = [] { chunk_t z; for (int j = 0; j < CH; ++j) z[j] = (bf16)0.f; return z; };
      auto row_sum = [&](float v) { v = rw_row16_sum(v); if (LPR == 32) v += __shfl_xor(v, 16, 64); return v; };
      chunk_t xn[CPL];  // x rows and statistics of the NEXT pass: in flight while the current pass is computed
      float mun, rsn;
      auto ld_ops = [&](int ps) {
        const int rl = RPP * ps + RPW * wid + rgl;
        const size_t row = (size_t)(m0 + (rl < ROWS ? rl : 0));
        const size_t xrow = a.x_period > 0 ? (size_t)((int)(row % (size_t)a.x_period)) : row;  // x broadcast over the batch (generator block 0)
        mun = e.mean[row]; rsn = e.rstd[row];
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          xn[i] = zero_chunk();
          if (!RW_DBG(16)) xn[i] = *(const chunk_t*)(e.x + xrow * RW_E + CH * (subl + LPR * i));
        }
      };
      ld_ops(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const f32x4 v = acc[nt][mt];
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = vg_f2bf(v[r]);
          *(bf16x4*)(smem + (16 * mt + li) * RW_TS + (16 * NT * wid + 16 * nt + 4 * g) * 2) = o;
        }
      __syncthreads();
      float ag[CPL][CH], ab[CPL][CH], ac[CPL][CH];
#pragma unroll
      for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int j = 0; j < CH; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; ac[i][j] = 0.f; }
      const unsigned char* gam_lds = smem + RW_BODY;
      float s_gs = 0.f, s_bs = 0.f;  // SLN: d gs = sum dy w (xhat gamma + lbias), d bs = sum dy w
      // part == nullptr: a pass that wants the input gradient only (the generator's pass through D) - no column sums, no fold
      const bool sums = SLN || a.part != nullptr;
      auto ld_f32 = [&](const unsigned char* base, int i, float (&dst)[CH]) {
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          const f32x4 t4 = *(const f32x4*)(base + 4 * (CH * (subl + LPR * i) + 4 * q));
#pragma unroll
          for (int j = 0; j < 4; ++j) dst[4 * q + j] = t4[j];
        }
      };
#pragma unroll 1
      for (int ps = 0; ps < PASSES; ++ps) {
        const int rem = ROWS - RPP * ps;  // LPR 16: 16 or >= 32 - with 16 only waves 0..3 have rows; LPR 32: always a whole pass
        chunk_t xc[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) xc[i] = xn[i];
        const float mu = mun, rs = rsn;
        if (ps + 1 < PASSES) ld_ops(ps + 1);
        if (rem >= RPP || wid < 4) {
          const int rl = RPP * ps + RPW * wid + rgl;
          const size_t row = (size_t)(m0 + rl);
          chunk_t gr[CPL];  // the residual-stream gradient is only needed behind the row sums: its latency sits under them
          chunk_t gr2[PEN ? CPL : 1];
          chunk_t wm[SLN ? CPL : 1];
#pragma unroll
          for (int i = 0; i < CPL; ++i) {
            if (SLN) wm[SLN ? i : 0] = *(const chunk_t*)(e.wmod + row * RW_E + CH * (subl + LPR * i));
            gr[i] = zero_chunk();
            if (e.gres && !RW_DBG(16)) gr[i] = *(const chunk_t*)(e.gres + row * RW_E + CH * (subl + LPR * i));
            if (PEN) {
              gr2[PEN ? i : 0] = zero_chunk();
              if (e.gres2) gr2[PEN ? i : 0] = *(const chunk_t*)(e.gres2 + row * RW_E + CH * (subl + LPR * i));
            }
          }
          float xh[CPL][CH], gg[CPL][CH];
          float c1 = 0.f, c2 = 0.f;
#pragma unroll
          for (int i = 0; i < CPL; ++i) {
            const chunk_t t = *(const chunk_t*)(smem + rl * RW_TS + 2 * CH * (subl + LPR * i));
            if (PEN) { if (e.dy_out) *(chunk_t*)(e.dy_out + row * RW_E + CH * (subl + LPR * i)) = t; }  // A W as the unfused pair stored it (bf16)
            float gm[CH], lb[SLN ? CH : 1], dwv[SLN ? CH : 1];
            ld_f32(gam_lds, i, gm);
            if (SLN) {
#pragma unroll
              for (int q = 0; q < CH / 4; ++q) {
                const f32x4 t4 = *(const f32x4*)(gam_lds + RW_E * 4 + 4 * (CH * (subl + LPR * i) + 4 * q));
#pragma unroll
                for (int j = 0; j < 4; ++j) lb[SLN ? 4 * q + j : 0] = t4[j];
              }
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
              const float h = (vg_bf2f(xc[i][j]) - mu) * rs;
              float d = vg_bf2f(t[j]);
              if (SLN) {  // norm.hip's SLN backward: the modulation's own gradients first, then dy_eff = dy w gs
                const float wv = vg_bf2f(wm[SLN ? i : 0][j]);
                const float l = fmaf(h, gm[j], lb[SLN ? j : 0]);
                dwv[SLN ? j : 0] = d * fmaf(g_s, l, b_s);
                const float dwm = d * wv;
                s_gs = fmaf(dwm, l, s_gs);
                s_bs += dwm;
                d = dwm * g_s;
              }
              xh[i][j] = h;
              ag[i][j] = fmaf(d, h, ag[i][j]);  // (unconditional: behind a run-time `sums` each became a v_cndmask - three selects per element)
              ab[i][j] += d;
              const float gv = d * gm[j];
              gg[i][j] = gv;
              c1 += gv;
              c2 = fmaf(gv, h, c2);
            }
            if (SLN) {  // d w: fp32, accumulated over the 2L + 1 uses of the modulation vector (CH = 4: one 16-byte access)
              float* dwp = e.dw_acc + row * RW_E + CH * (subl + LPR * i);
              f32x4 w0 = {dwv[0], dwv[SLN ? 1 : 0], dwv[SLN ? 2 : 0], dwv[SLN ? 3 : 0]};
              if (a.dw_accumulate) w0 += *(const f32x4*)dwp;
              *(f32x4*)dwp = w0;
            }
          }
          c1 = row_sum(c1) * (1.0f / RW_E);
          c2 = row_sum(c2) * (1.0f / RW_E);
#pragma unroll
          for (int i = 0; i < CPL; ++i) {
            const int c = CH * (subl + LPR * i);
            chunk_t o;
#pragma unroll
            for (int j = 0; j < CH; ++j) o[j] = vg_f2bf(fmaf(rs, fmaf(-xh[i][j], c2, gg[i][j] - c1), PEN ? vg_bf2f(gr[i][j]) + vg_bf2f(gr2[PEN ? i : 0][j]) : vg_bf2f(gr[i][j])));
            if (!RW_DBG(8)) *(chunk_t*)(e.dx + row * RW_E + c) = o;
            if (e.dxm) {  // gradient entering the dropped branch: the mask the forward epilogue applied
              const unsigned i4 = ((unsigned)row * drm * (unsigned)RW_E + (unsigned)c) >> 2;
#pragma unroll
              for (int q = 0; q < CH / 4; ++q) {
                const unsigned wd = vg_drop_word(dkey, i4 + q);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[4 * q + j] = vg_f2bf(vg_bf2f(o[4 * q + j]) * vg_drop_factor(wd, j, dthr, dscale));
              }
              if (!RW_DBG(8)) *(chunk_t*)(e.dxm + row * RW_E + c) = o;
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) ac[i][j] += vg_bf2f(o[j]);
          }
        }
      }
      // fold the column sums: every row group of every wave writes its partial row to LDS (over the tile and the ring: both are
      // done with) and 3 x 384 threads add the 8 RPW rows of a column in a fixed order.  (The first version reduced the row groups of
      // a wave with __shfl_xor first: 432 ds_bpermute + lgkmcnt round trips per wave, a visible part of this epilogue.)
      __syncthreads();  // the tile has been read
      float* red = (float*)smem;  // [3 or 1][NSL][N] (+ 16 scalars)
      constexpr int NSL = 8 * RPW;
      constexpr int NSUM = Cf::FOLD3 ? 3 : 1;  // sums folded per round (N = 512: the three sums take the region one after the other)
      const int slot = RPW * wid + rgl;
      auto put = [&](int which, int at) {  // this row group's partial row of sum `which` -> region `at`
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int col = CH * (subl + LPR * i);
#pragma unroll
          for (int q = 0; q < CH / 4; ++q) {
            const float (&src)[CPL][CH] = which == 0 ? ag : (which == 1 ? ab : ac);
            *(f32x4*)(red + (at * NSL + slot) * RW_E + col + 4 * q) = (f32x4){src[i][4 * q], src[i][4 * q + 1], src[i][4 * q + 2], src[i][4 * q + 3]};
          }
        }
      };
      auto fold_col = [&](const float* r0) {  // a wave's row groups first, then the waves pairwise: fixed association
        float t8[8];
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) {
          float t = r0[(RPW * w8) * RW_E];
#pragma unroll
          for (int r = 1; r < RPW; ++r) t += r0[(RPW * w8 + r) * RW_E];
          t8[w8] = t;
        }
        return ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
      };
      if (SLN) {  // the two scalars: wave sums behind the column sums' region
        const float a_ = vg_wave_sum(s_gs), b_ = vg_wave_sum(s_bs);
        if (lane == 0) { red[NSUM * NSL * RW_E + 2 * wid] = a_; red[NSUM * NSL * RW_E + 2 * wid + 1] = b_; }
      }
      if (Cf::FOLD3) {
        if (sums) { put(0, 0); put(1, 1); put(2, 2); }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int c = tid + 512 * k;
          if (sums && c < 3 * RW_E) {
            const int which = c / RW_E, col = c - which * RW_E;
            cum[k] += fold_col(red + (which * NSL) * RW_E + col);
          }
        }
      } else {  // RW_E == 512 threads: thread tid folds column tid of sum k, cum[k] <-> column tid + 512 k of the partial row as above
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (k > 0) __syncthreads();  // the previous sum has been folded
          if (sums) put(k, 0);
          __syncthreads();
          if (sums && tid < RW_E) cum[k] += fold_col(red + tid);
        }
      }
      if (SLN && tid < 2) {
        const float* r0 = red + NSUM * NSL * RW_E + tid;
        cum_s += ((r0[0] + r0[2]) + (r0[4] + r0[6])) + ((r0[8] + r0[10]) + (r0[12] + r0[14]));
      }
    }
  };

  int n = u1 - u0, m0 = u0 * 16;
  bool first = true;
#pragma unroll 1
  while (n > 0) {
    const int mt = n <= RW_MAXMT ? n : (RW_MAXMT == 9 ? 8 : (n >= 2 * RW_MAXMT ? RW_MAXMT : (n + 1) / 2));
    if (!first) __syncthreads();  // the previous tile's epilogue has finished with the ring
    switch (mt) {
      case 1: tile(std::integral_constant<int, 1>{}, m0); break;
      case 2: tile(std::integral_constant<int, 2>{}, m0); break;
      case 3: tile(std::integral_constant<int, 3>{}, m0); break;
      case 4: tile(std::integral_constant<int, 4>{}, m0); break;
      case 5: tile(std::integral_constant<int, 5>{}, m0); break;
      case 6: tile(std::integral_constant<int, 6>{}, m0); break;
      default:
        if constexpr (RW_MAXMT > 6) {  // (the taller tiles exist at N = 384 only: 4 NT MT accumulator registers)
          if (mt == 7) tile(std::integral_constant<int, 7>{}, m0);
          else if (mt == 8) tile(std::integral_constant<int, 8>{}, m0);
          else tile(std::integral_constant<int, RW_MAXMT>{}, m0);
        }
        break;
    }
    n -= mt; m0 += 16 * mt; first = false;
  }
  if (EPI != VG_ROW_LNFWD && a.part) {
    float* out = a.part + (size_t)blockIdx.x * a.part_w;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = tid + 512 * k;
      if (c < 3 * RW_E) out[c] = cum[k];
    }
    if (SLN && tid < 2) out[3 * RW_E + tid] = cum_s;
  }
}

// ---- weight packing -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vg_pack_rows_kernel(const VgPackJobs J) {
  const int NW = J.N;  // output width of the packed problems: rows of a stage image
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;  // 16-byte chunk of the block's packed area
  int q = 0;
  for (; q < J.n; ++q) {
    const long long chunks = (long long)NW * J.d[q].K / 8;
    if (t < chunks) break;
    t -= chunks;
  }
  if (q >= J.n) return;
  const VgPackDesc& D = J.d[q];
  const int s = (int)(t / (NW * 4)), rem = (int)(t - (long long)s * (NW * 4));
  const int n = rem >> 2, pc = rem & 3, c = pc ^ rw_row_f(n), k = 32 * s + 8 * c;
  const bf16* src = J.src + (long long)blockIdx.y * J.src_stride + D.src_off;
  bf16* dst = J.dst + (long long)blockIdx.y * J.dst_stride + D.dst_off + t * 8;
  bf16x8 v;
  if (!D.transposed) {
    v = *(const bf16x8*)(src + (size_t)n * D.ld + k);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k + j) * D.ld + n];
  }
  *(bf16x8*)dst = v;
}

int vg_row_width_ok(int N) { return N == 384 || N == 512; }

int vg_pack_rows_launch(const VgPackJobs& jobs0, hipStream_t st) {
  VgPackJobs jobs = jobs0;
  if (jobs.N == 0) jobs.N = VG_ROW_N;
  if (jobs.n < 1 || jobs.n > 4 || jobs.nblocks < 1 || !jobs.src || !jobs.dst || !vg_row_width_ok(jobs.N)) return -1;
  long long chunks = 0;
  for (int i = 0; i < jobs.n; ++i) {
    const VgPackDesc& d = jobs.d[i];
    if (d.K < 32 || (d.K & 31) || (d.ld & 7) || (d.src_off & 7) || (d.dst_off & 7)) return -3;
    chunks += (long long)jobs.N * d.K / 8;
  }
  if ((jobs.src_stride & 7) || (jobs.dst_stride & 7)) return -3;
  hipLaunchKernelGGL(vg_pack_rows_kernel, dim3((unsigned)((chunks + 255) / 256), jobs.nblocks), dim3(256), 0, st, jobs);
  return (int)hipGetLastError();
}

int vg_row_nwg(int M) {
  if (M < 16 || (M & 15)) return 0;
  const int units = M / 16;
#ifndef RW_MINUNITS
#define RW_MINUNITS 2
#endif
  // at least RW_MINUNITS units per workgroup while the chip is not full.  2, not 4: the generator's launches (8 192 rows = 512
  // units) run 2 us shorter on 256 workgroups of 2 units than on 128 of 4 - a small problem is prologue + epilogue, and both
  // halve with the rows of a workgroup (whole step 6.06 -> 5.98 ms)
  const int want = (units + RW_MINUNITS - 1) / RW_MINUNITS;
  return want < 256 ? want : 256;
}

template <int N>
static int rw_launch(VgRowArgs a, int epi, hipStream_t st) {
  const int nwg = a.nwg;
  const bool sln = a.wmod != nullptr;
  if (epi == VG_ROW_LNFWD) {
    if (sln) hipLaunchKernelGGL((vg_gemm_row_kernel<VG_ROW_LNFWD, true, N>), dim3(nwg), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((vg_gemm_row_kernel<VG_ROW_LNFWD, false, N>), dim3(nwg), dim3(512), 0, st, a);
  } else if (epi == VG_ROW_LNBWD_PEN) {
    hipLaunchKernelGGL((vg_gemm_row_kernel<VG_ROW_LNBWD_PEN, false, N>), dim3(nwg), dim3(512), 0, st, a);
  } else {
    if (sln) hipLaunchKernelGGL((vg_gemm_row_kernel<VG_ROW_LNBWD, true, N>), dim3(nwg), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((vg_gemm_row_kernel<VG_ROW_LNBWD, false, N>), dim3(nwg), dim3(512), 0, st, a);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 1 : -(int)e;
}

int vg_gemm_row_launch(VgRowArgs a, int epi, hipStream_t st) {
  if (a.N == 0) a.N = VG_ROW_N;
  const int RW_E = a.N;
  const int nwg = vg_row_nwg(a.M);
  if (!vg_row_width_ok(a.N) || !nwg || a.K < 128 || (a.K & 63) || (a.lda & 7) || (a.ldr & 7) || !a.A || !a.Wp) return 0;
  if ((long long)a.M * (a.drop_row_mul > 1 ? a.drop_row_mul : 1) * RW_E >= (1LL << 32)) return 0;  // dropout index arithmetic is 32-bit
  a.units = a.M / 16;
  a.nwg = nwg;
#ifdef VG_TUNING
  a.dbg = getenv("VG_ROW_DBG") ? atoi(getenv("VG_ROW_DBG")) : 0;
  if (a.dbg & 1) a.lda = 0;  // every row of A is row 0: the A stream comes from L2
#endif
  const bool sln = a.wmod != nullptr;
  if (sln && (!a.gs || !a.bs)) return -1;
  if (epi == VG_ROW_LNFWD) {
    if (!a.Y || (a.Yn && (!a.mean_out || !a.rstd_out || !a.gamma || !a.beta)) || (sln && !a.Yn)) return -1;
    if (a.resf && (a.res || a.res_period < 1)) return -1;
  } else if (epi == VG_ROW_LNBWD || epi == VG_ROW_LNBWD_PEN) {
    if (!a.x || !a.mean || !a.rstd || !a.gamma || !a.dx || (sln && (!a.part || !a.lbias || !a.dw_acc))) return -1;  // part == nullptr: no column sums (input gradient only)
    if (epi == VG_ROW_LNBWD_PEN && sln) return -4;
    a.part_w = 3 * RW_E + (sln ? 64 : 0);
  } else {
    return -4;
  }
  return a.N == 512 ? rw_launch<512>(a, epi, st) : rw_launch<384>(a, epi, st);
}
