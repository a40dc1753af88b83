// Weight-gradient GEMM with 128 x 384 (or 128 x 512) output tiles:  dW[m, n] = sum_k dY[k, m] X[k, n]  (VG_TN, both operands k-major), for
// the problems whose n extent is a multiple of 384 (E = 384 / 768 models: every Linear of the transformer blocks) or of 512
// (E = 512; template parameter NT = n-tiles per wave, 6 or 8).  gfx950 only.
//
// The tiled kernel of gemm.hip runs these with 128 x 128 tiles: 16 KiB of LDS-DMA per 64 MFMAs, and LDS-DMA issue slots
// are what that kernel is short of.  Here a workgroup (8 waves as 2 (m) x 4 (n), each 64 x 96 = 4 x 6 MFMA
// tiles, 96 accumulator registers) covers 128 x 384: 32 KiB per 192 MFMAs, 1.5x fewer staged bytes per flop, and one
// X row block serves three times the dY columns.  One workgroup per CU (128 KiB ring of four 32-row stages, two in
// flight), two waves per SIMD; the fragment reads of stage s+1 (20 ds_read_b64_tr_b16 per wave) are issued before the 24
// MFMAs of stage s and waited for after them - no wave ever waits for LDS latency, and the co-resident wave covers the
// barrier and the DMA issue.  Split-K slabs, the bias-gradient column sums (ones x dY fragments on the MFMA pipe) and the
// grouped launch are those of the tiled kernel.
#include "vg_gemm.h"
#include <stdlib.h>

namespace {
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int TN_NSLOT = 4;
// stage = [dY: 32 k x 128 m][X: 32 k x (BN / 128) x 128 n], 8 KiB each, tr form (gemm.hip): 32 KiB (NT = 6) or 40 KiB (NT = 8: the
// ring is then all 160 KiB of the CU)
template <int NT> constexpr int tn_stage() { return 8192 * (1 + NT / 2); }

__device__ __forceinline__ int tn_sigma(int kk) { return (kk & 3) | (((kk >> 3) & 1) << 2); }
__device__ __forceinline__ bf16x8 tn_frag(u32x2 lo, u32x2 hi) { return __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]}); }
}  // namespace

template <int NT>
__global__ __launch_bounds__(512, 2) void vg_gemm_tn384_kernel(const VgGemmGroup grp) {
  constexpr int TN_STAGE = tn_stage<NT>(), BN = 64 * NT, NB = NT / 2, PPS = 1 + NB;  // X images, pieces per wave and stage
  __shared__ __attribute__((aligned(16))) unsigned char smem[TN_NSLOT * TN_STAGE];
  int bid;
  {  // XCD-aware block order (gemm.hip): each XCD takes a contiguous run of tiles - the tiles of one K slice share its rows
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
  const int rr = local - split * tiles_mn;
  const int tm = rr / P.tiles_n, tn = rr - tm * P.tiles_n;
  const int m0 = tm * 128, n0 = tn * BN;
  const int k_begin = split * P.k_per_split, k_end = min(P.K, k_begin + P.k_per_split);
  const int nsteps = (k_end - k_begin) >> 5;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const unsigned sbase = (unsigned)(unsigned long)(lptr_t)smem;

  // LDS-DMA: wave w moves piece w (k rows 4w .. 4w+3, 256 B each) of each of the stage's four 8-KiB images
  unsigned offA, offB;
  {
    const int kk = 4 * wid + (lane >> 4);
    const int c = (lane & 15) ^ (2 * tn_sigma(kk));
    offA = ((unsigned)kk * (unsigned)P.lda + (unsigned)(m0 + c * 8)) * 2u;
    offB = ((unsigned)kk * (unsigned)P.ldb + (unsigned)(n0 + c * 8)) * 2u;
  }
  const char* baseA = (const char*)(P.A + (size_t)k_begin * P.lda);
  const char* baseB = (const char*)(P.B + (size_t)k_begin * P.ldb);
  const long long stepA = (long long)32 * P.lda * 2, stepB = (long long)32 * P.ldb * 2;
  // piece i of a stage: 0 = the dY image, 1..NB = the X images; stages are issued strictly in order, the bases run along
  auto issue_piece = [&](int slot, int i) {
    asm volatile("" : "+s"(baseA), "+s"(baseB));
    unsigned char* dst = smem + slot * TN_STAGE + 1024 * wid + 8192 * i;
    if (i == 0) __builtin_amdgcn_global_load_lds((gptr_t)(baseA + offA), (lptr_t)dst, 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((gptr_t)(baseB + offB + 256 * (i - 1)), (lptr_t)dst, 16, 0, 0);
  };
  auto advance = [&]() { baseA += stepA; baseB += stepB; };
  auto issue = [&](int slot) {
#pragma unroll
    for (int i = 0; i < PPS; ++i) issue_piece(slot, i);
    advance();
  };

  // fragment addresses inside a stage (tr form: gemm.hip FragAddr<true>); second half of a fragment at + 1024.  16 columns
  // further on is chunk index + 2 BEFORE the XOR with the row's swizzle (an even number), i.e. address ^ (t << 5) as long as
  // the fragments stay inside one 128-column image: true of the four dY fragments and, at NT = 8 (128 columns per wave), of
  // the X fragments - one address register each instead of 4 + 8 (NT = 8 has none to spare: 128 accumulators, 96 fragment
  // registers).  NT = 6 (96 columns per wave) crosses images and keeps its six X addresses.
  unsigned adA0, adB[NT == 8 ? 1 : NT];
  {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, sw = 2 * (q | ((g & 1) << 2)), kk0 = 8 * g + q;
    {
      const int c8 = ((64 * wm) >> 2) + p;
      adA0 = sbase + (unsigned)(kk0 * 256 + (((c8 >> 1) ^ sw) << 4) + ((c8 & 1) << 3));
    }
#pragma unroll
    for (int nt = 0; nt < (NT == 8 ? 1 : NT); ++nt) {
      const int i0 = 16 * NT * wn + 16 * nt, c8 = ((i0 & 127) >> 2) + p;
      adB[nt] = sbase + (unsigned)(8192 * (1 + (i0 >> 7)) + kk0 * 256 + (((c8 >> 1) ^ sw) << 4) + ((c8 & 1) << 3));
    }
  }
  struct Frags { u32x2 al[4], ah[4], bl[NT], bh[NT]; };
  auto read_frags = [&](Frags& f, int slot) {
    const unsigned so = (unsigned)(slot * TN_STAGE);  // a multiple of 8 KiB: commutes with the XOR of bits 5-7
    const unsigned a0 = adA0 + so;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:1024\n\tds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:1024\n\t"
        "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:1024\n\tds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:1024"
        : "=&v"(f.al[0]), "=&v"(f.ah[0]), "=&v"(f.al[1]), "=&v"(f.ah[1]), "=&v"(f.al[2]), "=&v"(f.ah[2]), "=&v"(f.al[3]), "=&v"(f.ah[3])
        : "v"(a0), "v"(a0 ^ 32u), "v"(a0 ^ 64u), "v"(a0 ^ 96u)
        : "memory");
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const unsigned ab = (NT == 8) ? ((adB[0] + so) ^ (unsigned)(nt << 5)) : (adB[NT == 8 ? 0 : nt] + so);
      asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(f.bl[nt]), "=&v"(f.bh[nt]) : "v"(ab) : "memory");
    }
  };
  auto wait_frags = [&](Frags& f) {  // the registers are tied to the wait, so no use of them can be scheduled above it
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.al[0]), "+v"(f.ah[0]), "+v"(f.al[1]), "+v"(f.ah[1]), "+v"(f.al[2]), "+v"(f.ah[2]), "+v"(f.al[3]), "+v"(f.ah[3])
                 :
                 : "memory");
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(f.bl[nt]), "+v"(f.bh[nt])::"memory");  // behind the same wait
  };

  // bias-gradient column sums ride along (first n-tile, the wn == 0 waves): rows of ones x dY-fragment are all equal
  const bool do_cs = P.colsum != nullptr && tn == 0 && wn == 0;
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;
  f32x4 accb[4], acc[NT][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: up to three stages in flight, fragments of stage 0 ----------------------------------------------
  issue(0);
  if (nsteps > 1) issue(1);
  if (nsteps > 2) issue(2);
  if (nsteps > 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * PPS) : "memory");
  else if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPS) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  Frags f0, f1;
  read_frags(f0, 0);

  // One stage: B(s) (stage s+1 landed, nobody reads stage s-1), DMA of stage s+3 into the slot stage s-1 left, the
  // fragment reads of stage s+1 into the other register set, then this stage's MFMAs.
#ifdef VG_TN_STAMPS  // diagnostic build (make var SRC=gemm_tn NAME=tnst DEFS=-DVG_TN_STAMPS, tools/tn_stamps.py): cycles per segment
  long long tacc[6] = {0, 0, 0, 0, 0, 0};
#define TSTAMP(i) do { const long long _n = (long long)__builtin_amdgcn_s_memtime(); tacc[i] += _n - tprev; tprev = _n; } while (0)
  long long tprev = (long long)__builtin_amdgcn_s_memtime();
  const long long tstart = tprev;
#else
#define TSTAMP(i) do { } while (0)
#endif
  auto stage = [&](Frags& cur, Frags& nxt, int s) {
    if (s + 1 < nsteps) {
      if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    TSTAMP(0);
    // The two waves of a SIMD (w and w + 4) run the rest of the stage in OPPOSITE order: waves 0-3 issue their four DMA
    // pieces and the 20 transposing fragment reads of stage s+1 first and multiply after, waves 4-7 multiply first.  Measured
    // (in-kernel stamps, tools/tn_stamps.py; cycles per stage for 768 of MFMA on the SIMD): both waves in the same order
    // 1670 - the pieces (~70 cycles each) and the reads (~16 each: the LDS queue takes one transposing read per ~4 cycles
    // from the whole CU) ran with the MFMA pipe idle, then both waves' MFMAs queued on it; opposite order 1420; pieces and
    // reads interleaved one by one behind the MFMAs 1590.  The reads are what is left: 0.83 per MFMA at this wave tile.
    wait_frags(cur);  // requested a stage ago
    if (wm == 0) {
      if (s + 3 < nsteps) issue((s + 3) & 3);
      if (s + 1 < nsteps) read_frags(nxt, (s + 1) & 3);
    }
    TSTAMP(1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    {
      bf16x8 fm[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) fm[mt] = tn_frag(cur.al[mt], cur.ah[mt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8 fn = tn_frag(cur.bl[nt], cur.bh[nt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = vg_mfma(fn, fm[mt], acc[nt][mt]);
      }
      if (do_cs) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) accb[mt] = vg_mfma(ones, fm[mt], accb[mt]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    TSTAMP(2);
    if (wm != 0) {
      if (s + 3 < nsteps) issue((s + 3) & 3);
      if (s + 1 < nsteps) read_frags(nxt, (s + 1) & 3);
    }
    TSTAMP(3);
  };
#pragma unroll 1
  for (int s = 0; s < nsteps; s += 2) {
    stage(f0, f1, s);
    if (s + 1 < nsteps) stage(f1, f0, s + 1);
  }

#ifdef VG_TN_STAMPS
  if (grp.zeros && lane == 0) {
    long long* o = (long long*)grp.zeros + ((size_t)bid * 8 + wid) * 8;
    o[0] = tacc[0]; o[1] = tacc[1]; o[2] = tacc[2]; o[3] = tacc[3]; o[4] = tprev - tstart; o[5] = nsteps;
  }
#endif
  // ---- epilogue: fp32 slab of this K slice (+ the column sums) --------------------------------------------------------
  const int g = lane >> 4, li = lane & 15;
  if (do_cs && g == 0) {
    float* cs = P.colsum + (size_t)split * P.colsum_split_stride;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) cs[m0 + wm * 64 + mt * 16 + li] = accb[mt][0];
  }
  float* const Cf = P.Cf + (size_t)split * P.cf_split_stride;
  const int ldcf = P.ldcf;
  const int ncol = n0 + wn * 16 * NT + ((g & 1) << 4) + ((g & 2) << 2);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + 16 * mt + li;
#pragma unroll
    for (int pr = 0; pr < NT / 2; ++pr) {
      const f32x4 te = acc[2 * pr][mt], to = acc[2 * pr + 1][mt];
      f32x4 lo, hi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // rows 1,3 of the even tile's register <-> rows 0,2 of the odd tile's: 8 consecutive n per lane
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(te[r]), __float_as_uint(to[r]), false, false);
        lo[r] = __uint_as_float(sw[0]);
        hi[r] = __uint_as_float(sw[1]);
      }
      float* dst = Cf + (unsigned)(m * ldcf + ncol + 32 * pr);
      *(f32x4*)dst = lo;
      *(f32x4*)(dst + 4) = hi;
    }
  }
}

// 1 = enqueued, 0 = not of this kernel's kind (the caller uses the tiled kernel), < 0 = -hipError.
// May lower p.splits (empty slices are dropped), exactly like vg_gemm_launch.
int vg_gemm_tn384_try(VgGemmProb* probs, int n, hipStream_t stream) {
  if (n < 1 || n > VG_MAX_GROUP) return 0;
  bool w384 = true, w512 = true;
  for (int i = 0; i < n; ++i) {
    const VgGemmProb& p = probs[i];
    if (p.M <= 0 || (p.M & 127) || p.N <= 0 || p.K < 64 || (p.K & 31)) return 0;
    if (p.N % 384) w384 = false;
    if (p.N % 512) w512 = false;
    if ((p.lda & 7) || (p.ldb & 7) || (p.ldcf & 3) || !p.Cf || p.cf_accumulate || p.act != VG_ACT_NONE) return 0;
    if (p.C || p.C2 || p.bias || p.res || p.resf || p.Z || p.Zf || p.row_in_per > 0 || p.drop_thresh) return 0;
    if ((long long)(p.M + 128) * p.ldcf >= (1LL << 31)) return 0;
    if ((long long)64 * (p.lda > p.ldb ? p.lda : p.ldb) * 2 >= (1LL << 31)) return 0;
  }
  if (!w384 && !w512) return 0;
  const int bn = w384 ? 384 : 512;  // one tile width per launch (a group is one kernel)
  VgGemmGroup grp;
  grp.n = n; grp.tpw = 1; grp.zeros = nullptr;
#ifdef VG_TN_STAMPS
  if (getenv("VG_STAMP_PTR")) grp.zeros = (const void*)strtoull(getenv("VG_STAMP_PTR"), nullptr, 16);
#endif
  int total = 0;
  for (int i = 0; i < n; ++i) {
    VgGemmProb& p = probs[i];
    p.tiles_m = p.M / 128;
    p.tiles_n = p.N / bn;
    int splits = p.splits > 0 ? p.splits : 1;
    const int ksteps = p.K / 32;
    const int per = (ksteps + splits - 1) / splits;
    p.k_per_split = per * 32;
    splits = (ksteps + per - 1) / per;  // drop empty slices
    p.splits = splits;
    p.tile_start = total;
    total += p.tiles_m * p.tiles_n * splits;
    grp.p[i] = p;
  }
  grp.total = total;
  if (w384) hipLaunchKernelGGL(vg_gemm_tn384_kernel<6>, dim3(total), dim3(512), 0, stream, grp);
  else hipLaunchKernelGGL(vg_gemm_tn384_kernel<8>, dim3(total), dim3(512), 0, stream, grp);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 1 : -(int)e;
}
