// Row chains: the MLP half of an encoder block as ONE launch (src/v2/modules.py:181-182 and the norm1 of the next block,
// :168): a1 = gelu(xn W1^T + b1); Y = res + drop(a1 W2^T + b2); Yn = LayerNorm(Y).  gfx950 only.
//
// Why: op by op every Linear is its own launch with an HBM round trip on both sides, and inside each launch the MFMA phase
// (k loop) and the HBM phase (epilogue) run one after the other.  Here a wave owns 16 rows for the whole chain:
// fc1's A operand (12 fragments), the 64-column slice of the hidden it is working on and fc2's 16 x 384 accumulators all
// live in its registers; the hidden is WRITTEN (a1 and the gelu' codes, for the backward) but never read back.  Only the
// weights move: the chain image (vg_chain.h) streams through a 6-slot LDS ring once per workgroup tile (as many 16-row units =
// active waves as fill the chip's 256 CUs once, at most 8; waves without rows only carry their share of the stream), one 24-KiB
// stage per barrier, three stages of LDS-DMA in flight; a fragment read is `stage + 1024 f + 16 lane`, one per MFMA, through a
// rolling register queue with a counted lgkmcnt - the step (wait, MFMA, next read) is one asm statement, so the schedule
// is the source order (gemm_wr.hip's protocol).
//
// Schedule of the 48 stages (fc1 runs one 64-column group ahead of fc2, so the GELU of group t+1 sits under fc2's MFMAs of
// group t):   W1(0) | W1(1) W2(0) | W1(2) W2(1) | ... | W1(11) W2(10) | W2(11)        W1(t): 2 stages [64 n][192 k]: hidden
// columns 64 t .. 64 t + 63 over k halves; W2(t): 2 stages [384 n][32 k]: fc2's k-steps 2 t, 2 t + 1.
#include "vg_chain.h"
#include <type_traits>

#pragma clang fp contract(off)

namespace {
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int CH_NS = 6;                       // ring slots
constexpr int CH_RING = CH_NS * VG_CH_STAGE;   // 147 456 B
#ifndef CH_QN_
#define CH_QN_ 6
#endif
constexpr int CH_QN = CH_QN_, CH_LA = CH_QN - 1;  // fragment queue slots / look-ahead
static_assert(24 % CH_QN == 0, "a stage must start at queue slot 0");
// Behind the ring: the chain's bias vectors in fp32.  Inside the stream NOTHING may be loaded from global memory by compiler-
// visible code: hipcc cannot count through the inline-asm steps and guards the first use of such a load with
// `s_waitcnt vmcnt(0) lgkmcnt(0)`, which drains the three stages of LDS-DMA in flight and the fragment queue (first version:
// 2.0 us per stage instead of 0.7).  A bias quad is an extra ds_read_b128 inside the counted fragment queue instead.
constexpr int CH_PAR_B1 = 0, CH_PAR_B2 = VG_CH_HID * 4, CH_PAR_GAM = CH_PAR_B2 + VG_CH_E * 4, CH_PAR_BET = CH_PAR_GAM + VG_CH_E * 4,
              CH_PAR_BO = CH_PAR_BET + VG_CH_E * 4, CH_PAR_GAM2 = CH_PAR_BO + VG_CH_E * 4, CH_PAR_BET2 = CH_PAR_GAM2 + VG_CH_E * 4,
              CH_PAR = CH_PAR_BET2 + VG_CH_E * 4;  // b1 | b2 | gamma | beta | (front:) bo | gamma2 | beta2
static_assert(CH_RING + CH_PAR <= 160 * 1024, "LDS");

__device__ __forceinline__ uint32_t ch_pk(float a, float b) {
  const bf16x2 v = {(bf16)a, (bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float ch_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float ch_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }

// gelu and gelu' of two values at once (vg_phi_e of vg_common.h, the same constants and association): the polynomial, the
// products and the fused multiply-adds go out as packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of
// work per issue slot), only |x|, the reciprocal, the exponential and the sign transfer stay scalar: 21 instructions per PAIR
// where the scalar form takes 16 per value.  The GELU of the hidden is what this kernel's vector pipe is busy with.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 ch_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void ch_gelu_both2(f32x2 x, f32x2& g, f32x2& dg) {
  const f32x2 one = {1.0f, 1.0f};
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 den = ch_fma2(ax, (f32x2){0.3275911f * 0.70710678118654752f, 0.3275911f * 0.70710678118654752f}, one);
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const f32x2 y = x * 0.84932180028801907f;
  const f32x2 yy = y * y;
  const f32x2 e = {__builtin_amdgcn_exp2f(-yy[0]), __builtin_amdgcn_exp2f(-yy[1])};
  f32x2 p = ch_fma2(t, (f32x2){1.061405429f, 1.061405429f}, (f32x2){-1.453152027f, -1.453152027f});
  p = ch_fma2(t, p, (f32x2){1.421413741f, 1.421413741f});
  p = ch_fma2(t, p, (f32x2){-0.284496736f, -0.284496736f});
  p = ch_fma2(t, p, (f32x2){0.254829592f, 0.254829592f});
  p = p * t;
  const f32x2 em = ch_fma2(-p, e, one);
  const f32x2 er = {copysignf(em[0], x[0]), copysignf(em[1], x[1])};
  const f32x2 phi = ch_fma2(er, (f32x2){0.5f, 0.5f}, (f32x2){0.5f, 0.5f});
  g = x * phi;
  dg = ch_fma2(x * 0.39894228040143268f, e, phi);
}

// diagnostic builds only (make var SRC=chain NAME=.. DEFS=-DCH_DBG=n): 1 no GELU arithmetic, 2 no a1 / code stores, 4 no fragment reads
// (MFMA stream alone), 8 no MFMAs (fragment reads alone), 16 no LDS-DMA behind the prologue, 32 no stage barriers
#ifndef CH_DBG
#define CH_DBG 0
#endif
// one step of the fragment stream: frag q has arrived -> MFMA -> request frag q + LA
#if (CH_DBG & 4)
#define CH_STEP(ACC, FQ, AF, FN, ADDR, OFF)                                                      \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0" : "+v"(ACC), "+v"(FN) : "v"(FQ), "v"(AF), "v"(ADDR) : "memory")
#elif (CH_DBG & 8)
#define CH_STEP(ACC, FQ, AF, FN, ADDR, OFF)                                                                         \
  asm volatile("s_waitcnt lgkmcnt(%5)\n\tds_read_b128 %1, %4 offset:%6" \
               : "+v"(ACC), "=&v"(FN)                                                                                \
               : "v"(FQ), "v"(AF), "v"(ADDR), "n"(CH_LA - 1), "n"(OFF)                                               \
               : "memory")
#else
#define CH_STEP(ACC, FQ, AF, FN, ADDR, OFF)                                                                         \
  asm volatile("s_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tds_read_b128 %1, %4 offset:%6" \
               : "+v"(ACC), "=&v"(FN)                                                                                \
               : "v"(FQ), "v"(AF), "v"(ADDR), "n"(CH_LA - 1), "n"(OFF)                                               \
               : "memory")
#endif
}  // namespace

// FRONT: the out-projection, its dropout + residual and norm2 in front of the MLP (modules.py:179-180, :172): 12 more K-major stages
// from the attention output's fragments, an epilogue in registers that leaves x_mid and xn2 in memory (the backward reads them)
// and xn2 in the A fragments fc1 multiplies - k order S (vg_chain.h): the 8 consecutive columns a lane holds behind the swap.
template <bool FRONT>
__global__ __launch_bounds__(512) void vg_chain_fwd_kernel(const VgChainMlpArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[CH_RING + CH_PAR + (((CH_DBG & 64) && !FRONT) ? 8 * 104 * 8 : 0)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned sbase = (unsigned)(unsigned long)(lptr_t)smem;
  const unsigned lane16 = (unsigned)lane * 16u;
  constexpr int S = VG_CH_MLP_STAGES + (FRONT ? VG_CH_FRONT_STAGES : 0);
  const int upw = a.upw;  // 16-row units (= active waves) of a workgroup tile
  const int ntiles = (a.units + upw - 1) / upw;
  const int g = lane >> 4, li = lane & 15;
  const int cg = ((g & 1) << 4) + ((g & 2) << 2);  // first of the lane's 8 consecutive columns inside a tile pair, after the swap

  // kernel arguments, read once (an s_load in the middle of the stream would sit in the counted lgkmcnt queue)
  const char* img = (const char*)a.img;
  const float* b1 = a.b1; const float* b2 = a.b2;
  bf16* a1p = a.a1; unsigned char* z8p = a.z8;
  asm volatile("" : "+s"(img), "+s"(b1), "+s"(b2), "+s"(a1p), "+s"(z8p));

  for (int i = tid; i < CH_PAR / 16; i += 512) {
    const int seg = i / (VG_CH_E / 4);  // 0,1: b1; 2: b2; 3: gamma; 4: beta; 5: bo; 6: gamma2; 7: beta2
    const float* src = seg < 2 ? b1 + 4 * i
                     : seg == 2 ? b2 + 4 * i - VG_CH_HID
                     : seg == 3 ? (a.Yn ? a.gamma + 4 * i - 3 * VG_CH_E : nullptr)
                     : seg == 4 ? (a.Yn ? a.beta + 4 * i - 4 * VG_CH_E : nullptr)
                     : !FRONT ? nullptr
                     : seg == 5 ? (a.bo ? a.bo + 4 * i - 5 * VG_CH_E : nullptr)
                     : seg == 6 ? a.gamma2 + 4 * i - 6 * VG_CH_E : a.beta2 + 4 * i - 7 * VG_CH_E;
    *(f32x4*)(smem + CH_RING + 16 * i) = src ? *(const f32x4*)src : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();

#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int nact = min(upw, a.units - upw * tile);
    const bool active = wid < nact;  // wave-uniform; a wave without rows still loads its share of every stage and meets every barrier
    const int row = (upw * tile + (active ? wid : 0)) * 16 + li;  // this lane's row

    auto issue = [&](int s) {  // this wave's three 1-KiB pieces of stage s -> slot s % NS
      const char* src = img + (size_t)s * VG_CH_STAGE + 3072 * wid;
      asm volatile("" : "+s"(src));
      unsigned char* d = smem + (s % CH_NS) * VG_CH_STAGE + 3072 * wid;
#pragma unroll
      for (int i = 0; i < 3; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + 1024 * i + lane16), (lptr_t)(d + 1024 * i), 16, 0, 0);
    };
    // The first Linear's A operand: rows of xn (FRONT: of the attention output) as 12 fragments, k order natural (16 B per lane and
    // k-step); requested before the ring's first stages so that the first counted wait covers them.  FRONT: the epilogue behind
    // the out-projection overwrites them with xn2's.
    u32x4 A[12];
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      A[s] = (u32x4){0u, 0u, 0u, 0u};
      if (active) A[s] = *(const u32x4*)((FRONT ? a.ao : a.xn) + (size_t)row * a.ldx + 32 * s + 8 * g);
    }
#pragma unroll
    for (int s = 0; s < CH_NS - 1; ++s) issue(s);
    f32x4 acc2[24], acc1[4];
#pragma unroll
    for (int j = 0; j < 24; ++j) acc2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 hid[2][2];  // fc2's A fragments of the group in flight: [t & 1][k-step], k order P
    u32x4 F[CH_QN];
    unsigned cur = sbase + lane16 - VG_CH_STAGE, nxt = sbase + lane16;  // fragment addresses of the current / next stage

    int s_idx = 0;
    auto stage_top = [&](bool early = false) {
      // my pieces of stage s+1 have landed; stages s+2 .. s+4 may still be in flight, and so may the a1 / code stores issued since
      // the pieces of stage s+1 went out: behind the tile's first two stages ALWAYS four of them (two per K-major stage, two such
      // stages among any four; the four of group 0 behind stage 1), so 13 operations may stay outstanding - counting the stores as
      // pieces (vmcnt(9)) made every top wait for a stage issued 1.7 stages ago instead of 3 (+10 us per launch).
      // Behind the barrier stage s+1 is whole and nobody reads stage s-1 any more: its slot takes stage s+5.
      const int ahead = S - 2 - s_idx;
      if ((CH_DBG & 64) && !FRONT) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (lane == 0) *(unsigned long long*)(smem + CH_RING + CH_PAR + (wid * 104 + 2 * s_idx) * 8) = t0;
      }
      if (ahead >= 3) { if (early || !active) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); }  // (a wave without rows stores nothing)
      else if (ahead == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(CH_DBG & 32)) asm volatile("s_barrier" ::: "memory");
      if ((CH_DBG & 64) && !FRONT) {
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0) *(unsigned long long*)(smem + CH_RING + CH_PAR + (wid * 104 + 2 * s_idx + 1) * 8) = t1;
      }
      if (!(CH_DBG & 16) && s_idx + CH_NS - 1 < S) issue(s_idx + CH_NS - 1);
      cur = nxt;
      nxt += VG_CH_STAGE;
      if (nxt >= sbase + lane16 + CH_RING) nxt -= CH_RING;
      // opaque: in the unrolled stretches the compiler otherwise computes the addresses of every stage up front, keeps them
      // live across the stream, spills them - and reloads one per stage behind a vmcnt(0) that drains the ring
      asm volatile("" : "+v"(cur), "+v"(nxt));
      ++s_idx;
    };
    // N-major stage [64 n][192 k] (fragment f = 4 ss + jj: consecutive MFMAs go to different accumulators): acc1[jj] += W1 frag x A[6 U + ss]
    auto stage_n = [&acc1, &F, &A, &cur, &nxt, active](auto u_c) {
      constexpr int U = decltype(u_c)::value;
      if (!active) return;  // a wave without rows only carries its share of the weight stream (stage_top)
#pragma unroll
      for (int f = 0; f < 24; ++f) {
        const int jj = f & 3, ss = f >> 2, fa = f + CH_LA;
        if (fa < 24) CH_STEP(acc1[jj], F[f % CH_QN], A[6 * U + ss], F[fa % CH_QN], cur, fa * 1024);
        else CH_STEP(acc1[jj], F[f % CH_QN], A[6 * U + ss], F[fa % CH_QN], nxt, (fa - 24) * 1024);
      }
    };
    // K-major stage [384 n][32 k] (fragment f = n-tile): acc2[f] += W2 frag x hid; `hook(f)` runs VALU work under the MFMAs
    auto stage_k = [&acc2, &F, &cur, &nxt, active](const u32x4& hf, auto&& hook) {
      if (!active) return;
#pragma unroll
      for (int f = 0; f < 24; ++f) {
        const int fa = f + CH_LA;
        if (fa < 24) CH_STEP(acc2[f], F[f % CH_QN], hf, F[fa % CH_QN], cur, fa * 1024);
        else CH_STEP(acc2[f], F[f % CH_QN], hf, F[fa % CH_QN], nxt, (fa - 24) * 1024);
        hook(f);
      }
    };
    // GELU of hidden tile jj of group t (columns 64 t + 16 jj + 4 g ..): bias, gelu, gelu' -> packed bf16 pair registers
    // (fc2's fragment, k order P) and one register of byte codes; the accumulator is cleared for the next group
    uint32_t codes[4];
    u32x4 bq[2];  // bias quads in flight (requested >= LA steps before their use: the counted waits of the steps in between cover them)
    // address of b1[64 t + 4 g ..] in the parameter block (+ 64 jj per tile): rebuilt per group from the lane offset that is live anyway -
    // kept across the stream it was spilled and came back behind a vmcnt(0), twice per group
    auto bias_addr = [&](int t) {
      unsigned l16 = lane16;
      asm volatile("" : "+v"(l16));
      return sbase + (unsigned)(CH_RING + CH_PAR_B1) + 256u * (unsigned)t + ((l16 >> 4) & 0x30u);
    };
    unsigned pbt = bias_addr(0);
    auto bias_req = [&bq, &pbt](auto jj_c) {
      constexpr int JJ = decltype(jj_c)::value;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(bq[JJ & 1]) : "v"(pbt), "n"(64 * JJ) : "memory");
    };
    auto gelu_tile = [&bq, &acc1, &codes](auto jj_c, u32x4 (&hdst)[2]) {
      constexpr int JJ = decltype(jj_c)::value;
      asm volatile("" : "+v"(bq[JJ & 1]), "+v"(acc1[JJ]));  // pinned behind the step it is called after (asm volatiles keep their order)
      const f32x4 bv = __builtin_bit_cast(f32x4, bq[JJ & 1]);
      const f32x4 z = acc1[JJ] + bv;
      f32x2 g0, d0, g1, d1;
      if (CH_DBG & 1) { g0 = d0 = (f32x2){z[0], z[1]}; g1 = d1 = (f32x2){z[2], z[3]}; }
      else { ch_gelu_both2((f32x2){z[0], z[1]}, g0, d0); ch_gelu_both2((f32x2){z[2], z[3]}, g1, d1); }
      hdst[JJ >> 1][2 * (JJ & 1)] = ch_pk(g0[0], g0[1]);
      hdst[JJ >> 1][2 * (JJ & 1) + 1] = ch_pk(g1[0], g1[1]);
      codes[JJ] = vg_g8_pack4(d0[0], d0[1], d1[0], d1[1]);
      acc1[JJ] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    // a1 / codes of tile pair SP of group t: permlane16_swap hands each lane 8 consecutive columns -> 16-B / 8-B stores
    auto store_pair = [&](auto sp_c, int t, const u32x4 (&h)[2]) {
      constexpr int SP = decltype(sp_c)::value;
      const auto s0 = __builtin_amdgcn_permlane16_swap(h[SP][0], h[SP][2], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(h[SP][1], h[SP][3], false, false);
      const auto sc = __builtin_amdgcn_permlane16_swap(codes[2 * SP], codes[2 * SP + 1], false, false);
      if (active && !(CH_DBG & 2)) {
        const size_t o = (size_t)row * VG_CH_HID + 64 * t + 32 * SP + cg;
        *(u32x4*)(a1p + o) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
        *(u32x2*)(z8p + o) = (u32x2){sc[0], sc[1]};
      }
    };
    auto no_hook = [](int) {};
    // Row epilogue in registers, for the 16 x 384 accumulators of a Linear whose output is the embedding.  Per tile pair p a lane
    // holds, behind the permlane swap, columns 32 p + cg .. + 7 of its row:  y = res + drop(acc + bias), rounded to bf16 ONCE (what
    // the unfused kernels stored) and written; with LN: the statistics of the rounded values, yn = LayerNorm(y) written and -
    // FRAGS - left in `fr` as the next Linear's A fragments (k order S: exactly these 8 columns per lane and k-step).
    // Everything it addresses with is derived HERE from opaque copies: hoisted above the stream its addresses would stay live
    // across it and be spilled (first version: 87 registers, reloaded one vmcnt(0) at a time).
    auto row_epi = [&](auto frags_c, auto batch_c, f32x4 (&acc)[24], u32x4 (&fr)[12], int par_bias, int par_gam, int par_bet, unsigned key_host,
                       const bf16* resp, bf16* Yp, bf16* Ynp, float* meanp, float* rstdp) {
      constexpr bool FRAGS = decltype(frags_c)::value;
      constexpr int RB = decltype(batch_c)::value;  // residual chunks requested ahead (12: all of them, where the registers are free)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      asm volatile("" : "+s"(resp), "+s"(Yp), "+s"(Ynp), "+s"(meanp), "+s"(rstdp));
      const int eg = ln >> 4;
      const int ecg = ((eg & 1) << 4) + ((eg & 2) << 2);
      const int erow = (a.upw * tile + wid) * 16 + (ln & 15);
      const unsigned char* par = smem + CH_RING;
      const unsigned dthr = a.drop_thresh, dkey = vg_drop_key(key_host, a.drop_step);
      const float dscale = a.drop_scale;
      const unsigned drm = a.drop_row_mul > 1 ? (unsigned)a.drop_row_mul : 1u;
      u32x4 rv[RB];
      auto ld_res = [&](int p) {
        u32x4 r = (u32x4){0u, 0u, 0u, 0u};
        if (resp) r = *(const u32x4*)(resp + (size_t)erow * VG_CH_E + 32 * p + ecg);
        return r;
      };
#pragma unroll
      for (int p = 0; p < RB; ++p) rv[p] = ld_res(p);
      u32x4 yk[12];
      float sm = 0.f;
#pragma unroll
      for (int p = 0; p < 12; ++p) {
        const int c0 = 32 * p + ecg;
        const u32x4 rc = rv[p % RB];
        if (p + RB < 12) rv[p % RB] = ld_res(p + RB);
        const f32x4 bA = *(const f32x4*)(par + par_bias + 4 * c0), bB = *(const f32x4*)(par + par_bias + 4 * c0 + 16);
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * p][r]), __float_as_uint(acc[2 * p + 1][r]), false, false);
          v[r] = __uint_as_float(sw[0]) + bA[r];
          v[r + 4] = __uint_as_float(sw[1]) + bB[r];
        }
        if (dthr) {
          const unsigned i4 = ((unsigned)erow * drm * (unsigned)VG_CH_E + (unsigned)c0) >> 2;
          const unsigned w0 = vg_drop_word(dkey, i4), w1 = vg_drop_word(dkey, i4 + 1);
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] *= vg_drop_factor(w0, r, dthr, dscale); v[r + 4] *= vg_drop_factor(w1, r, dthr, dscale); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t w = ch_pk(v[2 * q] + ch_lo(rc[q]), v[2 * q + 1] + ch_hi(rc[q]));
          yk[p][q] = w;
          sm += ch_lo(w); sm += ch_hi(w);
        }
        *(u32x4*)(Yp + (size_t)erow * VG_CH_E + c0) = yk[p];
        asm volatile("" ::: "memory");  // the scheduler otherwise hoists the bias / residual loads of all 12 pairs to the top: 100+ registers
      }
      if (Ynp) {
        sm += __shfl_xor(sm, 16, 64); sm += __shfl_xor(sm, 32, 64);
        const float mu = sm * (1.0f / VG_CH_E);
        float q2 = 0.f;
#pragma unroll
        for (int p = 0; p < 12; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float c0 = ch_lo(yk[p][q]) - mu, c1 = ch_hi(yk[p][q]) - mu;
            q2 += c0 * c0; q2 += c1 * c1;
          }
        q2 += __shfl_xor(q2, 16, 64); q2 += __shfl_xor(q2, 32, 64);
        const float rs = rsqrtf(fmaf(q2, 1.0f / VG_CH_E, a.eps));
        if (eg == 0) { meanp[erow] = mu; rstdp[erow] = rs; }
#pragma unroll
        for (int p = 0; p < 12; ++p) {
          const int c0 = 32 * p + ecg;
          const f32x4 gA = *(const f32x4*)(par + par_gam + 4 * c0), gB = *(const f32x4*)(par + par_gam + 4 * c0 + 16);
          const f32x4 eA = *(const f32x4*)(par + par_bet + 4 * c0), eB = *(const f32x4*)(par + par_bet + 4 * c0 + 16);
          u32x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float g0 = q < 2 ? gA[2 * q] : gB[2 * q - 4], g1 = q < 2 ? gA[2 * q + 1] : gB[2 * q - 3];
            const float e0 = q < 2 ? eA[2 * q] : eB[2 * q - 4], e1 = q < 2 ? eA[2 * q + 1] : eB[2 * q - 3];
            o[q] = ch_pk(fmaf((ch_lo(yk[p][q]) - mu) * rs, g0, e0), fmaf((ch_hi(yk[p][q]) - mu) * rs, g1, e1));
          }
          *(u32x4*)(Ynp + (size_t)erow * VG_CH_E + c0) = o;
          if (FRAGS) fr[p] = o;
          asm volatile("" ::: "memory");
        }
      }
    };

    if (FRONT) {
      // ---- out-projection: 12 K-major stages into the accumulators fc2 uses later; then x_mid, norm2 ----
#pragma unroll
      for (int ks = 0; ks < VG_CH_FRONT_STAGES; ++ks) {
        stage_top(true);  // (no stores in this phase: 9 operations behind the pieces of stage s+1, exactly)
        if (ks == 0 && active) {
#pragma unroll
          for (int q = 0; q < CH_LA; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(F[q]) : "v"(cur), "n"(q * 1024) : "memory");
        }
        stage_k(A[ks], no_hook);
      }
      // every fragment requested ahead has landed (the compiler does not know the queue's registers are still being written)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
      if (active) {
        row_epi(std::true_type{}, std::integral_constant<int, 4>{}, acc2, A, CH_PAR_BO, CH_PAR_GAM2, CH_PAR_BET2, a.drop_key_a, a.xin, a.xmid, a.xn_out,
                a.mean2, a.rstd2);
#pragma unroll
        for (int j = 0; j < 24; ++j) acc2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }

#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // ---- W1(0), its GELU (exposed once) ----
    // (FRONT: the stores and loads of the epilogue above, ~40 of them, are younger than every piece in flight: 13 is safe from here on)
    stage_top(!FRONT);
    if (!FRONT && active) {
#pragma unroll
      for (int q = 0; q < CH_LA; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(F[q]) : "v"(cur), "n"(q * 1024) : "memory");
    }
    stage_n(std::integral_constant<int, 0>{});
    stage_top(!FRONT);
    stage_n(std::integral_constant<int, 1>{});
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMAs' results, before the VALU reads them
    if (active) {
      constexpr std::integral_constant<int, 0> c0; constexpr std::integral_constant<int, 1> c1;
      constexpr std::integral_constant<int, 2> c2; constexpr std::integral_constant<int, 3> c3;
      bias_req(c0); bias_req(c1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (also the look-ahead fragments: once per tile)
      gelu_tile(c0, hid[0]); gelu_tile(c1, hid[0]);
      bias_req(c2); bias_req(c3);
      store_pair(c0, 0, hid[0]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      gelu_tile(c2, hid[0]); gelu_tile(c3, hid[0]);
      store_pair(c1, 0, hid[0]);
    }

    // ---- W1(t+1) then W2(t) with the GELU of group t+1 under it ----
    auto body = [&](auto hb_c, int t) {
      constexpr int HB = decltype(hb_c)::value;
      stage_top(); stage_n(std::integral_constant<int, 0>{});
      stage_top(); stage_n(std::integral_constant<int, 1>{});
      pbt = bias_addr(t + 1);
      stage_top();
      stage_k(hid[HB][0], [&](int f) {
        if (f == 0) bias_req(std::integral_constant<int, 0>{});
        if (f == 8) { gelu_tile(std::integral_constant<int, 0>{}, hid[HB ^ 1]); bias_req(std::integral_constant<int, 1>{}); }
        if (f == 16) gelu_tile(std::integral_constant<int, 1>{}, hid[HB ^ 1]);
        if (f == 22) store_pair(std::integral_constant<int, 0>{}, t + 1, hid[HB ^ 1]);
      });
      stage_top();
      stage_k(hid[HB][1], [&](int f) {
        if (f == 0) bias_req(std::integral_constant<int, 2>{});
        if (f == 8) { gelu_tile(std::integral_constant<int, 2>{}, hid[HB ^ 1]); bias_req(std::integral_constant<int, 3>{}); }
        if (f == 16) gelu_tile(std::integral_constant<int, 3>{}, hid[HB ^ 1]);
        if (f == 22) store_pair(std::integral_constant<int, 1>{}, t + 1, hid[HB ^ 1]);
      });
    };
#pragma unroll 1
    for (int t = 0; t < 10; t += 2) {
      body(std::integral_constant<int, 0>{}, t);
      body(std::integral_constant<int, 1>{}, t + 1);
    }
    body(std::integral_constant<int, 0>{}, 10);
    // ---- W2(11) ----
    stage_top(); stage_k(hid[1][0], no_hook);
    stage_top(); stage_k(hid[1][1], no_hook);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // (the look-ahead reads past the last stage are discarded)

    if ((CH_DBG & 64) && !FRONT) {
      const unsigned long long t2 = __builtin_readcyclecounter();
      if (lane == 0) *(unsigned long long*)(smem + CH_RING + CH_PAR + (wid * 104 + 2 * S) * 8) = t2;
    }
    // ================================================ epilogue ================================================
    // Y = res + drop(fc2 + b2) and the LayerNorm behind it (the next block's norm1); FRONT: the residual is the x_mid this lane
    // stored behind the out-projection (same addresses, same thread: program order makes it visible)
    if (active && !(CH_DBG & 128))
      row_epi(std::false_type{}, std::integral_constant<int, 12>{}, acc2, A, CH_PAR_B2, CH_PAR_GAM, CH_PAR_BET, a.drop_key, FRONT ? (const bf16*)a.xmid : a.res, a.Y,
              a.Yn, a.mean_out, a.rstd_out);
    if ((CH_DBG & 64) && !FRONT) {
      const unsigned long long t3 = __builtin_readcyclecounter();
      if (lane == 0) *(unsigned long long*)(smem + CH_RING + CH_PAR + (wid * 104 + 2 * S + 1) * 8) = t3;
      __syncthreads();
      if (a.stamps && tile == (int)blockIdx.x)
        for (int i = tid; i < 8 * 104; i += 512) a.stamps[(size_t)blockIdx.x * 8 * 104 + i] = *(unsigned long long*)(smem + CH_RING + CH_PAR + i * 8);
    }
    __syncthreads();  // the ring is reused by the next tile of this workgroup
  }
}

// ---- chain image ---------------------------------------------------------------------------------------------------
// thread = one 16-byte chunk (8 bf16) of the image: stage s, fragment f, lane l
__global__ __launch_bounds__(256) void vg_chain_pack_kernel(const bf16* __restrict__ Wo, const bf16* __restrict__ W1, const bf16* __restrict__ W2,
                                                            bf16* __restrict__ img, int k_in, int front) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int nst = VG_CH_MLP_STAGES + (front ? VG_CH_FRONT_STAGES : 0);
  if (c >= nst * 1536) return;
  int s = c / 1536;
  const int f = (c % 1536) >> 6, l = c & 63, g = l >> 4, li = l & 15;
  const bf16* src; int korder;
  if (front && s < VG_CH_FRONT_STAGES) {  // out-projection, [384 n][32 k] per stage, A straight from memory
    src = Wo + (size_t)(16 * f + li) * VG_CH_E + 32 * s;
    korder = VG_CH_KNAT;
  } else {
    if (front) s -= VG_CH_FRONT_STAGES;
    // schedule: stages 0,1 = W1(0); then for t = 0..10: W1(t+1) (2 stages), W2(t) (2 stages); 46,47 = W2(11)
    bool is_w1; int t, u;
    if (s < 2) { is_w1 = true; t = 0; u = s; }
    else if (s >= 46) { is_w1 = false; t = 11; u = s - 46; }
    else { const int q = (s - 2) >> 2, r = (s - 2) & 3; is_w1 = r < 2; t = is_w1 ? q + 1 : q; u = r & 1; }
    if (is_w1) {  // [64 n][192 k]: fragment f = 4 ss + jj
      const int jj = f & 3, ss = f >> 2;
      src = W1 + (size_t)(64 * t + 16 * jj + li) * VG_CH_E + 192 * u + 32 * ss;
      korder = k_in;
    } else {      // [384 n][32 k]: fragment f = n-tile
      src = W2 + (size_t)(16 * f + li) * VG_CH_HID + 64 * t + 32 * u;
      korder = VG_CH_KP;
    }
  }
  bf16x8 v;
  if (korder == VG_CH_KNAT) {
    v = *(const bf16x8*)(src + 8 * g);
  } else if (korder == VG_CH_KS) {
    v = *(const bf16x8*)(src + ((g & 1) << 4) + ((g & 2) << 2));
  } else {
    const bf16x4 lo = *(const bf16x4*)(src + 4 * g), hi = *(const bf16x4*)(src + 16 + 4 * g);
    v = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  *(bf16x8*)(img + (size_t)c * 8) = v;
}

int vg_chain_mlp_pack_launch(const bf16* W1, const bf16* W2, bf16* img, int k_in, hipStream_t st) {
  if (!W1 || !W2 || !img || (k_in != VG_CH_KNAT && k_in != VG_CH_KP && k_in != VG_CH_KS)) return -1;
  hipLaunchKernelGGL(vg_chain_pack_kernel, dim3(VG_CH_MLP_STAGES * 1536 / 256), dim3(256), 0, st, (const bf16*)nullptr, W1, W2, img, k_in, 0);
  return (int)hipGetLastError();
}
int vg_chain_block_pack_launch(const bf16* Wo, const bf16* W1, const bf16* W2, bf16* img, hipStream_t st) {
  if (!Wo || !W1 || !W2 || !img) return -1;
  hipLaunchKernelGGL(vg_chain_pack_kernel, dim3((VG_CH_MLP_STAGES + VG_CH_FRONT_STAGES) * 1536 / 256), dim3(256), 0, st, Wo, W1, W2, img, VG_CH_KS, 1);
  return (int)hipGetLastError();
}

#ifdef VG_TUNING
static unsigned long long* g_ch_stamps = nullptr;
extern "C" void vg_chain_dbg_stamps(void* p) { g_ch_stamps = (unsigned long long*)p; }
#endif
int vg_chain_mlp_fwd_launch(const VgChainMlpArgs& a0, hipStream_t st) {
  VgChainMlpArgs a = a0;
  a.stamps = nullptr;
#ifdef VG_TUNING
  a.stamps = g_ch_stamps;
#endif
  if (a.M < 16 || (a.M & 15) || (a.ldx & 7)) return 0;
  const bool front = a.ao != nullptr;
  if (!(front ? (const void*)a.ao : (const void*)a.xn) || !a.img || !a.b1 || !a.b2 || !a.a1 || !a.z8 || !a.Y) return -1;
  if (a.Yn && (!a.mean_out || !a.rstd_out || !a.gamma || !a.beta)) return -1;
  if (front && (!a.xmid || !a.xn_out || !a.mean2 || !a.rstd2 || !a.gamma2 || !a.beta2)) return -1;
  if ((long long)a.M * (a.drop_row_mul > 1 ? a.drop_row_mul : 1) * VG_CH_E >= (1LL << 32)) return 0;  // dropout index arithmetic is 32-bit
  a.units = a.M / 16;
  // units per workgroup tile = active waves: as few as fill the chip's 256 CUs once (a small problem is one or two waves per
  // workgroup, the other waves only carry their share of the weight stream), at most the 8 waves a CU's registers hold
  a.upw = (a.units + 255) / 256;
  if (a.upw > 8) a.upw = 8;
  const int ntiles = (a.units + a.upw - 1) / a.upw;
  const int grid = ntiles < 256 ? ntiles : 256;
  if (front) hipLaunchKernelGGL(vg_chain_fwd_kernel<true>, dim3(grid), dim3(512), 0, st, a);
  else hipLaunchKernelGGL(vg_chain_fwd_kernel<false>, dim3(grid), dim3(512), 0, st, a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 1 : -(int)e;
}
