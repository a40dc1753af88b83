// Prototype: "weights in registers" NT GEMM for K = 384 (C[m,n] = sum_k A[m,k] W[n,k] + bias[n]), gfx950.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/wr_proto.hip -o tools/micro/bin/wr_proto
// A workgroup (4 waves) owns one 128-column n-tile for its whole life: wave w keeps W[n0+32w .. +32][0..384) as 24 MFMA
// fragments (96 VGPRs) and walks a run of 128-row m-tiles; only A goes through LDS, as [128 rows][64 k] stages
// filled by LDS-DMA in FULL 128-byte lines (8 rows x 128 B per wave-instruction), 4-slot ring, 2 stages in flight.
#include "../../vit-gan_amd/csrc/vg_common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define WR_KS 6          // 64-deep stages per tile (K = 384)
#define WR_STAGE 16384   // 128 rows x 128 B
#define WR_NSLOT 4

struct WrArgs {
  const bf16* A; const bf16* W; const float* bias; bf16* C; bf16* C2;
  int M, N, lda, ldb, ldc;
  int tiles_m, n_tiles, gpx;  // 128-row tiles, 128-col tiles, m-groups per XCD
  long long* stamps;          // [workgroup][16] s_memtime stamps (wave 0) or null
  int dbg;                    // experiments: 1 = every tile reads tile 0's A rows, 2 = no stores
};

__device__ __forceinline__ void rd8(unsigned addr, u32x4 (&f)[8]) {
  asm volatile(
      "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:2048\n\tds_read_b128 %2, %8 offset:4096\n\tds_read_b128 %3, %8 offset:6144\n\t"
      "ds_read_b128 %4, %8 offset:8192\n\tds_read_b128 %5, %8 offset:10240\n\tds_read_b128 %6, %8 offset:12288\n\tds_read_b128 %7, %8 offset:14336"
      : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7])
      : "v"(addr)
      : "memory");
}
__device__ __forceinline__ void wait8(u32x4 (&f)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
               :
               : "memory");
}
__device__ __forceinline__ void rd2(unsigned addr, u32x4& a, u32x4& b) {  // the two n-tiles of a wave (rows +0, +16)
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2048" : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
}
__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int EST>  // EST: stores of an epilogue that may still be in flight at the next tile's first two barriers
__global__ __launch_bounds__(256, 2) void wr_kernel(const WrArgs P) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[WR_NSLOT * WR_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (m-group, n-tile): the n-tiles of one m-group sit in one XCD (blocks b, b+8, ... share an L2)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  if (idx >= P.gpx * P.n_tiles) return;
  const int group = xcd * P.gpx + idx / P.n_tiles, ntile = idx % P.n_tiles, ngroups = 8 * P.gpx;
  const int t0 = (int)((long long)group * P.tiles_m / ngroups), t1 = (int)((long long)(group + 1) * P.tiles_m / ngroups);
  if (t0 >= t1) return;
  const int n0 = ntile * 128;
  long long* stp = (P.stamps && wid == 0 && lane == 0) ? P.stamps + (size_t)blockIdx.x * 16 : nullptr;
  int sti = 0;
#define STAMP() do { if (stp) { stp[sti] = (long long)__builtin_amdgcn_s_memtime(); } ++sti; } while (0)
  if (stp) stp[15] = (long long)__builtin_amdgcn_s_memrealtime();
  STAMP();
  const unsigned sbase = (unsigned)(unsigned long)(lptr_t)smem;

  // per-lane DMA source offsets (bytes) of this wave's 4 pieces of a stage: piece p = wid + 4 i covers rows 8p .. 8p+7
  unsigned voffA[4], voffW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wid + 4 * i, r = 8 * p + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    voffA[i] = ((unsigned)r * (unsigned)P.lda + (unsigned)c * 8u) * 2u;
    voffW[i] = ((unsigned)r * (unsigned)P.ldb + (unsigned)c * 8u) * 2u;
  }
  // fragment read address inside a stage: row li (+16 per m-tile by immediate), chunk (g + 4 sub) ^ f(li)
  const int g = lane >> 4, li = lane & 15, f = (li >> 1) & 7;
  const unsigned fa0 = (unsigned)(li * 128 + (((g ^ (f & 3)) | (((f >> 2) & 1) << 2)) << 4));  // sub = 1: ^ 64

  auto issue = [&](const char* base, const unsigned (&voff)[4], int slot) {
    // the uniform base is made opaque so that base + lane offset stays "SGPR pair + 32-bit VGPR offset" in the DMA
    // instruction itself: folded into per-lane 64-bit addresses outside the loop it costs 8 registers per piece (spills)
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + voff[i]), (lptr_t)(smem + slot * WR_STAGE + 1024 * (wid + 4 * i)), 16, 0, 0);
  };

  // ---- W -> registers, through LDS in full lines -----------------------------------------------------------------
  bf16x8 wf[2][12];
  {
    const char* wb = (const char*)(P.W + (size_t)n0 * P.ldb);
#pragma unroll
    for (int s = 0; s < 4; ++s) issue(wb + 128 * s, voffW, s);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned wa = sbase + fa0 + (unsigned)(wid * 32 * 128);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 a, b, c, d;
      rd2(wa + s * WR_STAGE, a, b);
      rd2((wa + s * WR_STAGE) ^ 64u, c, d);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
      wf[0][2 * s] = as_frag(a); wf[1][2 * s] = as_frag(b); wf[0][2 * s + 1] = as_frag(c); wf[1][2 * s + 1] = as_frag(d);
    }
    asm volatile("s_barrier" ::: "memory");
#pragma unroll
    for (int s = 4; s < 6; ++s) issue(wb + 128 * s, voffW, s - 4);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int s = 4; s < 6; ++s) {
      u32x4 a, b, c, d;
      rd2(wa + (s - 4) * WR_STAGE, a, b);
      rd2((wa + (s - 4) * WR_STAGE) ^ 64u, c, d);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
      wf[0][2 * s] = as_frag(a); wf[1][2 * s] = as_frag(b); wf[0][2 * s + 1] = as_frag(c); wf[1][2 * s + 1] = as_frag(d);
    }
    asm volatile("s_barrier" ::: "memory");
  }

  STAMP();
  // ---- A pipeline ----------------------------------------------------------------------------------------------
  const char* Ab = (const char*)P.A;
  const long long tile_bytes = (P.dbg & 1) ? 0 : (long long)128 * P.lda * 2;
  auto stage_src = [&](int t, int ks) { return Ab + (long long)t * tile_bytes + 128 * ks; };
  // prologue: stages 0..2 of the first tile
  issue(stage_src(t0, 0), voffA, 0);
  issue(stage_src(t0, 1), voffA, 1);
  issue(stage_src(t0, 2), voffA, 2);
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  // A fragments: a rolling queue of NS registers-quads; the fragment consumed at step q was requested at step q - LA.
  // Step q = 16 ks + 8 sub + mt: wait for its fragment (counted lgkmcnt: the LA - 1 younger reads stay in flight), two
  // MFMAs (the wave's two n-tiles), then request fragment q + LA into the slot step q - 1 has just freed.
  constexpr int NS = 12, LA = 11;
  u32x4 F[NS];
  const unsigned fa1 = fa0 ^ 64u;
  int par = 0;  // slot of (tile, ks) = (par + ks) & 3; 6 stages per tile: par flips between 0 and 2
  auto frag_addr = [&](int q) {  // q counted from this tile's first step; may run into the next tile (q >= 96)
    const int ks = q >> 4, sub = (q >> 3) & 1;
    return sbase + (unsigned)(((par + ks) & 3) * WR_STAGE) + (sub ? fa1 : fa0);
  };
#pragma unroll
  for (int q = 0; q < LA; ++q)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(F[q]) : "v"(frag_addr(q)), "n"(2048 * (q & 7)) : "memory");

  f32x4 acc[2][8];
  const float* bias = P.bias;
  bf16* const C = P.C;
  const int ldc = P.ldc;

#pragma unroll 1
  for (int t = t0; t < t1; ++t) {
    const bool first = (t == t0), last = (t + 1 == t1);
#pragma unroll
    for (int ks = 0; ks < WR_KS; ++ks) {
      // B(g): my pieces of stage g+1 have landed (stage g+2 and, just after an epilogue, its stores may still fly);
      // behind the barrier stage g+1 is complete and nobody reads stage g-1 any more
      if (P.dbg & 8) {
        // ablation: no wait, no barrier
      } else if (last && ks == 5) {
        // nothing left in flight
      } else if (last && ks == 4) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      } else if (!first && ks <= 1) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 + EST) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      }
      {  // stage g+3 -> the slot stage g-1 has left
        const int ks3 = ks + 3;
        if (P.dbg & 4) {
        } else if (ks3 < WR_KS) issue(stage_src(t, ks3), voffA, (par + ks3) & 3);
        else if (!last) issue(stage_src(t + 1, ks3 - WR_KS), voffA, (par + ks3) & 3);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int q = 16 * ks + j, sub = j >> 3, mt = j & 7;
        const int qn = q + LA;  // fragment requested now
        const unsigned ad = frag_addr(qn);
        if (ks == 0 && sub == 0) {  // first k-step of a tile: accumulators start from zero
          asm volatile("s_waitcnt lgkmcnt(%9)\n\t"
                       "v_mfma_f32_16x16x32_bf16 %0, %3, %5, 0\n\t"
                       "v_mfma_f32_16x16x32_bf16 %1, %4, %5, 0\n\t"
                       "ds_read_b128 %2, %6 offset:%7"
                       : "=&v"(acc[0][mt]), "=&v"(acc[1][mt]), "=&v"(F[(q + NS - 1) % NS])
                       : "v"(wf[0][0]), "v"(wf[1][0]), "v"(F[q % NS]), "v"(ad), "n"(2048 * (qn & 7)), "n"(0), "n"(LA - 1)
                       : "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(%9)\n\t"
                       "v_mfma_f32_16x16x32_bf16 %0, %3, %5, %0\n\t"
                       "v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t"
                       "ds_read_b128 %2, %6 offset:%7"
                       : "+v"(acc[0][mt]), "+v"(acc[1][mt]), "=&v"(F[(q + NS - 1) % NS])
                       : "v"(wf[0][2 * ks + sub]), "v"(wf[1][2 * ks + sub]), "v"(F[q % NS]), "v"(ad), "n"(2048 * (qn & 7)), "n"(0), "n"(LA - 1)
                       : "memory");
        }
      }
    }
    par ^= 2;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMAs' results before the VALU reads them
    // ---- epilogue: bias, pair the two n-tiles, 16-byte stores -------------------------------------------------
    {
      const int m0 = t * 128;
      int ln = lane;
      asm volatile("" : "+v"(ln));  // store addresses are recomputed per tile (hoisted, they are 8 more live registers)
      const int g = ln >> 4, li = ln & 15;
      if (bias) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4 b4 = *(const f32x4*)(bias + n0 + 32 * wid + 16 * nt + 4 * g);
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) acc[nt][mt] += b4;
        }
      }
      const int ncol = n0 + 32 * wid + ((g & 1) << 4) + ((g & 2) << 2);
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const f32x4 te = acc[0][mt], to = acc[1][mt];
        f32x4 lo, hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(te[r]), __float_as_uint(to[r]), false, false);
          lo[r] = __uint_as_float(sw[0]);
          hi[r] = __uint_as_float(sw[1]);
        }
        bf16x8 o;
        const int m = m0 + 16 * mt + li;
        if (P.dbg & 16) {  // GELU + second output (gelu')
          bf16x8 o2;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float ga, gd;
            vg_gelu_both(lo[r], ga, gd); o[r] = vg_f2bf(ga); o2[r] = vg_f2bf(gd);
            vg_gelu_both(hi[r], ga, gd); o[r + 4] = vg_f2bf(ga); o2[r + 4] = vg_f2bf(gd);
          }
          *(bf16x8*)(P.C2 + (unsigned)(m * ldc + ncol)) = o2;
          *(bf16x8*)(C + (unsigned)(m * ldc + ncol)) = o;
          continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[r] = vg_f2bf(lo[r]); o[r + 4] = vg_f2bf(hi[r]); }
        if (!(P.dbg & 2) || o[0] == (bf16)123.f) *(bf16x8*)(C + (unsigned)(m * ldc + ncol)) = o;
      }
    }
    if (sti < 13) STAMP();
  }
  if (stp) { stp[13] = (long long)__builtin_amdgcn_s_memtime(); stp[14] = (long long)__builtin_amdgcn_s_memrealtime(); }
}
__global__ void ref_kernel(const bf16* A, const bf16* W, const float* bias, float* C, int M, int N, int K) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float s = bias ? bias[n] : 0.f;
  for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * K + k] * (float)W[(size_t)n * K + k];
  C[i] = s;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 33280, N = argc > 2 ? atoi(argv[2]) : 1152, K = 384;
  const int wgs = argc > 3 ? atoi(argv[3]) : 512;
  std::vector<bf16> hA((size_t)M * K), hW((size_t)N * K);
  std::vector<float> hb(N);
  srand(1);
  for (auto& v : hA) v = (bf16)((rand() % 2001 - 1000) / 1000.f);
  for (auto& v : hW) v = (bf16)((rand() % 2001 - 1000) / 8000.f);
  for (auto& v : hb) v = (rand() % 2001 - 1000) / 1000.f;
  bf16 *A, *W, *C; float *b, *Cr;
  hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&b, N * 4); hipMalloc(&Cr, (size_t)M * N * 4);
  hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
  hipMemset(C, 0, (size_t)M * N * 2);
  bf16* C2; hipMalloc(&C2, (size_t)M * N * 2);
  WrArgs P{A, W, b, C, C2, M, N, K, K, N, M / 128, N / 128, (wgs / 8) / (N / 128), nullptr, 0};
  if (M % 128 || N % 128) { printf("M, N must be multiples of 128\n"); return 1; }
  printf("M %d N %d: n_tiles %d, m-groups %d (%.2f tiles each), %d workgroups\n", M, N, P.n_tiles, 8 * P.gpx, (double)P.tiles_m / (8 * P.gpx), wgs);
  hipLaunchKernelGGL(wr_kernel<8>, dim3(wgs), dim3(256), 0, 0, P);
  hipLaunchKernelGGL(ref_kernel, dim3((unsigned)(((long long)M * N + 255) / 256)), dim3(256), 0, 0, A, W, b, Cr, M, N, K);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<bf16> hC((size_t)M * N); std::vector<float> hR((size_t)M * N);
  hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hR.data(), Cr, hR.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0; long long bad = 0;
  for (size_t i = 0; i < hC.size(); ++i) {
    const double e = fabs((double)(float)hC[i] - hR[i]);
    maxerr = e > maxerr ? e : maxerr; maxref = fabs(hR[i]) > maxref ? fabs(hR[i]) : maxref;
    if (e > 0.02 * (1.0 + fabs(hR[i]))) { if (bad < 5) printf("  bad at m %zu n %zu: got %f want %f\n", i / N, i % N, (float)hC[i], hR[i]); ++bad; }
  }
  printf("check: max |err| %.4g (max |ref| %.3g), %lld bad\n", maxerr, maxref, bad);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int dbg : {0, 16, 14}) {
  P.dbg = dbg;
  printf("dbg %d (1 = A always tile 0, 2 = no stores, 4 = no DMA in the loop, 8 = no vmcnt wait / barrier)\n", dbg);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) { if (dbg & 16) hipLaunchKernelGGL(wr_kernel<16>, dim3(wgs), dim3(256), 0, 0, P); else hipLaunchKernelGGL(wr_kernel<8>, dim3(wgs), dim3(256), 0, 0, P); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("wr_kernel: %.1f us  %.0f TFLOP/s\n", ms / 20 * 1e3, 2.0 * M * N * K / (ms / 20 * 1e-3) / 1e12);
  }
  }
  {
    long long* st; hipMalloc(&st, (size_t)wgs * 16 * 8); hipMemset(st, 0, (size_t)wgs * 16 * 8);
    P.dbg = 0; P.stamps = st;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(wr_kernel<8>, dim3(wgs), dim3(256), 0, 0, P);
    hipDeviceSynchronize();
    std::vector<long long> h((size_t)wgs * 16); hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    // per workgroup: [0] start, [1] after W prologue, [2..] after each tile, [13] end, [14]/[15] realtime end/start (100 MHz)
    double clk = 0, pro = 0, tile1 = 0, tilen = 0, life = 0; int n = 0, nt = 0;
    long long first_rt = -1, last_rt = 0;
    for (int w = 0; w < wgs; ++w) {
      const long long* x = &h[(size_t)w * 16];
      if (!x[13]) continue;
      ++n;
      const double cyc = (double)(x[13] - x[0]), rt = (double)(x[14] - x[15]) * 10.0;  // ns
      clk += cyc / rt; pro += (double)(x[1] - x[0]); life += cyc;
      tile1 += (double)(x[2] - x[1]);
      for (int i = 3; i < 13 && x[i]; ++i) { tilen += (double)(x[i] - x[i - 1]); ++nt; }
      if (first_rt < 0 || x[15] < first_rt) first_rt = x[15];
      if (x[14] > last_rt) last_rt = x[14];
    }
    printf("stamps over %d workgroups: clock %.2f GHz, W prologue %.0f cyc, first tile %.0f cyc, later tiles %.0f cyc (MFMA floor 3072 alone / 6144 shared), lifetime %.0f cyc, span %.1f us\n",
           n, clk / n, pro / n, tile1 / n, nt ? tilen / nt : 0.0, life / n, (double)(last_rt - first_rt) / 100.0);
  }
  return bad != 0;
}
