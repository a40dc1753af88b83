// Probe: LDS-DMA (global_load_lds_dwordx4) fill rate per CU as a function of the request shape, with the source L2-resident.
// Decides whether the GEMM's operand staging (16 rows x 64 B per 1-KiB piece at BK = 32) is what limits it.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_fill_probe.hip -o /tmp/lds_fill_probe && /tmp/lds_fill_probe
// Each workgroup (256 threads = 4 waves) fills a ring of NST stages x STAGE_KB KiB; a stage is 16*STAGE_KB/16 pieces of 1 KiB,
// dealt round-robin to the waves; before re-filling a stage the wave waits until its own pieces of that stage have landed
// (counted vmcnt), i.e. NST-1 stages stay in flight.  Source: a [ROWS][LDB] bf16 matrix per XCD-group, re-read by every
// workgroup (L2-resident after the first pass), walked along k like a GEMM operand panel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int SHAPE, int NST, int PPW>  // SHAPE: bytes per row segment (64, 128, 256, 1024); PPW: pieces per wave per stage
__global__ __launch_bounds__(256) void fill(const char* __restrict__ src, long long row_bytes, int rows, int steps, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = SHAPE / 16;          // lanes per row segment
  constexpr int RPP = 64 / LPR;            // rows per piece
  constexpr int STAGE = 4 * PPW * 1024;    // bytes per stage
  // this workgroup's panel: rows [r0, r0 + 4*PPW*RPP)
  const int panel_rows = 4 * PPW * RPP;
  const int r0 = (blockIdx.x * panel_rows) % (rows - panel_rows + 1);
  unsigned voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = wid + 4 * i;
    const int row = r0 + piece * RPP + lane / LPR;
    voff[i] = (unsigned)(row * row_bytes + (lane % LPR) * 16);
  }
  const char* base = src;
  const int ksteps_per_row = (int)(row_bytes / SHAPE);
  for (int s = 0; s < steps; ++s) {
    if (s >= NST) {  // wait for this wave's pieces of the stage about to be overwritten (issued NST steps ago)
      if (NST == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * PPW) : "memory");
    }
    unsigned char* st = ring + (s % NST) * STAGE;
    const char* b = base + (long long)(s % ksteps_per_row) * SHAPE;
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(b + voff[i]), (lptr_t)(st + 1024 * (wid + 4 * i)), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = ((unsigned*)ring)[lane];
}

template <int SHAPE, int NST, int PPW>
void run(const char* name, const char* src, long long row_bytes, int rows, unsigned* sink, int wg_per_cu) {
  const int steps = 4000;
  const size_t lds = (size_t)NST * 4 * PPW * 1024;
  hipFuncSetAttribute((const void*)fill<SHAPE, NST, PPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  hipLaunchKernelGGL((fill<SHAPE, NST, PPW>), dim3(grid), dim3(256), lds, 0, src, row_bytes, rows, 200, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((fill<SHAPE, NST, PPW>), dim3(grid), dim3(256), lds, 0, src, row_bytes, rows, steps, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * steps * 4 * PPW * 1024;
  printf("%-34s seg %4d B  ring %d x %2d KiB  %d WG/CU (in flight %3d KiB/CU): %7.1f GB/s/CU  %6.2f TB/s chip\n", name, SHAPE, NST, 4 * PPW,
         wg_per_cu, (int)((NST - 1 > 0 ? NST - 1 : 1) * 4 * PPW * wg_per_cu), bytes / ms / 1e6 / 256, bytes / ms / 1e9);
}

int main() {
  const int rows = 2560; const long long row_bytes = 768;   // 1.9 MB: resident in every XCD's 4-MiB L2 whatever the shape (a bigger matrix made the 64-B case an HBM test)
  char* src; unsigned* sink;
  hipMalloc(&src, (size_t)rows * row_bytes + 4096); hipMemset(src, 1, (size_t)rows * row_bytes + 4096); hipMalloc(&sink, 4096 * 4);
  // GEMM today: WM=4 ring = 3 stages x 24 KiB (A 16 + B 8), 2 WG/CU; model A+B as one 24-KiB stage (PPW = 6)
  run<64, 3, 6>("16 rows x 64 B  (BK=32 row form)", src, row_bytes, rows, sink, 2);
  run<128, 3, 6>("8 rows x 128 B  (BK=64 row form)", src, row_bytes, rows, sink, 2);
  run<256, 3, 6>("4 rows x 256 B  (tr form)", src, row_bytes, rows, sink, 2);
  run<64, 2, 4>("16 x 64 B, 2 x 16 KiB ring", src, row_bytes, rows, sink, 4);
  run<128, 2, 4>("8 x 128 B, 2 x 16 KiB ring", src, row_bytes, rows, sink, 4);
  run<64, 3, 6>("16 x 64 B, 1 WG/CU", src, row_bytes, rows, sink, 1);
  run<128, 3, 6>("8 x 128 B, 1 WG/CU", src, row_bytes, rows, sink, 1);
  run<64, 4, 8>("16 x 64 B, 4 x 32 KiB ring, 1 WG", src, row_bytes, rows, sink, 1);
  run<128, 4, 8>("8 x 128 B, 4 x 32 KiB ring, 1 WG", src, row_bytes, rows, sink, 1);
  run<256, 4, 8>("4 x 256 B, 4 x 32 KiB ring, 1 WG", src, row_bytes, rows, sink, 1);
  return 0;
}
