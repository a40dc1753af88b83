// Probe: what v_cvt_pk_fp8_f32 does with e4m3 subnormals / saturation, and whether the fp8 MFMA honours subnormal operands.
// Build and run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/micro/fp8_probe.hip -o /tmp/fp8_probe && /tmp/fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void probe(const float* in, unsigned* bytes, float* mm, int n) {
  const int l = threadIdx.x;
  if (l < n) { int w = 0; w = __builtin_amdgcn_cvt_pk_fp8_f32(in[l], 0.f, w, false); bytes[l] = (unsigned)w & 0xFF; }
  // MFMA: every A element = byte pattern `pat`, every B element = 1.0 (0x38): D[i][j] = 32 * value(pat)
  for (int t = 0; t < 4; ++t) {
    const unsigned pat = (t == 0) ? 0x01 : (t == 1) ? 0x04 : (t == 2) ? 0x08 : 0x38;  // 2^-9, 2^-7, 2^-6 (min normal), 1.0
    unsigned long a = 0, b = 0;
    for (int j = 0; j < 8; ++j) { a |= (unsigned long)pat << (8 * j); b |= 0x38ul << (8 * j); }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a, (long)b, c, 0, 0, 0);
    if (l == 0) mm[t] = c[0];
  }
}
int main() {
  const int n = 12;
  float h[n] = {0.001f, 0.00195312f, 0.003f, 0.0039f, 0.0078125f, 0.012f, 0.015625f, 0.02f, 1.0f, 448.f, 500.f, 1e6f};
  float* d; unsigned* b; float* mm;
  hipMalloc(&d, sizeof(h)); hipMalloc(&b, n * 4); hipMalloc(&mm, 16);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, b, mm, n);
  unsigned hb[n]; float hm[4];
  hipMemcpy(hb, b, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hm, mm, 16, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("cvt %g -> 0x%02x\n", h[i], hb[i]);
  printf("mfma 32*2^-9 = %g (want 0.0625), 32*2^-7 = %g (want 0.25), 32*2^-6 = %g (want 0.5), 32*1 = %g\n", hm[0], hm[1], hm[2], hm[3]);
  return 0;
}
