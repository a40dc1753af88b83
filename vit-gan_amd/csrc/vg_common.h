// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the ViTGAN hot path.
// Written for wave64 + MFMA 16x16x32 bf16 only; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define VG_WAVE 64

#define VG_CHECK_HIP(expr)                                   \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) return (int)_e;                    \
  } while (0)

__device__ __forceinline__ float vg_bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 vg_f2bf(float v) { return (bf16)v; }

// lane <-> MFMA 16x16x32 bf16 fragment coordinates (cdna_hip_programming.md s3):
//   A[row = lane&15][k = 8*(lane>>4) + j],  B[k = 8*(lane>>4) + j][col = lane&15],
//   C/D: col = lane&15, row = 4*(lane>>4) + reg.
__device__ __forceinline__ f32x4 vg_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Transposed LDS read (ds_read_b64_tr_b16): per 16-lane group a 4x16 block of 16-bit elements,
// lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives column i, rows 0..3.
__device__ __forceinline__ bf16x4 vg_lds_tr_read(const void* lds_ptr) {
  typedef bf16x4 __attribute__((address_space(3))) * lds_bf4_ptr;
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(uintptr_t)(uint32_t)(uintptr_t)lds_ptr);
}

__device__ __forceinline__ float vg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float vg_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Exact-erf GELU (nn.GELU default) and its derivative from ONE evaluation of Phi(x) = (1 + erf(x / sqrt 2)) / 2 and
// e = exp(-x^2 / 2).  erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, far below one bf16 ulp): one v_exp, one v_rcp,
// a 5-term Horner chain.  The GELU epilogue of the fc1 GEMM is VALU-bound, so the instruction count is the cost: the
// reciprocal is the bare v_rcp_f32 (1 ulp; __frcp_rn expands to the 10-instruction IEEE division sequence), the exponential
// the bare v_exp_f32 on -(k x)^2 with k^2 = log2(e) / 2, Phi one fma of the signed erf: 15 VALU instructions for both outputs.
__device__ __forceinline__ void vg_phi_e(float x, float& phi, float& e) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.3275911f * 0.70710678118654752f, 1.0f));
  const float y = x * 0.84932180028801907f;  // sqrt(log2(e) / 2)
  e = __builtin_amdgcn_exp2f(-(y * y));      // exp(-x^2 / 2)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float er = copysignf(fmaf(-poly, e, 1.0f), x);  // erf(x / sqrt 2)
  phi = fmaf(er, 0.5f, 0.5f);
}
__device__ __forceinline__ float vg_gelu(float x) {
  float phi, e;
  vg_phi_e(x, phi, e);
  return x * phi;
}
__device__ __forceinline__ float vg_gelu_grad(float x) {  // Phi(x) + x * phi(x)
  float phi, e;
  vg_phi_e(x, phi, e);
  return fmaf(x * 0.39894228040143268f, e, phi);
}
__device__ __forceinline__ void vg_gelu_both(float x, float& g, float& dg) {
  float phi, e;
  vg_phi_e(x, phi, e);
  g = x * phi;
  dg = fmaf(x * 0.39894228040143268f, e, phi);
}
// gelu'(x) lies in [-0.1290, 1.1290] (extrema at x = -/+ sqrt 2).  The forward keeps it as ONE BYTE per element on a grid of
// 1/200 that contains 0 and 1 exactly (code 27 <-> 0, code 227 <-> 1; the saturated derivatives of strongly negative / positive
// pre-activations stay exact): |error| <= 0.0025, the same size as bf16's rounding of a value in [0.5, 1.13) (0.002-0.004) and
// 1/4 of the bytes of keeping both gelu(.) and gelu'(.) in bf16 for the fc1 -> fc2 backward.
#define VG_G8_ZERO 27.0f
#define VG_G8_STEP 0.005f
__device__ __forceinline__ uint32_t vg_g8_pack4(float a, float b, float c, float d) {  // 4 derivatives -> 4 codes, element 0 in byte 0
  uint32_t w = 0;
  w = __builtin_amdgcn_cvt_pk_u8_f32(rintf(fmaf(a, 200.0f, VG_G8_ZERO)), 0, w);  // exact integers in: the conversion only saturates
  w = __builtin_amdgcn_cvt_pk_u8_f32(rintf(fmaf(b, 200.0f, VG_G8_ZERO)), 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(rintf(fmaf(c, 200.0f, VG_G8_ZERO)), 2, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(rintf(fmaf(d, 200.0f, VG_G8_ZERO)), 3, w);
  return w;
}
__device__ __forceinline__ float vg_g8_value(uint32_t w, int byte) {  // byte: compile-time constant after unrolling (v_cvt_f32_ubyteN)
  return ((float)((w >> (8 * byte)) & 0xFFu) - VG_G8_ZERO) * VG_G8_STEP;
}
__device__ __forceinline__ float vg_tanh(float x) {  // 1 - 2/(exp(2x)+1), saturates cleanly for |x| large
  const float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// Counter-based dropout: element e of a dropped tensor (row-major, linear index e) is kept iff byte (e & 3)
// of vg_drop_word(key, e >> 2) is >= thresh (thresh = round(p * 256), so p is quantised to 1/256 and the
// survivors are scaled by 256 / (256 - thresh)).  Stateless: forward epilogues and backward kernels
// regenerate the identical mask from (key, index); key = host hash of (seed, dropout site).
__device__ __forceinline__ uint32_t vg_drop_word(uint32_t key, uint32_t idx4) {
  uint32_t x = idx4 * 0x9E3779B1u + key;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
// key actually used by a launch: host key mixed with a device-resident step counter (so a replayed hipGraph
// draws a fresh mask every step); dstep == nullptr leaves the host key unchanged
__device__ __forceinline__ uint32_t vg_drop_key(uint32_t key, const unsigned* __restrict__ dstep) {
  return dstep ? key ^ (dstep[0] * 0x9E3779B1u + 0x7F4A7C15u) : key;
}
__device__ __forceinline__ float vg_drop_factor(uint32_t word, int e, uint32_t thresh, float scale) {
  return (((word >> (8 * (e & 3))) & 0xFFu) >= thresh) ? scale : 0.f;
}
