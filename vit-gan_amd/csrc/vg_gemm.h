// Host-side description of one GEMM problem for vg_gemm_launch (internal C++ API).
#pragma once
#include "vg_common.h"

// Operand forms.  All matrices are bf16, row-major.
//   VG_NT : C[m,n] = sum_k A[m,k] * B[n,k]     forward  (x @ W^T, W stored [out,in])
//   VG_NN : C[m,n] = sum_k A[m,k] * B[k,n]     dgrad    (dy @ W,   W stored [out,in])
//   VG_TN : C[m,n] = sum_k A[k,m] * B[k,n]     wgrad    (dy^T @ x), split over k, fp32 slabs
enum { VG_NT = 0, VG_NN = 1, VG_TN = 2 };

enum {
  VG_ACT_NONE = 0,
  VG_ACT_GELU = 1,       // exact erf GELU
  VG_ACT_SIN = 2,        // sin(act_scale * v)         (SIREN)
  VG_ACT_TANH = 3,
  VG_ACT_MUL_GELU_GRAD = 4,  // v *= gelu'(Z[m,n])       (fc2 dgrad epilogue)
  VG_ACT_MUL_COS = 5,        // v *= act_scale*cos(act_scale*Zf[m,n])  (SIREN dgrad epilogue)
  VG_ACT_MUL_TANH_GRAD = 6,  // v *= 1 - Z[m,n]^2, Z = tanh output  (classifier fc2 dgrad epilogue)
  VG_ACT_MUL_Z = 7,          // v *= Z[m,n]: Z = a derivative stored by the forward (see c2_gelu_grad)
  VG_ACT_MUL_Z8 = 8,         // v *= decode(Z8[m,n]): the derivative as one byte per element (c2_gelu_grad = 2; Z points at bytes, ldz in bytes)
};

struct VgGemmProb {
  const bf16* A; const bf16* B;
  int lda, ldb;
  int M, N, K;
  int splits;                  // TN only: number of K slices (slab s written at Cf + s*cf_split_stride)
  bf16* C; int ldc;            // bf16 result (nullable)
  float* Cf; int ldcf;         // fp32 result: TN slabs, or fp32 pre-activation copy when pre_f32
  long long cf_split_stride;
  int cf_accumulate;           // TN with ONE K slice: Cf += result (the gradient buffer itself; no slab, no fold pass)
  bf16* C2; int ldc2;          // bf16 pre-activation copy (nullable)
  int c2_gelu_grad;            // with VG_ACT_GELU: C2 receives gelu'(pre-activation) instead of the pre-activation itself -
                               // all the backward needs of it, and the fc2 dgrad epilogue becomes one multiply (VG_ACT_MUL_Z).
                               // 2: the derivative as ONE BYTE per element (vg_g8_pack4, vg_common.h): C2 points at bytes, ldc2 in bytes,
                               // ldc2 % 8 == 0; read back by VG_ACT_MUL_Z8
  const float* bias;           // [N] fp32 (nullable)
  const bf16* res; int ldr;    // residual added after the activation (nullable)
  const float* resf; int res_period;  // fp32 addend table [res_period, N] indexed by m % res_period
  const bf16* Z; int ldz;      // VG_ACT_MUL_GELU_GRAD input
  const float* Zf; int ldzf;   // VG_ACT_MUL_COS input
  int act; float act_scale;
  int pre_f32;                 // store bias-added pre-activation to Cf (fp32) when set (NT only)
  int row_in_per, row_out_per, row_out_off;  // output row remap: (m/in)*out + off + m%in  (0 = none)
  // dropout on the output (NT only): drop_thresh = round(p*256) (0 = off); applied after bias/activation and BEFORE
  // the residual (drop_post = 0: x + drop(y)) or after every addend (drop_post = 1: drop(y + pos)); index = row*N + col
  unsigned drop_thresh, drop_key; float drop_scale; int drop_post; const unsigned* drop_step;
  int drop_row_mul;            // > 1: the mask index is (row * drop_row_mul) * N + col - a compact [B, N] problem over rows 0, S, 2S, ..
                               // of a [B*S, N] tensor (the CLS rows) draws exactly the bits the full-size launch would have drawn there
  // TN only: column sums of A over k (= the bias gradient that goes with this weight gradient), one fp32 row [M] per
  // K slice at colsum + s*colsum_split_stride (nullable).  Computed on the MFMA pipe (ones x A fragments) by the
  // workgroups of the first n-tile: the wgrad kernel is L2->LDS bound, the extra MFMAs are free.
  float* colsum; long long colsum_split_stride;
  // filled by the launcher
  int tiles_m, tiles_n, tile_start, k_per_split;
};

#define VG_MAX_GROUP 8   // round 3: the weight gradients of TWO encoder blocks go out as one launch
struct VgGemmGroup {
  int n;
  int tpw, total;     // consecutive tiles per workgroup; number of tiles of the launch
  const void* zeros;  // 16 zero bytes in device memory: source of out-of-range LDS-DMA lanes
  VgGemmProb p[VG_MAX_GROUP];
};

// Enqueue 1..VG_MAX_GROUP problems of one operand form as a single launch.
int vg_gemm_launch(VgGemmProb* probs, int n, int mode, hipStream_t stream);
// gemm_wr.hip: 1 = enqueued on the weights-in-registers kernel, 0 = problem not of its kind, < 0 = -hipError.
int vg_gemm_wr_try(const VgGemmProb& p, int mode, hipStream_t stream);
// gemm_tn.hip: grouped weight gradients on 128 x 384 tiles; same return convention, may lower probs[i].splits.
int vg_gemm_tn384_try(VgGemmProb* probs, int n, hipStream_t stream);
// Convenience: zero-initialised problem.
static inline VgGemmProb vg_gemm_prob() { VgGemmProb p = {}; p.splits = 1; return p; }
