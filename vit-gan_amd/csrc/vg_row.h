// Full-row GEMM (gemm_row.hip): C[m, 0:384] = A[m, 0:K] Wp, with the LayerNorm that follows (forward) or precedes
// (backward) the Linear computed in the epilogue - a workgroup owns whole rows, so the row statistics never leave the CU.
#pragma once
#include "vg_common.h"

#define VG_ROW_N 384  // default output width of a full-row problem (the embedding width of the C1-C3 configurations); 512 (C4) since round 4
int vg_row_width_ok(int N);  // widths the kernel is built for: 384, 512

enum { VG_ROW_LNFWD = 0, VG_ROW_LNBWD = 1, VG_ROW_LNBWD_PEN = 2 };  // 2: LNBWD with the gradient penalty's two extra operands (plain LayerNorm only; separate instantiation)

struct VgRowArgs {
  const bf16* A; int lda;  // [M, K] row-major
  const bf16* Wp;          // packed weights: [K/32][N][32] stage images (vg_pack_rows_launch)
  int M, K;                // M % 16 == 0, K % 64 == 0
  int N;                   // output width = row length of every [M, N] operand below: 384 or 512 (0 = 384)
  int units, nwg;          // 16-row units of A; workgroups (filled by the launcher)
  int dbg;                 // diagnostic builds only (VG_TUNING)
  // ---- VG_ROW_LNFWD:  y = res + drop(A W^T + bias);  yn = LN(y) * gamma + beta ------------------------------------
  const float* bias;       // [384] (nullable)
  const bf16* res;         // [M, 384] (nullable)
  long long ldr;           // row stride of res in elements (0 = 384): e.g. S * 384 when the residual is the CLS rows of a [B*S, 384] tensor
  bf16* Y;                 // [M, 384]
  bf16* Yn;                // [M, 384] normalised rows; nullptr: no LayerNorm follows (Y only)
  float* mean_out; float* rstd_out;  // [M] statistics of Y (written when Yn)
  const float* beta;       // [384]
  float eps;
  // ---- VG_ROW_LNBWD:  dx = gres + LN'(A W) ;  dxm = dx * mask -----------------------------------------------------
  const bf16* x;           // [M, 384] the LayerNorm's input
  const float* mean; const float* rstd;  // [M]
  const bf16* gres;        // [M, 384] gradient arriving over the residual connection (nullable)
  bf16* dx; bf16* dxm;     // [M, 384]; dxm nullable
  const bf16* gres2;       // VG_ROW_LNBWD_PEN: [M, 384] a second gradient added to dx (what the double backward injected at the LayerNorm's input; nullable)
  bf16* dy_out;            // VG_ROW_LNBWD_PEN: [M, 384] receives A W itself, the LayerNorm backward's dY (the double backward's operand; nullable)
  float* part;             // [nwg][3*384]: per workgroup column sums  d gamma | d beta | colsum(dxm ? dxm : dx); nullptr (plain LayerNorm
                           // only): no sums at all - a backward that wants the input gradient alone (the generator's pass through D)
  // ---- self-modulated LayerNorm (v1 generator, src/v1/spectral_layer_norm.py:19-20): wmod != nullptr -------------------------
  //      forward:  yn = w * (gs * (LN(y) * gamma + beta) + bs);   backward: dy_eff = dy * w * gs feeds the LayerNorm backward,
  //      dw_acc (+)= dy * (gs * (xhat * gamma + lbias) + bs), and the partial row gets d gs, d bs at [3*384], [3*384 + 1]
  const bf16* wmod;        // [M, 384] modulation rows
  const float* gs; const float* bs;  // device scalars
  const float* lbias;      // LNBWD: [384] the LayerNorm's bias
  float* dw_acc; int dw_accumulate;  // LNBWD: fp32 [M, 384]
  const float* resf; int res_period; // LNFWD: residual from an fp32 table [res_period, 384] indexed by row % res_period (instead of res)
  int x_period;            // LNBWD: x has x_period rows, broadcast over the batch (0: one row of x per row of A)
  int part_w;              // LNBWD: row stride of part (filled by the launcher: 3*384, or 3*384 + 64 with wmod)
  // ---- both -------------------------------------------------------------------------------------------------------
  const float* gamma;      // [384]
  unsigned drop_thresh, drop_key; float drop_scale; const unsigned* drop_step;  // LNFWD: mask of drop(.); LNBWD: mask of dxm
  int drop_row_mul;        // > 1: row m of this problem draws the mask bits of row m * drop_row_mul of a larger tensor (the CLS rows of [B*S, 384])
};

// number of workgroups (= rows of `part`) a problem of M rows is run with; 0 when the kernel does not take it
int vg_row_nwg(int M);
// 1 = enqueued, 0 = not of this kernel's kind, < 0 = -hipError
int vg_gemm_row_launch(VgRowArgs a, int epi, hipStream_t st);

// Pack up to 4 weight matrices per block, for `nblocks` blocks laid out at a fixed stride, into stage images:
//   dst[(s * 384 + n) * 32 + ...] <- transposed ? src[(32 s + k) * ld + n] : src[n * ld + 32 s + k]
struct VgPackDesc { long long src_off, dst_off; int K, ld, transposed; };
struct VgPackJobs {
  const bf16* src; bf16* dst;
  long long src_stride, dst_stride;  // elements between blocks
  int nblocks, n;
  int N;                             // output width of the packed problems (rows of a stage image): 384 or 512 (0 = 384)
  VgPackDesc d[4];
};
int vg_pack_rows_launch(const VgPackJobs& jobs, hipStream_t st);
