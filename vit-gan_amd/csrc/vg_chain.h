// Row-chain kernels (chain.hip): several Linears of an encoder block in ONE launch, the activations between them never
// leaving the registers of the wave that owns their rows.
//
// A wave owns 16 rows (one MFMA m-tile) of the token matrix for the whole chain and keeps them as MFMA fragments; the only
// thing that moves is the weights, which stream through LDS once per workgroup as a "chain image": the stages of every
// Linear of the chain in the order the kernel consumes them, each stage 24 fragments x 64 lanes x 16 B already in register
// layout (a stage is one contiguous 24-KiB LDS-DMA, a fragment read is lane * 16 - no swizzle, no conflicts).
#pragma once
#include "vg_common.h"

#define VG_CH_E 384           // embedding width of the chains (C1-C3)
#define VG_CH_HID 768         // MLP hidden width (mlp_ratio 2)
#define VG_CH_STAGE 24576     // bytes of one stage image
#define VG_CH_MLP_STAGES 48   // fc1 (24 stages of [64 n][192 k]) + fc2 (24 stages of [384 n][32 k])
#define VG_CH_FRONT_STAGES 12 // the out-projection in front of them (12 stages of [384 n][32 k])

// k order inside a 32-deep MFMA k-step, position p = 8 g + i of lane group g (what the A fragment holds there):
//   NAT  k = 8 g + i                          A loaded from global memory, 16 B per lane
//   P    k = 4 g + i (i < 4), 16 + 4 g + i - 4  A = two neighbouring accumulator tiles of the producing Linear, packed in place
//   S    k = 16 (g & 1) + 8 (g >> 1) + i       A = the 8 consecutive columns a lane holds behind the permlane swap of the row epilogue
enum { VG_CH_KNAT = 0, VG_CH_KP = 1, VG_CH_KS = 2 };

struct VgChainMlpArgs {
  const bf16* xn; int ldx;         // [M, 384] input of fc1 (norm2's output)
  const bf16* img;                 // chain image (vg_chain_mlp_pack_launch)
  const float* b1; const float* b2;
  const bf16* res;                 // [M, 384] residual (x_mid), nullable
  bf16* a1; unsigned char* z8;     // [M, 768] gelu(fc1), and gelu'(fc1 pre-activation) as byte codes (vg_g8_pack4)
  bf16* Y;                         // [M, 384] res + drop(fc2(a1))
  bf16* Yn; float* mean_out; float* rstd_out; const float* gamma; const float* beta; float eps;  // LayerNorm of Y (Yn nullable)
  int M, units, upw;               // units = M / 16; upw: units (active waves) per workgroup tile (both filled by the launcher)
  unsigned drop_thresh, drop_key; float drop_scale; const unsigned* drop_step; int drop_row_mul;
  // ---- front: x_mid = xin + drop_a(ao Wo^T + bo);  xn_out = LayerNorm2(x_mid) feeds fc1 (ao != nullptr selects it; xn is then unused)
  const bf16* ao; const bf16* xin; const float* bo; const float* gamma2; const float* beta2;
  bf16* xmid; bf16* xn_out; float* mean2; float* rstd2; unsigned drop_key_a;
  unsigned long long* stamps;      // diagnostic builds only (CH_DBG & 64): [workgroup][wave][2 * stages + 2] s_memtime stamps
};

// chain image of an encoder block's MLP, forward: W1 [768, 384] and W2 [384, 768] (nn.Linear weights, row-major bf16)
// -> VG_CH_MLP_STAGES stage images.  k_in: fragment order of fc1's A operand (VG_CH_KNAT when xn is read from memory).
int vg_chain_mlp_pack_launch(const bf16* W1, const bf16* W2, bf16* img, int k_in, hipStream_t st);
// the same with the out-projection's 12 stages in front (Wo [384, 384]); fc1's A operand is then in order S
int vg_chain_block_pack_launch(const bf16* Wo, const bf16* W1, const bf16* W2, bf16* img, hipStream_t st);
// 1 = enqueued, 0 = not of this kernel's kind (M % 16), < 0 = -hipError
int vg_chain_mlp_fwd_launch(const VgChainMlpArgs& a, hipStream_t st);
