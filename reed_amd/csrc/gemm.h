// Internal GEMM interface (see gemm.hip). Public C ABI is in include/reed_hip.h.
#pragma once
#include "common.hpp"

enum { LAY_NT = 0, LAY_NN = 1, LAY_TN = 2,
       LAY_TN_TALL = 3, LAY_TN_WIDE = 4 };   // TN on gemm_tn.hip's 256x128 / 128x256 tile (fp32 epilogue only)
enum {
  EPI_BF16 = 0,       // C bf16 = bf16(acc [+bias])
  EPI_GELU = 1,       // C bf16 = pre = bf16(acc+bias) (optional), C2 bf16 = gelu_tanh(pre)
  EPI_SILU = 2,       // same with SiLU
  EPI_GATE_RES = 3,   // C f32 = R f32 + bf16(gate[m/rows_per_gate] * bf16(acc+bias)); C2 bf16 = y (optional)
  EPI_DGELU = 4,      // C bf16 = bf16(bf16(acc) * gelu_tanh'(R bf16))
  EPI_DSILU = 5,      // C bf16 = bf16(bf16(acc) * silu'(R bf16))
  EPI_F32 = 6,        // C f32 (+)= acc [+bias]; split-K writes slabs C + z*slab_stride
  EPI_ADDF32_RB = 7,  // C f32 += float(bf16(acc))
  EPI_ATOMIC_F32 = 8, // atomicAdd(C f32, acc)   (split-K into a pre-zeroed / accumulating buffer)
  // inference-only epilogues of the frozen CLIP image encoder (bf16 residual stream):
  EPI_QGELU = 9,      // C2 bf16 = QuickGELU(pre) = bf16(pre * bf16(sigmoid(bf16(1.702 pre)))), pre = bf16(acc+bias) (C optional)
  EPI_RES_BF16 = 10,  // C bf16 = bf16(bf16(acc+bias) + R bf16)
  EPI_LS_RES = 12,    // C f32 = R f32 + gamma f32[n] * float(bf16(acc+bias)): LayerScale + fp32 residual (DINOv2 blocks); `gate`
                      //   points at the fp32 gamma vector; NT only
  EPI_BF16_DOT = 13,  // C bf16 = bf16(acc [+ bias]) and dpart f32 [N / hd, S, M] (C2) = per row and head of hd = rows_per_gate columns
                      //   (64: S = 1, 72: S = 2) the partial dot products of the stored row with R bf16 [M, N]: slot s of head h is the
                      //   part inside the (s + 1)-th 64-column strip the head touches.  NN, 256^2 four-wave kernel only: the attention
                      //   backward's delta = rowsum(dO * O) formed where dO is produced (reed_attention_bwd_dp adds the slots)
  EPI_GELU_ERF = 11,  // C2 bf16 = GELU(erf)(pre): nn.GELU() of the timm / I-JEPA towers' Mlp (its own instantiation since round 4;
                      //   the fp32-operand build folds it into EPI_QGELU with GemmArgs::act_variant = 1)
  // Round 5: the activation's DERIVATIVE is formed where the activation is (the forward epilogue holds sigmoid(2u) already) and
  // saved in the array the pre-activation used to occupy — the pre-activation of fc1 / the projector layers was kept for the
  // backward's dGELU / dSiLU epilogue only — so that epilogue becomes one multiply per element (EPI_MUL):
  EPI_GELU_G = 14,    // C bf16 = bf16(gelu_tanh'(pre)) (optional), C2 bf16 = gelu_tanh(pre), pre = bf16(acc+bias)
  EPI_SILU_G = 15,    // same with SiLU
  EPI_MUL = 16        // C bf16 = bf16(bf16(acc) * R bf16)
};

struct GemmArgs {
  const bf16* P;
  const bf16* Q;
  long ldp, ldq;
  int M, N, K;
  void* C;
  long ldc;
  void* C2;
  long ldc2;
  const void* R;
  long ldr;
  const bf16* bias;
  const bf16* gate;
  long ldgate;
  int rows_per_gate;
  float* dbias;  // TN only: dbias[m] (+)= sum_k P[k][m]
  int accumulate;
  int ksplit_len;
  long slab_stride;
  int act_variant;   // fp32-operand build only: EPI_QGELU as 0 = QuickGELU (CLIP), 1 = GELU(erf) (timm / I-JEPA Mlp, nn.GELU)
  int tile_gm;   // gemm256: tile rows per XCD-local group of the workgroup -> tile map (set by launch256)
};

int reed_gemm_launch(int layout, int epi, GemmArgs a, int splits, hipStream_t stream);
