// Shared device/host helpers for the reed_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit operand type of every kernel.  The product library (libreed_hip.so) is built with bfloat16 — the reference's
// training precision under accelerate bf16 autocast.  The SAME sources build a second library (libreed_hip_f16.so,
// -DREED_FP16) with IEEE half operands for the sampling path: the reference samples with an fp32 model under TF32
// (image/generate.py:41,183), whose 10-bit mantissa is half's, at the same MFMA rate as bf16 (v_mfma_f32_16x16x32_f16).
// The type keeps its name `bf16` in the sources ("the 16-bit operand"); only this block knows which one it is.
#if defined(REED_FP32)
// Third build (libreed_hip_f32.so): the reference's `--mixed-precision no` / `generate.py --no-tf32` arithmetic
// (image/train.py:505, image/generate.py:41,183).  The "16-bit operand" IS float: every rounding point of the mixed-precision
// contract (f2bf, bfround) becomes the identity, activations and the weight shadow are fp32 arrays, and the contractions run
// on the fp32-input matrix instruction v_mfma_f32_32x32x2_f32 (exact fp32, 1/16 of the 16-bit rate; csrc/gemm_f32.hip,
// csrc/attention_f32.hip).  The MFMA-tuned 16-bit kernels (gemm*.hip, gemm_tn.hip, attention.hip, encoder.hip) are not part
// of this build; the row / loss / optimiser / sampler kernels are the same sources.
typedef float bf16;
#define REED_HALF_KIND 2
#elif defined(REED_FP16)
typedef _Float16 bf16;
#define REED_HALF_KIND 1
#define REED_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define REED_MFMA_MNEMONIC "v_mfma_f32_16x16x32_f16"
#else
typedef __bf16 bf16;
#define REED_HALF_KIND 0
#define REED_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define REED_MFMA_MNEMONIC "v_mfma_f32_16x16x32_bf16"
#define REED_DS_READ_TR16_B64(p) __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p)
#endif
typedef __attribute__((ext_vector_type(8))) bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) bf16 bf16x2;
#ifdef REED_FP16   // the typed builtin of the transposing LDS read wants clang's __fp16 vector
typedef __fp16 reed_tr16_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define REED_DS_READ_TR16_B64(p) \
  __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((reed_tr16_t __attribute__((address_space(3)))*)(p)))
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// acc += a * b with the accumulator held IN PLACE in accumulation registers (AGPRs).  With the builtin the compiler selects
// the untied early-clobber form (vdst != src C), which needs a spare register quad per instruction: a wave that keeps all
// 256 AGPRs live as accumulators (csrc/gemm256w.hip) then gets part of them shuffled through VGPRs around every MFMA.
// The asm is opaque to the hazard recognizer: no two consecutive uses of one accumulator, and s_nop before reading it back.
#define REED_MFMA_ACC(acc, a, b) asm volatile(REED_MFMA_MNEMONIC " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
// first MFMA of an accumulation: C = 0 as the inline constant (no zeroing pass over the accumulation registers)
#define REED_MFMA_ACC_Z(acc, a, b) asm volatile(REED_MFMA_MNEMONIC " %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b))
// the same with the accumulator in VGPRs, for operands the compiler may have just (re)materialised with VALU moves — e.g. a
// fragment of ones: the matrix pipe reads its sources too early for a VALU write in the previous cycles (measured: stale
// reads), and the hazard recognizer does not look into asm — hence the wait states in front
#define REED_MFMA_ACC_V(acc, a, b) \
  asm volatile("s_nop 3\n\t" REED_MFMA_MNEMONIC " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

#define REED_OK 0
#define REED_ERR_ARG 1001       // bad argument (shape/alignment/unsupported dim)
#define REED_ERR_UNSUPPORTED 1002

// Error string storage lives in api.cpp.
extern "C" void reed_set_error(const char* fmt, ...);

#define REED_CHECK_ARG(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      reed_set_error(__VA_ARGS__);       \
      return REED_ERR_ARG;               \
    }                                    \
  } while (0)

// hipGetLastError() is sticky per thread and shared with every other HIP user in the process (PyTorch): clear it
// before our launch so that REED_LAUNCH_CHECK reports only our own launch failure.
#define REED_KLAUNCH(...)              \
  do {                                 \
    (void)hipGetLastError();           \
    hipLaunchKernelGGL(__VA_ARGS__);   \
  } while (0)

#define REED_LAUNCH_CHECK()                                        \
  do {                                                             \
    hipError_t e__ = hipGetLastError();                            \
    if (e__ != hipSuccess) {                                       \
      reed_set_error("HIP launch error: %s", hipGetErrorString(e__)); \
      return (int)e__;                                             \
    }                                                              \
  } while (0)

// ---- scalar helpers -------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }   // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
__device__ __forceinline__ float bfround(float x) { return (float)(bf16)x; }

// Hardware transcendentals: v_rcp_f32 / v_exp_f32 are 1-ulp, quarter-rate single instructions.  (`__frcp_rn` and `1/x`
// expand to the 10-instruction IEEE division sequence, `__expf` to a range-checked scale: in the GEMM epilogues that
// VALU work, not HBM, was what the MFMA pipe waited on.)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// sigmoid(y) = 1 / (1 + 2^(-y log2 e)); exp2 -> inf gives rcp -> 0, the correct limit.
__device__ __forceinline__ float sigmoid_f(float y) { return fast_rcp(1.f + fast_exp2(-1.4426950408889634f * y)); }

// GELU(tanh approx) = 0.5 x (1 + tanh(u)), u = sqrt(2/pi)(x + 0.044715 x^3).  Since 0.5(1 + tanh(u)) = sigmoid(2u),
// gelu(x) = x * s with s = 1/(1 + 2^t), t = -2u log2(e) = x (c0 + c1 x^2): three full-rate ops + one v_exp_f32 + one
// v_rcp_f32 instead of a libm tanhf (300 M elements per fc1 call at b=256).  Error << bf16 resolution of the result.
__device__ __forceinline__ float gelu_sig(float x) {
  const float c0 = -2.f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
  return fast_rcp(1.f + fast_exp2(x * fmaf(c1, x * x, c0)));
}
__device__ __forceinline__ float gelu_tanh_f(float x) { return x * gelu_sig(x); }
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  // d/dx [x s(2u)] = s + x s (1 - s) 2 u',  2u' = 2 sqrt(2/pi)(1 + 3*0.044715 x^2)
  const float d0 = 2.f * 0.7978845608028654f, d1 = d0 * 3.f * 0.044715f;
  float s = gelu_sig(x);
  float xs = x * s;
  return fmaf(xs - xs * s, fmaf(d1, x * x, d0), s);
}
// Two elements at a time (round 4): the same operations as gelu_tanh_f / gelu_tanh_grad_f on v_pk_mul_f32 / v_pk_fma_f32 /
// v_pk_add_f32 — for ONE wave per SIMD a packed instruction issues in the 4 cycles of a plain one (tools/micro/valu_rate.hip), the
// compiler's SLP pass packs only about a third of the scalar form's multiplies and adds, and the GELU / dGELU epilogues of the
// four-wave GEMM are vector-issue-bound.  exp and rcp stay per element (8 cycles each).
__device__ __forceinline__ f32x2 gelu_sig2(f32x2 x) {
  const float c0 = -2.f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
  const f32x2 t = x * __builtin_elementwise_fma(f32x2{c1, c1}, x * x, f32x2{c0, c0});
  const f32x2 d = f32x2{fast_exp2(t[0]), fast_exp2(t[1])} + 1.f;
  return f32x2{fast_rcp(d[0]), fast_rcp(d[1])};
}
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) { return x * gelu_sig2(x); }
__device__ __forceinline__ f32x2 gelu_tanh_grad2(f32x2 x) {
  const float d0 = 2.f * 0.7978845608028654f, d1 = d0 * 3.f * 0.044715f;
  const f32x2 s = gelu_sig2(x);
  const f32x2 xs = x * s;
  return __builtin_elementwise_fma(xs - xs * s, __builtin_elementwise_fma(f32x2{d1, d1}, x * x, f32x2{d0, d0}), s);
}
// Round 5: activation AND derivative from the one sigmoid (the forward epilogues EPI_GELU_G / EPI_SILU_G save the derivative for
// the backward's one-multiply epilogue EPI_MUL): gelu = x s, gelu' = s + (x s - x s s)(d0 + d1 x^2) — three more packed
// instructions per pair on top of gelu_tanh2.
__device__ __forceinline__ void gelu_tanh_both2(f32x2 x, f32x2& act, f32x2& grad) {
  const float d0 = 2.f * 0.7978845608028654f, d1 = d0 * 3.f * 0.044715f;
  const f32x2 s = gelu_sig2(x);
  act = x * s;
  grad = __builtin_elementwise_fma(act - act * s, __builtin_elementwise_fma(f32x2{d1, d1}, x * x, f32x2{d0, d0}), s);
}
__device__ __forceinline__ void gelu_tanh_both(float x, float& act, float& grad) {
  const float d0 = 2.f * 0.7978845608028654f, d1 = d0 * 3.f * 0.044715f;
  const float s = gelu_sig(x);
  act = x * s;
  grad = fmaf(act - act * s, fmaf(d1, x * x, d0), s);
}
// nn.GELU() (exact): 0.5 x (1 + erf(x / sqrt 2)) — the frozen ViT towers' Mlp activation (inference only)
#if defined(REED_FP32)
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.7071067811865476f)); }
#else
// 16-bit builds (round 4): x Phi(x) with Phi from Abramowitz & Stegun 7.1.26 — 0.5 erfc(z) = 0.5 t (a1 + t (a2 + t (a3 + t (a4 + t a5))))
// exp(-z^2), t = 1 / (1 + p z), z = |x| / sqrt 2, |error of erf| <= 1.5e-7 — formed on the side where it does not cancel (x < 0: Phi = that
// value, x >= 0: 1 - it): 12 full-rate instructions + v_rcp_f32 + v_exp_f32 where libm's erff is about 60 per element (the fc1
// GEMM of a DINOv2 / MAE / I-JEPA tower spent 58 of its 192 us there at batch 64).  The result is rounded to 16 bits (2^-9
// relative); the fp32 build keeps erff.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.7071067811865476f;
  const float t = fast_rcp(fmaf(0.3275911f, z, 1.f));
  float q = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  q = fmaf(q, t, 0.5f * 1.421413741f);
  q = fmaf(q, t, 0.5f * -0.284496736f);
  q = fmaf(q, t, 0.5f * 0.254829592f);
  q = q * t * fast_exp2(z * z * -1.4426950408889634f);
  return x * (x < 0.f ? q : 1.f - q);
}
#endif
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float silu_grad_f(float x) {
  float s = sigmoid_f(x);
  return s * (1.f + x * (1.f - s));
}
__device__ __forceinline__ void silu_both(float x, float& act, float& grad) {
  const float s = sigmoid_f(x);
  act = x * s;
  grad = s * (1.f + x * (1.f - s));
}

// ---- wave / block reductions (wave = 64 lanes) -----------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
