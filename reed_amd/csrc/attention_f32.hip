// Multi-head self-attention of the fp32 build (libreed_hip_f32.so, -DREED_FP32): timm Attention under the reference's
// `--mixed-precision no` / `generate.py --no-tf32` (image/models/sit.py:114-118,134; image/train.py:505;
// image/generate.py:41,183), i.e. SDPA on fp32 q, k, v with fp32 softmax:
//   o = softmax(q k^T / sqrt(hd)) v      q,k,v = qkv.reshape(B,T,3,H,hd).permute(2,0,3,1,4)        qkv, o, dqkv: fp32 arrays
// Attention is 3.6 % of the model's flop and this build's GEMMs run at the fp32 MFMA rate (1/16 of bf16), so the form here
// is the plain one: exact fp32 on the vector ALUs, one thread per query row (forward, dQ) or per key row (dK, dV), the
// other side's rows staged through LDS 32 at a time and read as wave-wide broadcasts, online softmax in 8-key chunks,
// backward recomputed from the saved log-sum-exp in two deterministic kernels (no atomics).  Any T (256 / 1024 tokens at
// 256^2 / 512^2), head_dim 64 or 72.  Same entry points and layouts as csrc/attention.hip, which is not part of this build.
#include "../../include/reed_hip.h"
#include "common.hpp"

#ifndef REED_FP32
#error "attention_f32.hip belongs to the fp32-operand build (-DREED_FP32) only"
#endif

namespace {

constexpr int KT = 32;   // rows of the other side staged per LDS tile

// rows [r0, r0 + KT) of q, k or v (which = 0, 1, 2) of head h of sample b -> tile[KT][HD]; rows >= T read as zero
template <int HD>
__device__ __forceinline__ void stage_rows(float* tile, const float* qkv, int which, long b, int h, int H, int T, int r0,
                                           int tid) {
  constexpr int NV = HD / 4;
  for (int i = tid; i < KT * NV; i += 256) {
    const int r = i / NV, c = i % NV;
    const int row = r0 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < T) v = *(const f32x4*)(qkv + (((b * T + row) * 3 + which) * H + h) * HD + 4 * c);
    *(f32x4*)(tile + r * HD + 4 * c) = v;
  }
}
// the same for an [M, H*HD] array (o, d_o)
template <int HD>
__device__ __forceinline__ void stage_rows_o(float* tile, const float* x, long b, int h, int H, int T, int r0, int tid) {
  constexpr int NV = HD / 4;
  for (int i = tid; i < KT * NV; i += 256) {
    const int r = i / NV, c = i % NV;
    const int row = r0 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < T) v = *(const f32x4*)(x + ((b * T + row) * H + h) * HD + 4 * c);
    *(f32x4*)(tile + r * HD + 4 * c) = v;
  }
}

template <int HD>
__device__ __forceinline__ float dot_row(const float (&a)[HD], const float* row) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    const f32x4 k = *(const f32x4*)(row + 4 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) s = fmaf(a[4 * c + e], k[e], s);
  }
  return s;
}
template <int HD>
__device__ __forceinline__ void axpy_row(float (&acc)[HD], float p, const float* row) {
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    const f32x4 v = *(const f32x4*)(row + 4 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[4 * c + e] = fmaf(p, v[e], acc[4 * c + e]);
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                           float* __restrict__ lse, int T, int H) {
  __shared__ __attribute__((aligned(16))) float sk[KT * HD], sv[KT * HD];
  const int tid = threadIdx.x;
  const long b = blockIdx.x / H;
  const int h = blockIdx.x % H;
  const int q = blockIdx.y * 256 + tid;
  const bool live = q < T;
  const float scale = rsqrtf((float)HD);
  float qv[HD], acc[HD];
  {
    const float* qp = qkv + (((b * T + (live ? q : 0)) * 3 + 0) * H + h) * HD;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const f32x4 v = *(const f32x4*)(qp + 4 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { qv[4 * c + e] = v[e] * scale; acc[4 * c + e] = 0.f; }
    }
  }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < T; k0 += KT) {
    __syncthreads();
    stage_rows<HD>(sk, qkv, 1, b, h, H, T, k0, tid);
    stage_rows<HD>(sv, qkv, 2, b, h, H, T, k0, tid);
    __syncthreads();
#pragma unroll 1
    for (int c0 = 0; c0 < KT; c0 += 8) {
      float s[8];
      float mc = -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s[j] = (k0 + c0 + j < T) ? dot_row<HD>(qv, sk + (c0 + j) * HD) : -INFINITY;
        mc = fmaxf(mc, s[j]);
      }
      if (mc == -INFINITY) continue;       // a chunk beyond T (uniform across the block)
      const float mn = fmaxf(m, mc);
      const float corr = expf(m - mn);   // m = -inf on the first chunk: exp(-inf) = 0
      l *= corr;
#pragma unroll
      for (int e = 0; e < HD; ++e) acc[e] *= corr;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float p = expf(s[j] - mn);   // masked keys: exp(-inf) = 0
        l += p;
        axpy_row<HD>(acc, p, sv + (c0 + j) * HD);
      }
      m = mn;
    }
  }
  if (live) {
    const float inv = 1.f / l;
    float* op = o + ((b * T + q) * H + h) * HD;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c)
      *(f32x4*)(op + 4 * c) = f32x4{acc[4 * c] * inv, acc[4 * c + 1] * inv, acc[4 * c + 2] * inv, acc[4 * c + 3] * inv};
    if (lse) lse[(b * H + h) * T + q] = m + logf(l);
  }
}

// dQ: thread = query row i.  p_ij = exp(scale q_i.k_j - lse_i); dp = dO_i.v_j; ds = p (dp - delta_i), delta_i = dO_i.o_i;
// dq_i = scale * sum_j ds_ij k_j
template <int HD>
__global__ __launch_bounds__(256) void attn_f32_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                              const float* __restrict__ d_o, const float* __restrict__ lse,
                                                              float* __restrict__ dqkv, int T, int H) {
  __shared__ __attribute__((aligned(16))) float sk[KT * HD], sv[KT * HD];
  const int tid = threadIdx.x;
  const long b = blockIdx.x / H;
  const int h = blockIdx.x % H;
  const int q = blockIdx.y * 256 + tid;
  const bool live = q < T;
  const int qc = live ? q : 0;
  const float scale = rsqrtf((float)HD);
  float qv[HD], dov[HD], dq[HD];
  float delta = 0.f;
  {
    const float* qp = qkv + (((b * T + qc) * 3 + 0) * H + h) * HD;
    const float* dp = d_o + ((b * T + qc) * H + h) * HD;
    const float* op = o + ((b * T + qc) * H + h) * HD;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const f32x4 v = *(const f32x4*)(qp + 4 * c), g = *(const f32x4*)(dp + 4 * c), ov = *(const f32x4*)(op + 4 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qv[4 * c + e] = v[e] * scale;
        dov[4 * c + e] = g[e];
        dq[4 * c + e] = 0.f;
        delta = fmaf(g[e], ov[e], delta);
      }
    }
  }
  const float ls = lse[(b * H + h) * T + qc];
  for (int k0 = 0; k0 < T; k0 += KT) {
    __syncthreads();
    stage_rows<HD>(sk, qkv, 1, b, h, H, T, k0, tid);
    stage_rows<HD>(sv, qkv, 2, b, h, H, T, k0, tid);
    __syncthreads();
    const int kn = min(KT, T - k0);
#pragma unroll 1
    for (int j = 0; j < kn; ++j) {
      const float p = expf(dot_row<HD>(qv, sk + j * HD) - ls);
      const float ds = p * (dot_row<HD>(dov, sv + j * HD) - delta);
      axpy_row<HD>(dq, ds, sk + j * HD);
    }
  }
  if (live) {
    float* gp = dqkv + (((b * T + q) * 3 + 0) * H + h) * HD;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c)
      *(f32x4*)(gp + 4 * c) = f32x4{dq[4 * c] * scale, dq[4 * c + 1] * scale, dq[4 * c + 2] * scale, dq[4 * c + 3] * scale};
  }
}

// dK, dV: thread = key row j.  dv_j = sum_i p_ij dO_i;  dk_j = scale * sum_i ds_ij q_i
template <int HD>
__global__ __launch_bounds__(256) void attn_f32_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                               const float* __restrict__ d_o, const float* __restrict__ lse,
                                                               float* __restrict__ dqkv, int T, int H) {
  __shared__ __attribute__((aligned(16))) float sq[KT * HD], sdo[KT * HD];
  __shared__ float sls[KT], sdl[KT];
  const int tid = threadIdx.x;
  const long b = blockIdx.x / H;
  const int h = blockIdx.x % H;
  const int kr = blockIdx.y * 256 + tid;
  const bool live = kr < T;
  const int kc = live ? kr : 0;
  const float scale = rsqrtf((float)HD);
  float kv[HD], vv[HD], dk[HD], dv[HD];
  {
    const float* kp = qkv + (((b * T + kc) * 3 + 1) * H + h) * HD;
    const float* vp = qkv + (((b * T + kc) * 3 + 2) * H + h) * HD;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const f32x4 a = *(const f32x4*)(kp + 4 * c), v = *(const f32x4*)(vp + 4 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        kv[4 * c + e] = a[e] * scale;
        vv[4 * c + e] = v[e];
        dk[4 * c + e] = 0.f;
        dv[4 * c + e] = 0.f;
      }
    }
  }
  for (int q0 = 0; q0 < T; q0 += KT) {
    __syncthreads();
    stage_rows<HD>(sq, qkv, 0, b, h, H, T, q0, tid);
    stage_rows_o<HD>(sdo, d_o, b, h, H, T, q0, tid);
    if (tid < KT) {     // per staged query: log-sum-exp and delta = dO.o (rows >= T: p = exp(0 - inf) = 0)
      const int row = q0 + tid;
      float dl = 0.f, ls = INFINITY;
      if (row < T) {
        const float* op = o + ((b * T + row) * H + h) * HD;
        const float* gp = d_o + ((b * T + row) * H + h) * HD;
        for (int e = 0; e < HD; ++e) dl = fmaf(gp[e], op[e], dl);
        ls = lse[(b * H + h) * T + row];
      }
      sls[tid] = ls;
      sdl[tid] = dl;
    }
    __syncthreads();
    const int qn = min(KT, T - q0);
#pragma unroll 1
    for (int i = 0; i < qn; ++i) {
      const float p = expf(dot_row<HD>(kv, sq + i * HD) - sls[i]);
      const float ds = p * (dot_row<HD>(vv, sdo + i * HD) - sdl[i]);
      axpy_row<HD>(dv, p, sdo + i * HD);
      axpy_row<HD>(dk, ds, sq + i * HD);
    }
  }
  if (live) {
    float* gk = dqkv + (((b * T + kr) * 3 + 1) * H + h) * HD;
    float* gv = dqkv + (((b * T + kr) * 3 + 2) * H + h) * HD;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      *(f32x4*)(gk + 4 * c) = f32x4{dk[4 * c] * scale, dk[4 * c + 1] * scale, dk[4 * c + 2] * scale, dk[4 * c + 3] * scale};
      *(f32x4*)(gv + 4 * c) = f32x4{dv[4 * c], dv[4 * c + 1], dv[4 * c + 2], dv[4 * c + 3]};
    }
  }
}

}  // namespace

extern "C" int reed_attention_fwd(const void* qkv, void* o, float* lse, int B, int T, int H, int hd, void* stream) {
  REED_CHECK_ARG(qkv && o, "attention_fwd: null pointer");
  REED_CHECK_ARG(hd == 64 || hd == 72, "attention (fp32 build): head_dim %d unsupported (64 or 72)", hd);
  REED_CHECK_ARG(B > 0 && T > 0 && H > 0, "attention: bad dims B=%d T=%d H=%d", B, T, H);
  const dim3 grid(B * H, (T + 255) / 256);
  if (hd == 64) REED_KLAUNCH(attn_f32_fwd_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)qkv, (float*)o, lse, T, H);
  else REED_KLAUNCH(attn_f32_fwd_kernel<72>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)qkv, (float*)o, lse, T, H);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int B, int T,
                                  int H, int hd, void* stream) {
  REED_CHECK_ARG(qkv && o && d_o && lse && dqkv, "attention_bwd: null pointer");
  REED_CHECK_ARG(hd == 64 || hd == 72, "attention (fp32 build): head_dim %d unsupported (64 or 72)", hd);
  REED_CHECK_ARG(B > 0 && T > 0 && H > 0, "attention: bad dims B=%d T=%d H=%d", B, T, H);
  const dim3 grid(B * H, (T + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
#define REED_BWD_F32(HD)                                                                                                 \
  do {                                                                                                                   \
    REED_KLAUNCH(attn_f32_bwd_dq_kernel<HD>, grid, dim3(256), 0, s, (const float*)qkv, (const float*)o, (const float*)d_o, \
                 lse, (float*)dqkv, T, H);                                                                               \
    REED_LAUNCH_CHECK();                                                                                                 \
    REED_KLAUNCH(attn_f32_bwd_dkv_kernel<HD>, grid, dim3(256), 0, s, (const float*)qkv, (const float*)o,                  \
                 (const float*)d_o, lse, (float*)dqkv, T, H);                                                            \
  } while (0)
  if (hd == 64) REED_BWD_F32(64);
  else REED_BWD_F32(72);
#undef REED_BWD_F32
  REED_LAUNCH_CHECK();
  return REED_OK;
}

// the workspace form of the 16-bit builds (their persistent backward wants delta = rowsum(dO * O) precomputed): no workspace here
extern "C" int64_t reed_attention_bwd_ws_floats(int, int, int) { return 0; }
extern "C" int reed_attention_bwd_ws(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, float*, int B,
                                     int T, int H, int hd, void* stream) {
  return reed_attention_bwd(qkv, o, d_o, lse, dqkv, B, T, H, hd, stream);
}

// the partial-dot-product form belongs to the 16-bit GEMM's epilogue 13 (csrc/gemm_common.hpp): not in this build
extern "C" int reed_attention_bwd_dp(const void*, const void*, const float*, const float*, void*, float*, int, int, int, int, void*) {
  reed_set_error("reed_attention_bwd_dp: not built for fp32 operands (use reed_attention_bwd_ws)");
  return REED_ERR_UNSUPPORTED;
}
