// Optimiser side of the train step as ONE pass over a flat parameter arena
// (image/train.py:402-409 clip_grad_norm_(1.0) + AdamW.step, :94-105/:411-412 update_ema):
// the reference walks 298 tensors with foreach kernels (~40 B/param of HBM traffic spread over
// hundreds of launches); here grads, master weights, both Adam moments, the EMA copy and the bf16
// compute shadow are contiguous arenas with identical layout, so it is three launches:
// squared-norm partials, finalize (norm + clip coefficient stay on the device: no host sync), and a
// fused clip * AdamW + EMA + bf16-shadow update.  Deterministic (fixed reduction order).
#include "../../include/reed_hip.h"
#include <stdlib.h>

#include "common.hpp"

#ifndef REED_OPT_NT
#define REED_OPT_NT 1
#endif
namespace {

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n4, long n,
                                                     float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v = *(const f32x4*)(g + i * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void clip_finalize_kernel(const float* __restrict__ partial, int nb, float max_norm,
                                                            float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += (double)partial[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float norm = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
    out[0] = norm;
    out[1] = fminf(1.f, max_norm / (norm + 1e-6f));
  }
}

// Dynamic loss scaling for fp16 training (torch.amp.GradScaler as accelerate drives it: scale(loss).backward(), unscale_
// before clip_grad_norm_, step skipped on inf / nan, update(); image/train.py:401-409 under --mixed-precision fp16), on the
// device: state = [scale, growth_tracker, found_inf (this step), good_steps (optimiser steps actually taken)].
// The gradients in the arena are scale x the true ones; out[0] = the true norm (inf / nan on overflow, as the reference
// logs it), out[1] = clip coefficient / scale (unscale and clip in the one multiply the update applies), 0 on overflow.
__global__ __launch_bounds__(256) void clip_finalize_scaled_kernel(const float* __restrict__ partial, int nb, float max_norm,
                                                                   float* __restrict__ out, float* __restrict__ st,
                                                                   float growth, float backoff, float interval) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += (double)partial[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float scale = st[0];
    const float norm_s = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
    const bool bad = !(norm_s <= 3.0e38f);          // inf or nan anywhere in the scaled gradients
    const float norm = norm_s / scale;
    out[0] = norm;
    out[1] = bad ? 0.f : (max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f) / scale;
    if (bad) {
      st[0] = scale * backoff;
      st[1] = 0.f;
      st[2] = 1.f;
    } else {
      float tr = st[1] + 1.f;
      if (tr >= interval) { st[0] = scale * growth; tr = 0.f; }
      st[1] = tr;
      st[2] = 0.f;
      st[3] += 1.f;
    }
  }
}

struct AdamArgs {
  float lr, beta1, beta2, eps, wd, bc1, bc2, ema_decay;
};

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ ema, bf16* __restrict__ shadow, long n4_train,
                                                        long n4_total, const float* __restrict__ norm_clip,
                                                        const float* __restrict__ scaler, AdamArgs a) {
  const float clip = norm_clip ? norm_clip[1] : 1.f;
  // with a loss scaler: an overflowed step leaves p, m, v untouched (GradScaler.step skips optimizer.step; the EMA update
  // and the shadow still run, train.py:411-412), and the bias corrections count the steps actually taken
  const bool skip = scaler && scaler[2] != 0.f;
  const float bc1 = scaler ? 1.f - powf(a.beta1, scaler[3]) : a.bc1;
  const float bc2 = scaler ? 1.f - powf(a.beta2, scaler[3]) : a.bc2;
  const float step_size = a.lr / bc1;
  const float bc2s = sqrtf(bc2);
  // REED_OPT_NT (round 4, default): the state arrays (read once and written once per step: 38 B per parameter) with the
  // non-temporal policy, so that the pass, which runs beside the next forward, does not walk through the caches its GEMMs'
  // operands are served from; the 16-bit weights it writes for that forward keep the default policy.  Bit-identical.  The pass
  // alone 5.13 -> 4.84 ms (5.60 -> 5.94 TB/s); whole step at b = 32 934.5 -> 947.8 images/s (three pairs), at b = 256 unchanged
  // (profiles/r4_optimizer_nt.txt)
#if REED_OPT_NT
#define OPT_LD(ptr) __builtin_nontemporal_load((const f32x4*)(ptr))
#define OPT_ST(ptr, val) __builtin_nontemporal_store((val), (f32x4*)(ptr))
#else
#define OPT_LD(ptr) (*(const f32x4*)(ptr))
#define OPT_ST(ptr, val) (*(f32x4*)(ptr) = (val))
#endif
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4_total; i += (long)gridDim.x * 256) {
    f32x4 pv = OPT_LD(p + i * 4);
    if (i < n4_train && !skip) {
      f32x4 gv = OPT_LD(g + i * 4);
      f32x4 mv = OPT_LD(m + i * 4);
      f32x4 vv = OPT_LD(v + i * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gg = gv[j] * clip;
        float pp = pv[j] * (1.f - a.lr * a.wd);
        mv[j] = mv[j] + (1.f - a.beta1) * (gg - mv[j]);           // exp_avg.lerp_(grad, 1-beta1)
        vv[j] = vv[j] * a.beta2 + (1.f - a.beta2) * gg * gg;      // mul_(beta2).addcmul_(g, g, 1-beta2)
        float denom = sqrtf(vv[j]) / bc2s + a.eps;
        pv[j] = pp - step_size * (mv[j] / denom);
      }
      OPT_ST(p + i * 4, pv);
      OPT_ST(m + i * 4, mv);
      OPT_ST(v + i * 4, vv);
    }
    if (ema) {
      f32x4 ev = OPT_LD(ema + i * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) ev[j] = ev[j] * a.ema_decay + pv[j] * (1.f - a.ema_decay);  // mul_(d).add_(p, alpha=1-d)
      OPT_ST(ema + i * 4, ev);
    }
    if (shadow) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = f2bf(pv[j]);
      *(bf16x4*)(shadow + i * 4) = o;
    }
  }
}

}  // namespace

extern "C" int reed_grad_sqnorm(const float* g, int64_t n, float* partial, int nblocks, void* stream) {
  REED_CHECK_ARG(g && partial && nblocks > 0 && n >= 0, "grad_sqnorm: bad args");
  REED_CHECK_ARG(((uintptr_t)g % 16) == 0, "grad_sqnorm: misaligned");
  REED_KLAUNCH(sqnorm_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, g, (long)(n >> 2), (long)n,
                     partial);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_clip_finalize(const float* partial, int nblocks, float max_norm, float* norm_clip,
                                  void* stream) {
  REED_CHECK_ARG(partial && norm_clip, "clip_finalize: null pointer");
  REED_KLAUNCH(clip_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblocks, max_norm,
                     norm_clip);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_clip_finalize_scaled(const float* partial, int nblocks, float max_norm, float* norm_clip,
                                         float* scaler_state, float growth_factor, float backoff_factor,
                                         float growth_interval, void* stream) {
  REED_CHECK_ARG(partial && norm_clip && scaler_state, "clip_finalize_scaled: null pointer");
  REED_KLAUNCH(clip_finalize_scaled_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblocks, max_norm,
               norm_clip, scaler_state, growth_factor, backoff_factor, growth_interval);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, void* shadow,
                              int64_t n_train, int64_t n_total, const float* norm_clip, const float* scaler_state,
                              float lr, float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2,
                              float ema_decay, void* stream) {
  REED_CHECK_ARG(p && (n_train == 0 || (g && m && v)), "adamw_ema: null pointer");
  REED_CHECK_ARG(n_train % 4 == 0 && n_total % 4 == 0 && n_train <= n_total,
                 "adamw_ema: n_train=%ld n_total=%ld must be multiples of 4 with n_train <= n_total", (long)n_train,
                 (long)n_total);
  AdamArgs a{lr, beta1, beta2, eps, weight_decay, bc1, bc2, ema_decay};
  long n4 = n_total >> 2;
  int blocks = (int)((n4 + 255) / 256);
  constexpr int cap = 8192;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  REED_KLAUNCH(adamw_ema_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema,
                     (bf16*)shadow, (long)(n_train >> 2), n4, norm_clip, scaler_state, a);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
