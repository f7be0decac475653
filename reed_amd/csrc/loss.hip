// SILoss arithmetic (image/loss.py:49-64 interpolant, :172-186 x_t / v-target / mean_flat MSE,
// :204-222 cosine alignment of normalized projector outputs against frozen-encoder features).
// All f32, HBM-bound; one wave per feature row for the cosine terms.
#include "../../include/reed_hip.h"
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void interpolant_kernel(const float* __restrict__ x, const float* __restrict__ nz,
                                                          const float* __restrict__ t, float* __restrict__ xt,
                                                          float* __restrict__ tgt, long per, int path_type) {
  const int b = blockIdx.y;
  const float tt = t[b];
  float a, s, da, ds;
  if (path_type == 0) { a = 1.f - tt; s = tt; da = -1.f; ds = 1.f; }
  else {
    const float hp = 1.5707963267948966f;
    a = cosf(tt * hp); s = sinf(tt * hp); da = -hp * sinf(tt * hp); ds = hp * cosf(tt * hp);
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
    long k = (long)b * per + i;
    float xv = x[k], nv = nz[k];
    xt[k] = a * xv + s * nv;
    tgt[k] = da * xv + ds * nv;
  }
}

// sample_posterior (train.py:84-91): z = (mean + std * eps) * scale + bias, moments = cat([mean, std], dim=1)
__global__ __launch_bounds__(256) void sample_posterior_kernel(const float* __restrict__ mom, const float* __restrict__ eps,
                                                               float* __restrict__ out, long half, float scale,
                                                               float bias) {
  const int b = blockIdx.y;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < half; i += (long)gridDim.x * 256) {
    float mean = mom[(long)b * 2 * half + i], sd = mom[(long)b * 2 * half + half + i];
    out[(long)b * half + i] = (mean + sd * eps[(long)b * half + i]) * scale + bias;
  }
}

__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ o, const float* __restrict__ tg,
                                                      float* __restrict__ loss, long per) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float s = 0.f;
  for (long i = threadIdx.x; i < per; i += 256) { float d = o[(long)b * per + i] - tg[(long)b * per + i]; s += d * d; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[b] = (red[0] + red[1] + red[2] + red[3]) / (float)per;
}
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ o, const float* __restrict__ tg,
                                                      const float* __restrict__ gs, float* __restrict__ dout,
                                                      long per) {
  const int b = blockIdx.y;
  const float g = gs[b] * 2.f / (float)per;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
    long k = (long)b * per + i;
    dout[k] = g * (o[k] - tg[k]);
  }
}

// rowdot[m] = <z/max(|z|,eps), zt/max(|zt|,eps)>
__global__ __launch_bounds__(256) void cosine_rows_kernel(const bf16* __restrict__ zt, const float* __restrict__ z,
                                                          float* __restrict__ rowdot, int M, int Z) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  float dot = 0.f, n1 = 0.f, n2 = 0.f;
  for (int i = lane * 4; i < Z; i += 256) {
    bf16x4 a = *(const bf16x4*)(zt + (long)row * Z + i);
    f32x4 b = *(const f32x4*)(z + (long)row * Z + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) { float av = bf2f(a[j]); dot += av * b[j]; n1 += av * av; n2 += b[j] * b[j]; }
  }
  dot = wave_sum(dot); n1 = wave_sum(n1); n2 = wave_sum(n2);
  if (lane == 0) rowdot[row] = dot / (fmaxf(sqrtf(n1), 1e-12f) * fmaxf(sqrtf(n2), 1e-12f));
}
__global__ __launch_bounds__(256) void cosine_sample_kernel(const float* __restrict__ rowdot, float* __restrict__ loss,
                                                            int T) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) s += rowdot[(long)b * T + t];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[b] = -(red[0] + red[1] + red[2] + red[3]) / (float)T;
}
// dzt = gscale[b] * (-1/T) * (zhat - cos * zthat) / max(|zt|, eps)
__global__ __launch_bounds__(256) void cosine_bwd_kernel(const bf16* __restrict__ zt, const float* __restrict__ z,
                                                         const float* __restrict__ gs, bf16* __restrict__ dzt, int M,
                                                         int T, int Z) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  float dot = 0.f, n1 = 0.f, n2 = 0.f;
  for (int i = lane * 4; i < Z; i += 256) {
    bf16x4 a = *(const bf16x4*)(zt + (long)row * Z + i);
    f32x4 b = *(const f32x4*)(z + (long)row * Z + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) { float av = bf2f(a[j]); dot += av * b[j]; n1 += av * av; n2 += b[j] * b[j]; }
  }
  dot = wave_sum(dot); n1 = wave_sum(n1); n2 = wave_sum(n2);
  const float nt = fmaxf(sqrtf(n1), 1e-12f), nz = fmaxf(sqrtf(n2), 1e-12f);
  const float cosv = dot / (nt * nz);
  const float g = -gs[row / T] / (float)T / nt;
  for (int i = lane * 4; i < Z; i += 256) {
    bf16x4 a = *(const bf16x4*)(zt + (long)row * Z + i);
    f32x4 b = *(const f32x4*)(z + (long)row * Z + i);
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f2bf(g * (b[j] / nz - cosv * bf2f(a[j]) / nt));
    *(bf16x4*)(dzt + (long)row * Z + i) = o;
  }
}

}  // namespace

extern "C" int reed_interpolant(const float* x, const float* noise, const float* t, float* xt, float* target,
                                int B, int64_t per, int path_type, void* stream) {
  REED_CHECK_ARG(x && noise && t && xt && target, "interpolant: null pointer");
  REED_CHECK_ARG(path_type == 0 || path_type == 1, "interpolant: path_type %d (0 linear, 1 cosine)", path_type);
  REED_KLAUNCH(interpolant_kernel, dim3(cdiv(per, 1024), B), dim3(256), 0, (hipStream_t)stream, x, noise, t, xt,
                     target, (long)per, path_type);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
extern "C" int reed_sample_posterior(const float* moments, const float* eps, float* out, int B, int64_t half,
                                     float scale, float bias, void* stream) {
  REED_CHECK_ARG(moments && eps && out, "sample_posterior: null pointer");
  REED_KLAUNCH(sample_posterior_kernel, dim3(cdiv(half, 1024), B), dim3(256), 0, (hipStream_t)stream, moments, eps,
                     out, (long)half, scale, bias);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
extern "C" int reed_mse_fwd(const float* out, const float* target, float* loss, int B, int64_t per, void* stream) {
  REED_CHECK_ARG(out && target && loss, "mse_fwd: null pointer");
  REED_KLAUNCH(mse_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, out, target, loss, (long)per);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
extern "C" int reed_mse_bwd(const float* out, const float* target, const float* gscale, float* dout, int B,
                            int64_t per, void* stream) {
  REED_CHECK_ARG(out && target && gscale && dout, "mse_bwd: null pointer");
  REED_KLAUNCH(mse_bwd_kernel, dim3(cdiv(per, 1024), B), dim3(256), 0, (hipStream_t)stream, out, target, gscale,
                     dout, (long)per);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
extern "C" int reed_cosine_fwd(const void* zt, const float* z, float* rowdot, float* loss, int B, int T, int Z,
                               void* stream) {
  REED_CHECK_ARG(zt && z && rowdot && loss, "cosine_fwd: null pointer");
  REED_CHECK_ARG(Z % 4 == 0, "cosine: Z=%d must be a multiple of 4", Z);
  const int M = B * T;
  REED_KLAUNCH(cosine_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)zt, z,
                     rowdot, M, Z);
  REED_KLAUNCH(cosine_sample_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, rowdot, loss, T);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
extern "C" int reed_cosine_bwd(const void* zt, const float* z, const float* gscale, void* dzt, int B, int T,
                               int Z, void* stream) {
  REED_CHECK_ARG(zt && z && gscale && dzt, "cosine_bwd: null pointer");
  REED_CHECK_ARG(Z % 4 == 0, "cosine: Z=%d must be a multiple of 4", Z);
  const int M = B * T;
  REED_KLAUNCH(cosine_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)zt, z,
                     gscale, (bf16*)dzt, M, T, Z);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
