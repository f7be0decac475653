// fp64 sampler state updates (image/samplers.py:46-104 Euler/Heun ODE with interval CFG,
// :107-187 Euler-Maruyama SDE).  The reference keeps the latent state in float64 and evaluates the
// model in float32; these kernels do the fp64 arithmetic in the reference's operation order with
// contraction disabled, so given identical model outputs the state is bit-identical.
#include "../../include/reed_hip.h"
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void sampler_input_kernel(const double* __restrict__ x, float* __restrict__ out,
                                                            long n, int dup) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = (float)x[i];
  out[i] = v;
  if (dup) out[n + i] = v;
}

__global__ __launch_bounds__(256) void sampler_update_kernel(const double* __restrict__ xc, const float* __restrict__ mo,
                                                             const double* __restrict__ dprev,
                                                             double* __restrict__ dstore, double* __restrict__ xn,
                                                             long n, int cfg, double s, double dt, double w0,
                                                             double w1) {
#pragma clang fp contract(off)
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double d = (double)mo[i];
  if (cfg) {
    double du = (double)mo[n + i];
    d = du + s * (d - du);  // d_uncond + cfg_scale * (d_cond - d_uncond)
  }
  if (dstore) dstore[i] = d;
  double inc;
  if (dprev) {
    double a = w1 * dprev[i];  // 0.5 * d_cur
    double b = w0 * d;         // 0.5 * d_prime
    inc = a + b;
  } else {
    inc = d;
  }
  xn[i] = xc[i] + dt * inc;
}

__global__ __launch_bounds__(256) void sde_update_kernel(const double* __restrict__ xc, const float* __restrict__ mo,
                                                         const double* __restrict__ eps, double* __restrict__ xn,
                                                         long n, int cfg, double s, double t, double dt,
                                                         int path_type, int last) {
#pragma clang fp contract(off)
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double alpha, dalpha, sigma, dsigma;
  if (path_type == 0) { alpha = 1.0 - t; dalpha = -1.0; sigma = t; dsigma = 1.0; }
  else {
    const double hp = 3.141592653589793 / 2;
    alpha = cos(t * hp); sigma = sin(t * hp); dalpha = -hp * sin(t * hp); dsigma = hp * cos(t * hp);
  }
  const double ratio = alpha / dalpha;
  const double var = sigma * sigma - ratio * dsigma * sigma;
  const double diffusion = 2 * t;
  const double x = xc[i];
  double v = (double)mo[i];
  double d = v - 0.5 * diffusion * ((ratio * v - x) / var);
  if (cfg) {
    double vu = (double)mo[n + i];
    double du = vu - 0.5 * diffusion * ((ratio * vu - x) / var);
    d = du + s * (d - du);
  }
  if (last) xn[i] = x + dt * d;
  else {
    double deps = eps[i] * sqrt(fabs(dt));
    xn[i] = x + d * dt + sqrt(diffusion) * deps;
  }
}

}  // namespace

extern "C" int reed_sampler_input(const double* x, float* out, int64_t n_elems, int dup, void* stream) {
  REED_CHECK_ARG(x && out && n_elems > 0, "sampler_input: bad args");
  REED_KLAUNCH(sampler_input_kernel, dim3(cdiv(n_elems, 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                     (long)n_elems, dup);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
extern "C" int reed_sampler_update(const double* x_cur, const float* model_out, const double* d_prev,
                                   double* d_store, double* x_next, int64_t n_elems, int cfg, double cfg_scale,
                                   double dt, double w0, double w1, void* stream) {
  REED_CHECK_ARG(x_cur && model_out && x_next && n_elems > 0, "sampler_update: bad args");
  REED_KLAUNCH(sampler_update_kernel, dim3(cdiv(n_elems, 256)), dim3(256), 0, (hipStream_t)stream, x_cur,
                     model_out, d_prev, d_store, x_next, (long)n_elems, cfg, cfg_scale, dt, w0, w1);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
extern "C" int reed_sde_update(const double* x_cur, const float* model_out, const double* eps, double* x_next,
                               int64_t n_elems, int cfg, double cfg_scale, double t_cur, double dt,
                               int path_type, int last_step, void* stream) {
  REED_CHECK_ARG(x_cur && model_out && x_next && n_elems > 0 && (last_step || eps), "sde_update: bad args");
  REED_KLAUNCH(sde_update_kernel, dim3(cdiv(n_elems, 256)), dim3(256), 0, (hipStream_t)stream, x_cur, model_out,
                     eps, x_next, (long)n_elems, cfg, cfg_scale, t_cur, dt, path_type, last_step);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
