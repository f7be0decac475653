// SD-VAE decoder row passes: the `vae.decode(...)` of image/generate.py:87,156 and image/train.py:446-447.
// The reference calls diffusers' AutoencoderKL; its decoder is 3x3 / 1x1 convolutions, GroupNorm(32, eps 1e-6) + SiLU, nearest x2
// upsampling and one single-head attention over the 32x32 positions.  Here the activations live in HBM as fp32 NHWC, i.e. as the
// row-major token matrix [B*H*W, C], so every convolution IS a reed_gemm call (NT: rows = output positions, Q = the weight as
// [Cout, 9*Cin] in (ky, kx, ci) order, fp32 output + bias, `accumulate` = the residual connection) and a 1x1 convolution / Linear
// needs no data movement at all.  What is left around the contractions are these three HBM-bound passes:
//   reed_groupnorm_stats  one read of the activation -> (mean, rstd) per (image, group), fp64 partial sums in a fixed order; and
//                         the per-channel table (mean, rstd * gamma, beta) the next pass applies
//   reed_conv_rows        the GEMM's P operand: for each output position the 9 (or 1) input pixels of its window, normalised,
//                         scaled, SiLU'd and rounded to the operand type on the way (GroupNorm apply + SiLU + zero padding +
//                         nearest x2 upsampling + im2col in ONE pass: 4 B read, taps * operand bytes written per element)
//   reed_softmax_rows     softmax(scale * S) of the attention's fp32 score rows -> operand type
// Built in all three libraries (operand type = bf16 / half / float).
#include "../../include/reed_hip.h"
#include "common.hpp"

namespace {

// ---- GroupNorm statistics ---------------------------------------------------------------------------------------------
// grid (nchunk, B), 256 threads.  A thread owns the 4-channel piece p = tid % P (P = C / 4) of the rows rl, rl + RP, ... of its
// chunk (RP = 256 / P row lanes), sums x and x^2 per channel in fp64, the row lanes are then added in lane order through LDS:
// ws[(b * nchunk + chunk) * C + c] = {sum, sum of squares} of channel c over the chunk's rows.
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, long hw, int C, long rows_per_chunk,
                                                         double* __restrict__ ws) {
  __shared__ double red[256 * 8];
  const int P = C >> 2, RP = 256 / P;
  const int p = threadIdx.x % P, rl = threadIdx.x / P;
  const long b = blockIdx.y, r0 = (long)blockIdx.x * rows_per_chunk;
  const long r1 = min(hw, r0 + rows_per_chunk);
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (rl < RP) {
    const float* xp = x + (b * hw) * C + 4 * p;
    for (long r = r0 + rl; r < r1; r += RP) {
      const f32x4 v = *(const f32x4*)(xp + r * C);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] += (double)v[e];
        q[e] += (double)v[e] * (double)v[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[threadIdx.x * 8 + e] = s[e];
    red[threadIdx.x * 8 + 4 + e] = q[e];
  }
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < RP; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[p * 8 + e] += red[(k * P + p) * 8 + e];
    double* o = ws + ((b * gridDim.x + blockIdx.x) * C + 4 * p) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] = red[p * 8 + e];
      o[2 * e + 1] = red[p * 8 + 4 + e];
    }
  }
}

// one wave per (image, group): lane l sums chunks l, l + 64, ... (channels of the group in order), then the fixed butterfly.
// table f32 [B, 3, C] (optional) = per channel (mean of its group, rstd * gamma[c], beta[c]): what reed_conv_rows applies as
// fma(x - mean, rstd * gamma, beta)
__global__ __launch_bounds__(256) void gn_final_kernel(const double* __restrict__ ws, int B, long hw, int C, int G, int nchunk,
                                                       float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ stats, float* __restrict__ table) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= B * G) return;
  const int b = i / G, g = i % G, cpg = C / G;
  double s = 0, q = 0;
  for (int k = lane; k < nchunk; k += 64) {
    const double* w = ws + (((long)b * nchunk + k) * C + (long)g * cpg) * 2;
    for (int c = 0; c < cpg; ++c) {
      s += w[2 * c];
      q += w[2 * c + 1];
    }
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  const double n = (double)hw * cpg, mean = s / n;
  const double var = fmax(q / n - mean * mean, 0.0);       // biased, as nn.GroupNorm
  const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (stats && lane == 0) {
    stats[2 * i] = meanf;
    stats[2 * i + 1] = rstd;
  }
  if (table) {
    float* t = table + (long)b * 3 * C + g * cpg;
    for (int c = lane; c < cpg; c += 64) {
      t[c] = meanf;
      t[C + c] = rstd * gamma[g * cpg + c];
      t[2 * C + c] = beta[g * cpg + c];
    }
  }
}

// ---- the GEMM's row operand ----------------------------------------------------------------------------------------------
// A thread owns the VW-channel piece p = tid % PX of the row lane ty = tid / PX (PX = pieces per tap, at most 256; RPB = 256 / PX
// rows per block and pass): the row's pixel is decomposed once (32-bit), the norm table of (image, piece) is loaded once, and
// the TAPS window positions are VW * 4-byte loads / VW * operand-size stores; columns [TAPS * C, kcols) are written as zeros.
template <int VW> struct RowVec;
template <> struct RowVec<4> { typedef bf16x4 type; };
template <> struct RowVec<8> { typedef bf16x8 type; };

template <int TAPS, int VW>
__global__ __launch_bounds__(256) void conv_rows_kernel(const float* __restrict__ x, const float* __restrict__ table, int Hi, int Wi,
                                                        int C, int silu, int up, long row0, long nrows, int kcols,
                                                        bf16* __restrict__ out, long ldo, int PX, int RPB) {
  typedef typename RowVec<VW>::type vec_t;
  const int p0 = threadIdx.x % PX, ty = threadIdx.x / PX;
  if (ty >= RPB) return;
  const unsigned Ho = Hi << up, Wo = Wi << up, HWo = Ho * Wo;
  const int pieces = C / VW, padp = (kcols - TAPS * C) / VW;
  vec_t zero;
#pragma unroll
  for (int e = 0; e < VW; ++e) zero[e] = (bf16)0.f;
  for (long rr = (long)blockIdx.x * RPB + ty; rr < nrows; rr += (long)gridDim.x * RPB) {
    const unsigned r = (unsigned)(row0 + rr);
    const unsigned b = r / HWo, rem = r - b * HWo, yo = rem / Wo, xo = rem - yo * Wo;
    bf16* orow = out + rr * ldo;
    const float* xb = x + (long)b * Hi * Wi * C;
    for (int p = p0; p < pieces; p += PX) {
      const int c = p * VW;
      alignas(16) float mu[VW], sc[VW], sh[VW];
      if (table) {
        const float* t = table + (long)b * 3 * C + c;
#pragma unroll
        for (int e = 0; e < VW; e += 4) {
          *(f32x4*)(mu + e) = *(const f32x4*)(t + e);
          *(f32x4*)(sc + e) = *(const f32x4*)(t + C + e);
          *(f32x4*)(sh + e) = *(const f32x4*)(t + 2 * C + e);
        }
      }
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int yy = (int)yo + (TAPS == 9 ? tap / 3 - 1 : 0), xx = (int)xo + (TAPS == 9 ? tap % 3 - 1 : 0);
        vec_t o = zero;
        if ((unsigned)yy < Ho && (unsigned)xx < Wo) {
          const float* src = xb + ((long)(yy >> up) * Wi + (xx >> up)) * C + c;
          alignas(16) float v[VW];
#pragma unroll
          for (int e = 0; e < VW; e += 4) *(f32x4*)(v + e) = *(const f32x4*)(src + e);
          if (table) {
#pragma unroll
            for (int e = 0; e < VW; ++e) v[e] = fmaf(v[e] - mu[e], sc[e], sh[e]);
          }
          if (silu) {
#pragma unroll
            for (int e = 0; e < VW; ++e) v[e] = silu_f(v[e]);
          }
#pragma unroll
          for (int e = 0; e < VW; ++e) o[e] = f2bf(v[e]);
        }
        *(vec_t*)(orow + tap * C + c) = o;
      }
    }
    for (int q = p0; q < padp; q += PX) *(vec_t*)(orow + TAPS * C + q * VW) = zero;
  }
}

// ---- attention probabilities ----------------------------------------------------------------------------------------------
// one wave per row: p = softmax(scale * s) in fp32 (max-subtracted, expf), rounded to the operand type
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, long lds_, bf16* __restrict__ p,
                                                           long ldp, int rows, int cols, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* sr = s + (long)row * lds_;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, sr[c] * scale);
  m = wave_max(m);
  float z = 0.f;
  for (int c = lane; c < cols; c += 64) z += expf(sr[c] * scale - m);
  z = wave_sum(z);
  const float inv = 1.f / z;
  bf16* pr = p + (long)row * ldp;
  for (int c = lane; c < cols; c += 64) pr[c] = f2bf(expf(sr[c] * scale - m) * inv);
}

constexpr long GN_ROWS_MIN = 64;   // rows per chunk at least; at most 256 chunks per image

int gn_chunks(long hw, long* rows_per_chunk) {
  long rpc = (hw + 255) / 256;
  if (rpc < GN_ROWS_MIN) rpc = GN_ROWS_MIN;
  *rows_per_chunk = rpc;
  return (int)((hw + rpc - 1) / rpc);
}

}  // namespace

extern "C" int64_t reed_groupnorm_ws_doubles(int B, int64_t hw, int C) {
  long rpc;
  return (int64_t)B * gn_chunks(hw, &rpc) * C * 2;
}

extern "C" int reed_groupnorm_stats(const float* x, int B, int64_t hw, int C, int G, float eps, const float* gamma,
                                    const float* beta, double* ws, float* stats, float* table, void* stream) {
  REED_CHECK_ARG(B > 0 && hw > 0 && C > 0 && G > 0, "reed_groupnorm_stats: empty problem");
  REED_CHECK_ARG(C % 4 == 0 && C <= 1024 && C % G == 0, "reed_groupnorm_stats: C=%d must be a multiple of 4 and of G=%d, at most 1024", C, G);
  REED_CHECK_ARG(((uintptr_t)x % 16) == 0 && ws && (stats || table), "reed_groupnorm_stats: x must be 16-byte aligned; ws and an output required");
  REED_CHECK_ARG(!table || (gamma && beta), "reed_groupnorm_stats: the per-channel table needs gamma and beta");
  long rpc;
  const int nchunk = gn_chunks(hw, &rpc);
  REED_KLAUNCH(gn_partial_kernel, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, x, (long)hw, C, rpc, ws);
  REED_LAUNCH_CHECK();
  REED_KLAUNCH(gn_final_kernel, dim3(cdiv((long)B * G, 4)), dim3(256), 0, (hipStream_t)stream, ws, B, (long)hw, C, G, nchunk, eps,
               gamma, beta, stats, table);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

namespace {
template <int TAPS, int VW>
void launch_rows(const float* x, const float* table, int Hi, int Wi, int C, int silu, int up, long row0, long nrows, int kcols,
                 bf16* out, long ldo, hipStream_t stream) {
  const int pieces = C / VW, PX = pieces < 256 ? pieces : 256, RPB = 256 / PX;
  long blocks = (nrows + RPB - 1) / RPB;
  if (blocks > (1 << 20)) blocks = 1 << 20;
  REED_KLAUNCH((conv_rows_kernel<TAPS, VW>), dim3((unsigned)blocks), dim3(256), 0, stream, x, table, Hi, Wi, C, silu, up, row0, nrows,
               kcols, out, ldo, PX, RPB);
}
}  // namespace

extern "C" int reed_conv_rows(const float* x, const float* table, int B, int Hi, int Wi, int C, int silu, int upsample, int taps,
                              int64_t row0, int64_t nrows, int kcols, void* out, int64_t ldo, void* stream) {
  REED_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && C > 0 && nrows > 0, "reed_conv_rows: empty problem");
  REED_CHECK_ARG(taps == 1 || taps == 9, "reed_conv_rows: taps=%d (1: 1x1 / Linear rows, 9: 3x3 with padding 1)", taps);
  REED_CHECK_ARG(upsample == 0 || upsample == 1, "reed_conv_rows: upsample must be 0 or 1 (nearest x2)");
  REED_CHECK_ARG(C % 4 == 0 && kcols % 4 == 0 && kcols >= taps * C && ldo >= kcols && ldo % 4 == 0,
                 "reed_conv_rows: C=%d, kcols=%d, ldo=%ld must be multiples of 4 with taps*C <= kcols <= ldo", C, kcols, (long)ldo);
  const long total = (long)B * (Hi << upsample) * (Wi << upsample);
  REED_CHECK_ARG(total < (1l << 31), "reed_conv_rows: %ld output positions (32-bit row arithmetic): pass fewer images", total);
  REED_CHECK_ARG(row0 >= 0 && row0 + nrows <= total, "reed_conv_rows: rows [%ld, %ld) outside the %ld output positions", (long)row0,
                 (long)(row0 + nrows), total);
  REED_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % (4 * sizeof(bf16))) == 0 && (!table || (uintptr_t)table % 16 == 0),
                 "reed_conv_rows: x / table must be 16-byte aligned, out aligned to 4 operand elements");
  // 16-bit operands: 8 channels per thread (16-byte stores) when every row piece is 16-byte aligned
  const bool wide = sizeof(bf16) == 2 && C % 8 == 0 && kcols % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)out % 16) == 0;
  hipStream_t st = (hipStream_t)stream;
  (void)hipGetLastError();
  if (taps == 9) {
    if (wide) launch_rows<9, 8>(x, table, Hi, Wi, C, silu, upsample, row0, nrows, kcols, (bf16*)out, ldo, st);
    else launch_rows<9, 4>(x, table, Hi, Wi, C, silu, upsample, row0, nrows, kcols, (bf16*)out, ldo, st);
  } else {
    if (wide) launch_rows<1, 8>(x, table, Hi, Wi, C, silu, upsample, row0, nrows, kcols, (bf16*)out, ldo, st);
    else launch_rows<1, 4>(x, table, Hi, Wi, C, silu, upsample, row0, nrows, kcols, (bf16*)out, ldo, st);
  }
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_softmax_rows(const float* s, int64_t lds, void* p, int64_t ldp, int rows, int cols, float scale,
                                 void* stream) {
  REED_CHECK_ARG(rows > 0 && cols > 0 && lds >= cols && ldp >= cols, "reed_softmax_rows: rows=%d cols=%d lds=%ld ldp=%ld", rows, cols,
                 (long)lds, (long)ldp);
  REED_KLAUNCH(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, s, (long)lds, (bf16*)p, (long)ldp, rows,
               cols, scale);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
