// Row kernels of the frozen CLIP image encoder's forward (reference: image/models/clip_vit.py:159-230, the
// UpdatedVisionTransformer wrapper train.py:351-357 runs under bf16 autocast to produce the alignment targets;
// SURVEY.md §8f N2).  Inference only.  The contractions (patch-embedding conv as a GEMM, in_proj / out_proj / c_fc /
// c_proj) run on the bf16 MFMA GEMM kernels with the QuickGELU and bf16-residual epilogues, the attention on
// attn_fwd_kernel<64> at T = 257; what is left are three HBM-bound passes:
//   im2col     normalised fp32 image -> bf16 patch rows (c, py, px), zero-padded to the GEMM's K multiple
//   tokens     [class token | patch embeddings] + positional embedding, bf16 (clip_vit.py:217-218)
//   ln_affine  LayerNorm(eps 1e-5, affine) computed in fp32 on a bf16 row, bf16 out (clip_vit.py:159-165)
#include "../../include/reed_hip.h"
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void clip_im2col_kernel(const float* __restrict__ img, bf16* __restrict__ out, int B,
                                                          int S, int P, int Kp) {
  // one thread per (row, 8-column chunk): row = (b, gy, gx); column k = c*P*P + py*P + px
  const int G = S / P, K = 3 * P * P, nch = Kp >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * G * G * nch) return;
  const int ch = (int)(idx % nch);
  const long row = idx / nch;
  const int gx = (int)(row % G), gy = (int)((row / G) % G), b = (int)(row / ((long)G * G));
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = ch * 8 + j;
    float x = 0.f;
    if (k < K) {
      const int c = k / (P * P), r = k - c * P * P, py = r / P, px = r - py * P;
      x = img[(((long)b * 3 + c) * S + gy * P + py) * S + gx * P + px];
    }
    v[j] = f2bf(x);
  }
  *(bf16x8*)(out + row * Kp + ch * 8) = v;
}

__global__ __launch_bounds__(256) void clip_tokens_kernel(const bf16* __restrict__ patches, const float* __restrict__ cls,
                                                          const float* __restrict__ pos, bf16* __restrict__ out, int B,
                                                          int T, int D) {
  // out[b, 0] = bf16(bf16(cls) + bf16(pos[0])); out[b, t] = bf16(patches[b, t-1] + bf16(pos[t]))   (T = patches + 1)
  const int nch = D >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * T * nch) return;
  const int ch = (int)(idx % nch);
  const long row = idx / nch;
  const int t = (int)(row % T);
  const long b = row / T;
  bf16x8 v;
  if (t == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f2bf(cls[ch * 8 + j]);
  } else {
    v = *(const bf16x8*)(patches + (b * (T - 1) + t - 1) * D + ch * 8);
  }
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(v[j]) + bfround(pos[(long)t * D + ch * 8 + j]));
  *(bf16x8*)(out + row * D + ch * 8) = o;
}

constexpr int LN_MAXC = 4;  // 16-byte chunks per lane: D <= 64 * 8 * 4 = 2048

__global__ __launch_bounds__(256) void ln_affine_bf16_kernel(const bf16* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bsh, bf16* __restrict__ out,
                                                             int M, int D, float eps) {
  // one wave per row; lanes own interleaved 8-element chunks
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63, nch = D >> 3;
  const bf16* xr = x + (long)row * D;
  float v[LN_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) {
    const int c = lane + 64 * k;
    const bool ok = c < nch;
    bf16x8 t = *(const bf16x8*)(xr + (ok ? c : 0) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[k][j] = ok ? bf2f(t[j]) : 0.f;
      s += v[k][j];
    }
  }
  const float mu = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k)
    if (lane + 64 * k < nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[k][j] - mu; q += d * d; }
    }
  const float r = rsqrtf(wave_sum(q) / D + eps);
  bf16* orow = out + (long)row * D;
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < nch) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf((v[k][j] - mu) * r * w[c * 8 + j] + bsh[c * 8 + j]);
      *(bf16x8*)(orow + c * 8) = o;
    }
  }
}

}  // namespace

extern "C" int reed_clip_im2col(const float* img, void* out, int B, int S, int P, int Kp, void* stream) {
  REED_CHECK_ARG(img && out && B > 0 && S > 0 && P > 0, "clip_im2col: bad args");
  REED_CHECK_ARG(S % P == 0 && Kp % 8 == 0 && Kp >= 3 * P * P, "clip_im2col: S=%d P=%d Kp=%d (S %% P == 0, Kp >= 3 P^2, Kp %% 8 == 0)", S, P, Kp);
  const long n = (long)B * (S / P) * (S / P) * (Kp / 8);
  REED_KLAUNCH(clip_im2col_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, img, (bf16*)out, B, S, P, Kp);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_clip_tokens(const void* patches, const float* cls, const float* pos, void* out, int B, int T,
                                int D, void* stream) {
  REED_CHECK_ARG(patches && cls && pos && out && B > 0 && T > 1 && D % 8 == 0, "clip_tokens: bad args");
  const long n = (long)B * T * (D / 8);
  REED_KLAUNCH(clip_tokens_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)patches, cls,
               pos, (bf16*)out, B, T, D);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_ln_affine_bf16(const void* x, const float* w, const float* b, void* out, int M, int D, float eps,
                                   void* stream) {
  REED_CHECK_ARG(x && w && b && out && M > 0, "ln_affine_bf16: bad args");
  REED_CHECK_ARG(D % 8 == 0 && D <= 512 * LN_MAXC, "ln_affine_bf16: D=%d unsupported (multiple of 8, <= %d)", D, 512 * LN_MAXC);
  REED_KLAUNCH(ln_affine_bf16_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, w, b,
               (bf16*)out, M, D, eps);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
