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

// XF / OF: the row is fp32 (the ViT towers' residual stream under autocast) or bf16 (CLIP's); the result bf16 (the next
// linear's operand) or fp32 (a tower's final norm: F.layer_norm returns fp32 under autocast)
template <bool XF, bool OF>
__global__ __launch_bounds__(256) void ln_affine_kernel(const void* __restrict__ xv, const float* __restrict__ w,
                                                        const float* __restrict__ bsh, void* __restrict__ outv,
                                                        int M, int D, long ldo, float eps) {
  // one wave per row; lanes own interleaved 8-element chunks
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63, nch = D >> 3;
  float v[LN_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) {
    const int c = lane + 64 * k;
    const bool ok = c < nch;
    if constexpr (XF) {
      const float* xr = (const float*)xv + (long)row * D + (ok ? c : 0) * 8;
      const f32x4 t0 = *(const f32x4*)xr, t1 = *(const f32x4*)(xr + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[k][j] = ok ? t0[j] : 0.f; v[k][4 + j] = ok ? t1[j] : 0.f; }
    } else {
      bf16x8 t = *(const bf16x8*)((const bf16*)xv + (long)row * D + (ok ? c : 0) * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[k][j] = ok ? bf2f(t[j]) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[k][j];
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
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < nch) {
      float y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = (v[k][j] - mu) * r * w[c * 8 + j] + bsh[c * 8 + j];
      if constexpr (OF) {
        float* orow = (float*)outv + (long)row * ldo + c * 8;
        *(f32x4*)orow = f32x4{y[0], y[1], y[2], y[3]};
        *(f32x4*)(orow + 4) = f32x4{y[4], y[5], y[6], y[7]};
      } else {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(y[j]);
        *(bf16x8*)((bf16*)outv + (long)row * ldo + c * 8) = o;
      }
    }
  }
}

// ViT tower token assembly (timm VisionTransformer._pos_embed / I-JEPA forward): fp32 residual stream
//   out[b, t] = prefix[t] + pos[t] for the ncls prefix rows (class token; DINOv2: class token + register tokens, whose pos rows
//   the caller zeroes); out[b, t] = float(patches[b, t - ncls]) + pos[t]
__global__ __launch_bounds__(256) void vit_tokens_kernel(const bf16* __restrict__ patches, const float* __restrict__ cls,
                                                         const float* __restrict__ pos, float* __restrict__ out, int B,
                                                         int T, int D, int ncls) {
  const int nch = D >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * T * nch) return;
  const int ch = (int)(idx % nch);
  const long row = idx / nch;
  const int t = (int)(row % T);
  const long b = row / T;
  float v[8];
  if (t < ncls) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = cls[(long)t * D + ch * 8 + j];
  } else {
    const bf16x8 p = *(const bf16x8*)(patches + (b * (T - ncls) + t - ncls) * D + ch * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bf2f(p[j]);
  }
  float* o = out + row * D + ch * 8;
  const float* pr = pos + (long)t * D + ch * 8;
  *(f32x4*)o = f32x4{v[0] + pr[0], v[1] + pr[1], v[2] + pr[2], v[3] + pr[3]};
  *(f32x4*)(o + 4) = f32x4{v[4] + pr[4], v[5] + pr[5], v[6] + pr[6], v[7] + pr[7]};
}

// preprocess_raw_image (image/train.py:53-74): uint8 [B,3,R,R] -> fp32 [B,3,S,S].
//   order 0 ('clip'):            x / 255 -> bicubic -> (x - mean) / std
//   order 1 ('dinov2', 'jepa'):  x / 255 -> (x - mean) / std -> bicubic
//   S == R ('mocov3', 'mae', 'dinov1'): no resampling, x / 255 -> normalise
// Bicubic = torch.nn.functional.interpolate(mode='bicubic', align_corners=False): source coordinate
// (dst + 0.5) * R / S - 0.5, 4 x 4 taps with the cubic convolution kernel A = -0.75, indices clamped to the image.
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
  const float A = -0.75f;
  float x = t + 1.f;
  c[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;
  c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;
  c[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;
  c[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}
__global__ __launch_bounds__(256) void preprocess_image_kernel(const uint8_t* __restrict__ raw, float* __restrict__ out,
                                                               int B, int R, int S, float m0, float m1, float m2,
                                                               float s0, float s1, float s2, int order) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * 3 * S * S) return;
  const int ox = (int)(idx % S), oy = (int)((idx / S) % S), ch = (int)((idx / ((long)S * S)) % 3);
  const long b = idx / ((long)3 * S * S);
  const float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2), std = ch == 0 ? s0 : (ch == 1 ? s1 : s2);
  const uint8_t* img = raw + (b * 3 + ch) * (long)R * R;
  auto px = [&](int y, int x) {
    const float v = (float)img[(long)y * R + x] / 255.f;
    return order == 1 ? (v - mean) / std : v;
  };
  float r;
  if (S == R) {
    r = px(oy, ox);
    if (order != 1) r = (r - mean) / std;
  } else {
    const float scale = (float)R / (float)S;
    const float ry = scale * (oy + 0.5f) - 0.5f, rx = scale * (ox + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    float cy[4], cx[4];
    cubic_coeffs(ry - iy, cy);
    cubic_coeffs(rx - ix, cx);
    r = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = min(max(iy - 1 + i, 0), R - 1);
      float row = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) row += px(y, min(max(ix - 1 + j, 0), R - 1)) * cx[j];
      r += row * cy[i];
    }
    if (order != 1) r = (r - mean) / std;
  }
  out[idx] = r;
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
  REED_KLAUNCH((ln_affine_kernel<false, false>), dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, w, b, out, M, D,
               (long)D, eps);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_ln_affine_f32(const float* x, const float* w, const float* b, void* out, int out_is_f32, int M, int D,
                                  int64_t ldo, float eps, void* stream) {
  REED_CHECK_ARG(x && w && b && out && M > 0, "ln_affine_f32: bad args");
  REED_CHECK_ARG(D % 8 == 0 && D <= 512 * LN_MAXC && ldo >= D && ldo % 8 == 0,
                 "ln_affine_f32: D=%d ldo=%ld unsupported (multiples of 8, D <= %d)", D, (long)ldo, 512 * LN_MAXC);
  if (out_is_f32)
    REED_KLAUNCH((ln_affine_kernel<true, true>), dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, w, b, out, M, D,
                 (long)ldo, eps);
  else
    REED_KLAUNCH((ln_affine_kernel<true, false>), dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, w, b, out, M, D,
                 (long)ldo, eps);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_vit_tokens(const void* patches, const float* cls, int nprefix, const float* pos, float* out, int B, int T,
                               int D, void* stream) {
  REED_CHECK_ARG(patches && pos && out && B > 0 && nprefix >= 0 && (cls || nprefix == 0) && T > nprefix && D % 8 == 0,
                 "vit_tokens: bad args");
  const long n = (long)B * T * (D / 8);
  REED_KLAUNCH(vit_tokens_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)patches, cls, pos,
               out, B, T, D, nprefix);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_preprocess_image(const uint8_t* raw, float* out, int B, int R, int S, const float* mean3,
                                     const float* std3, int order, void* stream) {
  REED_CHECK_ARG(raw && out && mean3 && std3 && B > 0 && R > 0 && S > 0 && (order == 0 || order == 1),
                 "preprocess_image: bad args");
  const long n = (long)B * 3 * S * S;
  REED_KLAUNCH(preprocess_image_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, raw, out, B, R, S, mean3[0],
               mean3[1], mean3[2], std3[0], std3[1], std3[2], order);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
