// qk_norm row kernels (split out of attention.hip so that every build of the library — bf16, fp16 and the fp32-operand build,
// whose attention lives in attention_f32.hip — compiles the same source).
#include "../../include/reed_hip.h"
#include "common.hpp"

// qk_norm (timm Attention(qk_norm=True), reference flag --qk-norm, image/models/sit.py:114-116): LayerNorm over
// head_dim (eps 1e-5, affine) on q and k before the attention product.  Under bf16 autocast the norm runs in fp32
// on the bf16 q/k and its output is cast back to bf16 by SDPA: qn = bf16(LN(float(q)) * w + b).
// One thread owns one (token, q|k, head) segment of hd elements (8 or 9 16-byte chunks): the statistics are
// thread-local, no cross-lane traffic.  v is copied through so the attention kernels keep their single-buffer layout.
namespace {

template <int HD>
__global__ __launch_bounds__(256) void qk_norm_fwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ qw,
                                                          const float* __restrict__ qb, const float* __restrict__ kw,
                                                          const float* __restrict__ kb, bf16* __restrict__ out,
                                                          float* __restrict__ stats, long nseg, int H, float eps) {
  constexpr int NCH = HD / 8;
  const long seg = (long)blockIdx.x * 256 + threadIdx.x;  // (token, which in {q,k,v}, head)
  if (seg >= nseg) return;
  const int h = (int)(seg % H), which = (int)((seg / H) % 3);
  const bf16* src = qkv + seg * HD;
  bf16* dst = out + seg * HD;
  bf16x8 v[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) v[c] = *(const bf16x8*)(src + c * 8);
  if (which == 2) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) *(bf16x8*)(dst + c * 8) = v[c];
    return;
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += bf2f(v[c][j]);
  const float mu = s / HD;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) { float d = bf2f(v[c][j]) - mu; q += d * d; }
  const float r = rsqrtf(q / HD + eps);
  const float* w = which == 0 ? qw : kw;
  const float* b = which == 0 ? qb : kb;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf((bf2f(v[c][j]) - mu) * r * w[c * 8 + j] + b[c * 8 + j]);
    *(bf16x8*)(dst + c * 8) = o;
  }
  if (stats) {
    const long tok = seg / (3 * H);
    stats[((tok * 2 + which) * H + h) * 2 + 0] = mu;
    stats[((tok * 2 + which) * H + h) * 2 + 1] = r;
  }
}

// backward: dpre = LNbwd(dn * w) for q,k (v copied); per-block partial sums of dw = sum dn*xhat, db = sum dn.
// Deterministic: per element a fixed butterfly reduction over the wave, then an ordered sum over the 4 waves.
template <int HD>
__global__ __launch_bounds__(256) void qk_norm_bwd_kernel(const bf16* __restrict__ dn, const bf16* __restrict__ qkv,
                                                          const float* __restrict__ stats, const float* __restrict__ qw,
                                                          const float* __restrict__ kw, bf16* __restrict__ dpre,
                                                          float* __restrict__ part, long nseg, int H) {
  constexpr int NCH = HD / 8;
  extern __shared__ float red[];  // [4 waves][2 (q,k)][2 (dw,db)][HD]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long seg = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = seg < nseg;
  const long sg = live ? seg : 0;
  const int h = (int)(sg % H), which = live ? (int)((sg / H) % 3) : 2;
  const bf16* gsrc = dn + sg * HD;
  const bf16* xsrc = qkv + sg * HD;
  bf16x8 gv[NCH], xv[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) { gv[c] = *(const bf16x8*)(gsrc + c * 8); xv[c] = *(const bf16x8*)(xsrc + c * 8); }
  const long tok = sg / (3 * H);
  const int wq = which < 2 ? which : 0;
  const float mu = stats[((tok * 2 + wq) * H + h) * 2 + 0], r = stats[((tok * 2 + wq) * H + h) * 2 + 1];
  const float* w = which == 1 ? kw : qw;
  const float isq = (live && which == 0) ? 1.f : 0.f, isk = (live && which == 1) ? 1.f : 0.f;
  float a1 = 0.f, a2 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = c * 8 + j;
      const float g = bf2f(gv[c][j]);
      const float xh = (bf2f(xv[c][j]) - mu) * r;
      const float gy = g * w[e];
      a1 += gy;
      a2 += gy * xh;
      const float s0 = wave_sum(isq * g * xh), s1 = wave_sum(isq * g), s2 = wave_sum(isk * g * xh), s3 = wave_sum(isk * g);
      if (lane == 0) {
        float* rw = red + wave * 4 * HD;
        rw[0 * HD + e] = s0; rw[1 * HD + e] = s1; rw[2 * HD + e] = s2; rw[3 * HD + e] = s3;
      }
    }
  a1 /= HD;
  a2 /= HD;
  if (live) {
    bf16* dst = dpre + seg * HD;
    if (which == 2) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) *(bf16x8*)(dst + c * 8) = gv[c];
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = c * 8 + j;
          const float xh = (bf2f(xv[c][j]) - mu) * r;
          o[j] = f2bf(r * (bf2f(gv[c][j]) * w[e] - a1 - xh * a2));
        }
        *(bf16x8*)(dst + c * 8) = o;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * HD; i += 256)
    part[(long)blockIdx.x * 4 * HD + i] = (red[i] + red[4 * HD + i]) + (red[8 * HD + i] + red[12 * HD + i]);
}

}  // namespace

extern "C" int reed_qk_norm_fwd(const void* qkv, const float* qw, const float* qb, const float* kw, const float* kb,
                                void* out, float* stats, int M, int H, int hd, float eps, void* stream) {
  REED_CHECK_ARG(qkv && qw && qb && kw && kb && out, "qk_norm_fwd: null pointer");
  REED_CHECK_ARG(hd == 64 || hd == 72, "qk_norm: head_dim %d unsupported (64 or 72)", hd);
  const long nseg = (long)M * 3 * H;
  if (hd == 64) REED_KLAUNCH(qk_norm_fwd_kernel<64>, dim3(cdiv(nseg, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv, qw, qb, kw, kb, (bf16*)out, stats, nseg, H, eps);
  else REED_KLAUNCH(qk_norm_fwd_kernel<72>, dim3(cdiv(nseg, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv, qw, qb, kw, kb, (bf16*)out, stats, nseg, H, eps);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int64_t reed_qk_norm_bwd_part_floats(int M, int H, int hd) { return (int64_t)cdiv((long)M * 3 * H, 256) * 4 * hd; }

extern "C" int reed_qk_norm_bwd(const void* dn, const void* qkv, const float* stats, const float* qw, const float* kw,
                                void* dpre, float* part, int M, int H, int hd, void* stream) {
  REED_CHECK_ARG(dn && qkv && stats && qw && kw && dpre && part, "qk_norm_bwd: null pointer");
  REED_CHECK_ARG(hd == 64 || hd == 72, "qk_norm: head_dim %d unsupported (64 or 72)", hd);
  const long nseg = (long)M * 3 * H;
  if (hd == 64) REED_KLAUNCH(qk_norm_bwd_kernel<64>, dim3(cdiv(nseg, 256)), dim3(256), 16 * 64 * sizeof(float), (hipStream_t)stream, (const bf16*)dn, (const bf16*)qkv, stats, qw, kw, (bf16*)dpre, part, nseg, H);
  else REED_KLAUNCH(qk_norm_bwd_kernel<72>, dim3(cdiv(nseg, 256)), dim3(256), 16 * 72 * sizeof(float), (hipStream_t)stream, (const bf16*)dn, (const bf16*)qkv, stats, qw, kw, (bf16*)dpre, part, nseg, H);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
