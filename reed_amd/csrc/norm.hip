// HBM-bound row kernels around the GEMMs of a SiT block (image/models/sit.py:26-27,113,119,130-137):
// LayerNorm(eps, no affine) + modulate forward/backward, the adaLN gate backward, the reduction of
// per-chunk column partials into bf16 modulation gradients, token mean (text projector tap) and casts.
// One wave (64 lanes) owns one token row; lanes own interleaved float4 columns, so every access is a
// 16-byte (f32) or 8-byte (bf16) per-lane coalesced vector access.  D <= 1280, D % 4 == 0.
#include "../../include/reed_hip.h"
#include "common.hpp"

namespace {

constexpr int MAXV = 5;  // float4 per lane: D <= 64*4*5 = 1280

__device__ __forceinline__ f32x4 ld_bf4(const bf16* p) {
  bf16x4 v = *(const bf16x4*)p;
  return f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
}
__device__ __forceinline__ void st_bf4(bf16* p, f32x4 v) {
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
  *(bf16x4*)p = o;
}

// NT: the residual stream read with the non-temporal policy — for activations larger than the Infinity Cache, where the next reader
// of x (the gate + residual epilogue two kernels later) cannot find it cached anyway: + 0.25 % per step at b = 256 (three pairs:
// 1223.5 -> 1226.3 images/s), - 0.2 % at b = 32 where x (38 MB) stays cached; the host picks per call (M * D * 4 >= 256 MiB)
template <bool NT>
__global__ __launch_bounds__(256) void ln_mod_fwd_kernel(const float* __restrict__ x, const bf16* __restrict__ shift,
                                                         const bf16* __restrict__ scale, long ldmod,
                                                         bf16* __restrict__ h, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int M, int D, int T, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63, nv = D >> 2;
  const float* xr = x + (long)row * D;
  f32x4 v[MAXV];
  float s = 0.f;
  // unconditional loads on a clamped index (a bounds branch per load would serialise them: guide trap (c));
  // out-of-range lanes are masked arithmetically
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int idx = lane + 64 * k;
    const bool ok = idx < nv;
    if constexpr (NT) v[k] = __builtin_nontemporal_load((const f32x4*)(xr + (ok ? idx : 0) * 4));
    else v[k] = *(const f32x4*)(xr + (ok ? idx : 0) * 4);
    const float m = ok ? 1.f : 0.f;
    v[k] *= m;
    s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
  }
  bf16* hr = h + (long)row * D;
  if (scale == nullptr) {  // plain cast
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      int idx = lane + 64 * k;
      if (idx < nv) st_bf4(hr + idx * 4, v[k]);
    }
    return;
  }
  const float mu = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int idx = lane + 64 * k;
    if (idx < nv) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { float d = v[k][j] - mu; q += d * d; }
    }
  }
  const float r = rsqrtf(wave_sum(q) / D + eps);
  if (lane == 0 && mean) { mean[row] = mu; rstd[row] = r; }
  const bf16* sc = scale + (long)(row / T) * ldmod;
  const bf16* sh = shift + (long)(row / T) * ldmod;
  f32x4 av[MAXV], bv[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int idx = lane + 64 * k;
    int ic = idx < nv ? idx : 0;
    av[k] = ld_bf4(sc + ic * 4);
    bv[k] = ld_bf4(sh + ic * 4);
  }
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int idx = lane + 64 * k;
    if (idx < nv) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[k][j] - mu) * r * bfround(1.f + av[k][j]) + bv[k][j];
      st_bf4(hr + idx * 4, o);
    }
  }
}

// 16 rows per block (4 per wave); all 16 rows belong to one sample (T % 16 == 0).
// GATE: the gate backward of the NEXT branch in backward order (the one that reads the dx this kernel just finished)
// rides along — dg = bf16(dx_new); dy = bf16(dg*gate); partial sums of bf16(dg*y) and of dy — which saves re-reading
// the fp32 dx (b*T*D*4 bytes) in a separate pass.
template <bool GATE>
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const bf16* __restrict__ dh, const float* __restrict__ x,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const bf16* __restrict__ scale, long ldmod,
                                                         float* __restrict__ dx, float* __restrict__ part,
                                                         const bf16* __restrict__ y, const bf16* __restrict__ gate,
                                                         long ldgate, bf16* __restrict__ dy, float* __restrict__ part_g,
                                                         float* __restrict__ part_dy, int M, int D, int T) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][2][D]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nv = D >> 2;
  const int row0 = blockIdx.x * 16 + wave * 4;
  const long smp = (blockIdx.x * 16) / T;
  const bf16* sc = scale + smp * ldmod;
  f32x4 s1[MAXV], ps[MAXV], pq[MAXV];
  f32x4 gv[GATE ? MAXV : 1], pg[GATE ? MAXV : 1], pd[GATE ? MAXV : 1];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int idx = lane + 64 * k;
    ps[k] = f32x4{0, 0, 0, 0};
    pq[k] = f32x4{0, 0, 0, 0};
    if (GATE) { pg[k] = f32x4{0, 0, 0, 0}; pd[k] = f32x4{0, 0, 0, 0}; }
    if (idx < nv) {
      f32x4 a = ld_bf4(sc + idx * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s1[k][j] = bfround(1.f + a[j]);
      if (GATE) gv[k] = ld_bf4(gate + smp * ldgate + idx * 4);
    }
  }
  for (int rr = 0; rr < 4; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;
    const float mu = mean[row], r = rstd[row];
    f32x4 xh[MAXV], gy[MAXV], xin[MAXV], gin[MAXV];
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      int idx = lane + 64 * k;
      int ic = idx < nv ? idx : 0;
      xin[k] = *(const f32x4*)(x + (long)row * D + ic * 4);
      gin[k] = ld_bf4(dh + (long)row * D + ic * 4);
    }
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      int idx = lane + 64 * k;
      if (idx < nv) {
        f32x4 xv = xin[k];
        f32x4 g = gin[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[k][j] = (xv[j] - mu) * r;
          gy[k][j] = g[j] * s1[k][j];
          a1 += gy[k][j];
          a2 += gy[k][j] * xh[k][j];
          ps[k][j] += g[j];
          pq[k][j] += g[j] * xh[k][j];
        }
      }
    }
    f32x4 dold[MAXV], yin[GATE ? MAXV : 1];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      int idx = lane + 64 * k;
      dold[k] = *(const f32x4*)(dx + (long)row * D + (idx < nv ? idx : 0) * 4);
      if (GATE) yin[k] = ld_bf4(y + (long)row * D + (idx < nv ? idx : 0) * 4);
    }
    a1 = wave_sum(a1) / D;
    a2 = wave_sum(a2) / D;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      int idx = lane + 64 * k;
      if (idx < nv) {
        float* dp = dx + (long)row * D + idx * 4;
        f32x4 o = dold[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += r * (gy[k][j] - a1 - xh[k][j] * a2);
        *(f32x4*)dp = o;
        if (GATE) {
          f32x4 oy;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float dg = bfround(o[j]);
            oy[j] = dg * gv[k][j];
            pg[k][j] += bfround(dg * yin[k][j]);
            pd[k][j] += bfround(oy[j]);
          }
          st_bf4(dy + (long)row * D + idx * 4, oy);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int idx = lane + 64 * k;
    if (idx < nv) {
      *(f32x4*)(red + (wave * 2 + 0) * D + idx * 4) = ps[k];
      *(f32x4*)(red + (wave * 2 + 1) * D + idx * 4) = pq[k];
    }
  }
  __syncthreads();
  float* out = part + (long)blockIdx.x * 2 * D;
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    out[i] = red[i] + red[2 * D + i] + red[4 * D + i] + red[6 * D + i];
  }
  if (GATE) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      int idx = lane + 64 * k;
      if (idx < nv) {
        *(f32x4*)(red + (wave * 2 + 0) * D + idx * 4) = pg[k];
        *(f32x4*)(red + (wave * 2 + 1) * D + idx * 4) = pd[k];
      }
    }
    __syncthreads();
    float* og = part_g + (long)blockIdx.x * D;
    for (int i = threadIdx.x; i < D; i += 256) og[i] = red[i] + red[2 * D + i] + red[4 * D + i] + red[6 * D + i];
    if (part_dy) {
      float* od = part_dy + (long)blockIdx.x * D;
      for (int i = threadIdx.x; i < D; i += 256) od[i] = red[D + i] + red[3 * D + i] + red[5 * D + i] + red[7 * D + i];
    }
  }
}

// Two waves per token row (D % 128 == 0): the one-wave-per-row kernel above needs 288 registers in its GATE form
// (18 floats of x, dh, dx, y per lane plus six per-column accumulators), i.e. ONE wave per SIMD — a quarter of a CU's
// load slots — and measured 4.3 TB/s at b = 256 and 2.5 TB/s at b = 32 (one round of 512 four-wave blocks).  Here a
// row's columns are split between a wave pair, per-lane state halves (<= 128 registers, four waves per SIMD) and the
// pair exchanges its two row sums (sum gy, sum gy*xhat) through LDS once per row.  Block = 8 waves = 4 row groups x 2
// halves, 16 rows per block as before, so the per-16-row column partials keep their layout and summation order.
// A lane owns NF float4 column groups (stride 64 groups) plus TS single columns of the half row's tail, so that no
// lane idles: D/128 = 4 NF + TS floats per lane (XL 1152: 2 + 1; L 1024: 2 + 0; B 768: 1 + 2; S 384: 0 + 3).
// All global accesses go through raw buffer descriptors: one SGPR descriptor per array (based at the block's first
// row), the row as a scalar byte offset, and 2 (NF + TS) per-lane column offsets shared by every array — with plain
// pointers the compiler keeps a 64-bit VGPR address per array and column group and the GATE form spills.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? bytes : 0u, 0x00020000);
}
template <int NF, int TS>
struct HalfRow {
  static constexpr int NE = 4 * NF + TS, NV = NF + TS;
  int vo[NV > 0 ? NV : 1];   // column (element) offset of each of the lane's vector / scalar accesses
  __device__ __forceinline__ HalfRow(int hf, int lane) {
    const int half0 = hf * 64 * NE;
#pragma unroll
    for (int k = 0; k < NF; ++k) vo[k] = half0 + (lane + 64 * k) * 4;
#pragma unroll
    for (int t = 0; t < TS; ++t) vo[NF + t] = half0 + 256 * NF + 64 * t + lane;
  }
  template <int AUX = 0>
  __device__ __forceinline__ void ld_f32(__amdgpu_buffer_rsrc_t rs, int soff, float (&v)[NE]) const {
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo[k] * 4, soff, AUX));
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * k + j] = t[j];
    }
#pragma unroll
    for (int t = 0; t < TS; ++t)
      v[4 * NF + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo[NF + t] * 4, soff, AUX));
  }
  template <int AUX = 0>
  __device__ __forceinline__ void st_f32(__amdgpu_buffer_rsrc_t rs, int soff, const float (&v)[NE]) const {
#pragma unroll
    for (int k = 0; k < NF; ++k)
      __builtin_amdgcn_raw_buffer_store_b128(
          __builtin_bit_cast(u32x4_t, f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]}), rs, vo[k] * 4, soff, AUX);
#pragma unroll
    for (int t = 0; t < TS; ++t)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[4 * NF + t]), rs, vo[NF + t] * 4, soff, AUX);
  }
#ifdef REED_FP32   // the "16-bit" arrays are fp32 arrays in this build (common.hpp): the same accesses as ld_f32 / st_f32
  template <int AUX = 0>
  __device__ __forceinline__ void ld_bf(__amdgpu_buffer_rsrc_t rs, int soff, float (&v)[NE]) const { ld_f32<AUX>(rs, soff, v); }
  template <int AUX = 0>
  __device__ __forceinline__ void st_bf(__amdgpu_buffer_rsrc_t rs, int soff, const float (&v)[NE]) const { st_f32<AUX>(rs, soff, v); }
#else
  template <int AUX = 0>
  __device__ __forceinline__ void ld_bf(__amdgpu_buffer_rsrc_t rs, int soff, float (&v)[NE]) const {   // bf16 -> f32
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      const bf16x4 t = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, vo[k] * 2, soff, AUX));
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * k + j] = bf2f(t[j]);
    }
#pragma unroll
    for (int t = 0; t < TS; ++t)
      v[4 * NF + t] = bf2f(__builtin_bit_cast(bf16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, vo[NF + t] * 2, soff, AUX)));
  }
  template <int AUX = 0>
  __device__ __forceinline__ void st_bf(__amdgpu_buffer_rsrc_t rs, int soff, const float (&v)[NE]) const {
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = f2bf(v[4 * k + j]);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, o), rs, vo[k] * 2, soff, AUX);
    }
#pragma unroll
    for (int t = 0; t < TS; ++t)
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, f2bf(v[4 * NF + t])), rs, vo[NF + t] * 2, soff, AUX);
  }
#endif
  // LDS (column partials): plain pointers
  __device__ __forceinline__ void st_lds(float* p, const float (&v)[NE]) const {
#pragma unroll
    for (int k = 0; k < NF; ++k) *(f32x4*)(p + vo[k]) = f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
#pragma unroll
    for (int t = 0; t < TS; ++t) p[vo[NF + t]] = v[4 * NF + t];
  }
};

#ifndef REED_LNB_NT
#define REED_LNB_NT 1
#endif
constexpr int LNB_AUX = REED_LNB_NT ? 2 : 0;
template <bool GATE, int NF, int TS>
__global__ __launch_bounds__(512, 4) void ln_mod_bwd2_kernel(
    const bf16* __restrict__ dh, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const bf16* __restrict__ scale, long ldmod, float* __restrict__ dx,
    float* __restrict__ part, const bf16* __restrict__ y, const bf16* __restrict__ gate, long ldgate,
    bf16* __restrict__ dy, float* __restrict__ part_g, float* __restrict__ part_dy, int M, int T) {
  using HR = HalfRow<NF, TS>;
  constexpr int NE = HR::NE, D = 128 * NE, HB = sizeof(bf16);   // HB: bytes per element of the 16-bit (fp32 build: 32-bit) arrays
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 groups][2][D], then row sums [4][4][2 halves][2]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row offsets stay in SGPRs
  const int rg = wave >> 1, hf = wave & 1;
  const HR hr(hf, lane);
  float* rsum = red + 8 * D;
  const long blk0 = (long)blockIdx.x * 16;      // first row of the block
  const long smp = blk0 / T;
  const __amdgpu_buffer_rsrc_t rsX = row_rsrc(x + blk0 * D, 16 * D * 4), rsDX = row_rsrc(dx + blk0 * D, 16 * D * 4),
                               rsDH = row_rsrc(dh + blk0 * D, 16 * D * HB),
                               rsY = row_rsrc(GATE ? y + blk0 * D : nullptr, 16 * D * HB),
                               rsDY = row_rsrc(GATE ? dy + blk0 * D : nullptr, 16 * D * HB);
  // per-sample constants: bf16(1 + scale) and the gate (bf16 values held as f32)
  float s1[NE], gv[GATE ? NE : 1], ps[NE], pq[NE], pg[GATE ? NE : 1], pd[GATE ? NE : 1];
  hr.ld_bf(row_rsrc(scale + smp * ldmod, D * HB), 0, s1);
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    s1[e] = bfround(1.f + s1[e]);
    ps[e] = 0.f;
    pq[e] = 0.f;
    if (GATE) { pg[e] = 0.f; pd[e] = 0.f; }
  }
  if constexpr (GATE) hr.ld_bf(row_rsrc(gate + smp * ldgate, D * HB), 0, gv);
#pragma unroll 1
  for (int rr = 0; rr < 4; ++rr) {
    const int lr = rg * 4 + rr;      // row inside the block; global row < M: the grid is M / 16 blocks, M % 16 == 0
    const float mu = mean[blk0 + lr], r = rstd[blk0 + lr];
    float xh[NE], gy[NE], o[NE], yin[GATE ? NE : 1];
    // REED_LNB_NT (round 4, default): the saved activations x and y (read once, from HBM) and the residual gradient (read and
    // re-written in place, next touched two GEMMs and an attention later) with the non-temporal policy; dh (the dgrad GEMM's output)
    // and dy (the next GEMMs' operand) stay cacheable.  Whole step at b = 256: 1269.9 -> 1279.3 images/s (three pairs), b = 32
    // unchanged (profiles/r4_row_kernel_nt.txt); bit-identical
    hr.template ld_f32<LNB_AUX>(rsX, lr * D * 4, xh);   // all of the row's loads first
    hr.ld_bf(rsDH, lr * D * HB, gy);
    hr.template ld_f32<LNB_AUX>(rsDX, lr * D * 4, o);
    if constexpr (GATE) hr.template ld_bf<LNB_AUX>(rsY, lr * D * HB, yin);
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const float g = gy[e];
      const float xv = (xh[e] - mu) * r;
      xh[e] = xv;
      gy[e] = g * s1[e];
      a1 += gy[e];
      a2 += gy[e] * xv;
      ps[e] += g;
      pq[e] += g * xv;
    }
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if (lane == 0) *(float2*)(rsum + (lr * 2 + hf) * 2) = make_float2(a1, a2);
    __syncthreads();
    {
      const f32x4 t = *(const f32x4*)(rsum + lr * 4);   // (a1, a2) of half 0, (a1, a2) of half 1
      a1 = (t[0] + t[2]) / D;
      a2 = (t[1] + t[3]) / D;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) o[e] += r * (gy[e] - a1 - xh[e] * a2);
    hr.template st_f32<LNB_AUX>(rsDX, lr * D * 4, o);
    if constexpr (GATE) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const float dg = bfround(o[e]);
        o[e] = dg * gv[e];
        pg[e] += bfround(dg * yin[e]);
        pd[e] += bfround(o[e]);
      }
      hr.st_bf(rsDY, lr * D * HB, o);
    }
  }
  // column partials of the 16 rows: per row group in registers, summed over the 4 groups in a fixed order
  hr.st_lds(red + (rg * 2 + 0) * D, ps);
  hr.st_lds(red + (rg * 2 + 1) * D, pq);
  __syncthreads();
  float* out = part + (long)blockIdx.x * 2 * D;
  for (int i = threadIdx.x; i < 2 * D; i += 512) out[i] = red[i] + red[2 * D + i] + red[4 * D + i] + red[6 * D + i];
  if constexpr (GATE) {
    __syncthreads();
    hr.st_lds(red + (rg * 2 + 0) * D, pg);
    hr.st_lds(red + (rg * 2 + 1) * D, pd);
    __syncthreads();
    float* og = part_g + (long)blockIdx.x * D;
    for (int i = threadIdx.x; i < D; i += 512) og[i] = red[i] + red[2 * D + i] + red[4 * D + i] + red[6 * D + i];
    if (part_dy) {   // column sums of dy (bias gradient of the linear that produced y) — optional: the wgrad GEMM can fuse it
      float* od = part_dy + (long)blockIdx.x * D;
      for (int i = threadIdx.x; i < D; i += 512) od[i] = red[D + i] + red[3 * D + i] + red[5 * D + i] + red[7 * D + i];
    }
  }
}

// D -> (NF, TS) instantiation; returns false when D has none (the caller falls back to the one-wave-per-row kernel)
template <bool GATE>
bool launch_ln_mod_bwd2(hipStream_t stream, const bf16* dh, const float* x, const float* mean, const float* rstd,
                        const bf16* scale, long ldmod, float* dx, float* part, const bf16* y, const bf16* gate,
                        long ldgate, bf16* dy, float* part_g, float* part_dy, int M, int D, int T) {
#define REED_LNB2(NF_, TS_)                                                                                          \
  if (D == 128 * (4 * NF_ + TS_)) {                                                                                   \
    REED_KLAUNCH((ln_mod_bwd2_kernel<GATE, NF_, TS_>), dim3(M / 16), dim3(512), (8 * D + 64) * sizeof(float),  \
                       stream, dh, x, mean, rstd, scale, ldmod, dx, part, y, gate, ldgate, dy, part_g, part_dy, M, T); \
    return true;                                                                                                      \
  }
  REED_LNB2(0, 1) REED_LNB2(0, 2) REED_LNB2(0, 3) REED_LNB2(1, 0) REED_LNB2(1, 2) REED_LNB2(2, 0) REED_LNB2(2, 1)
#undef REED_LNB2
  return false;
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dx, const bf16* __restrict__ y,
                                                       const bf16* __restrict__ gate, long ldgate,
                                                       bf16* __restrict__ dy, float* __restrict__ part,
                                                       float* __restrict__ part_dy, int M, int D, int T) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][D]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nv = D >> 2;
  const int row0 = blockIdx.x * 16 + wave * 4;
  const bf16* gp = gate + (long)((blockIdx.x * 16) / T) * ldgate;
  f32x4 gv[MAXV], pg[MAXV], pd[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int idx = lane + 64 * k;
    pg[k] = f32x4{0, 0, 0, 0};
    pd[k] = f32x4{0, 0, 0, 0};
    if (idx < nv) gv[k] = ld_bf4(gp + idx * 4);
  }
  for (int rr = 0; rr < 4; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;
    f32x4 din[MAXV], yin[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      int idx = lane + 64 * k;
      int ic = idx < nv ? idx : 0;
      din[k] = *(const f32x4*)(dx + (long)row * D + ic * 4);
      yin[k] = ld_bf4(y + (long)row * D + ic * 4);
    }
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      int idx = lane + 64 * k;
      if (idx < nv) {
        f32x4 d = din[k];
        f32x4 yv = yin[k], o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float dg = bfround(d[j]);
          o[j] = dg * gv[k][j];
          pg[k][j] += bfround(dg * yv[j]);
          pd[k][j] += bfround(o[j]);
        }
        st_bf4(dy + (long)row * D + idx * 4, o);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int idx = lane + 64 * k;
    if (idx < nv) *(f32x4*)(red + wave * D + idx * 4) = pg[k];
  }
  __syncthreads();
  float* out = part + (long)blockIdx.x * D;
  for (int i = threadIdx.x; i < D; i += 256) out[i] = red[i] + red[D + i] + red[2 * D + i] + red[3 * D + i];
  if (part_dy) {  // column sums of dy over the chunk: the bias gradient of the linear that produced y
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      int idx = lane + 64 * k;
      if (idx < nv) *(f32x4*)(red + wave * D + idx * 4) = pd[k];
    }
    __syncthreads();
    float* o2 = part_dy + (long)blockIdx.x * D;
    for (int i = threadIdx.x; i < D; i += 256) o2[i] = red[i] + red[D + i] + red[2 * D + i] + red[3 * D + i];
  }
}

// dst[C,R] = src[R,C]^T, bf16, 64x64 tiles through LDS (both sides 128-byte coalesced)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int R,
                                                             int C) {
  __shared__ bf16 t[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4)
    if (r0 + i < R && c0 + tx < C) t[i][tx] = src[(long)(r0 + i) * C + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 64; i += 4)
    if (c0 + i < C && r0 + tx < R) dst[(long)(c0 + i) * R + r0 + tx] = t[tx][i];
}

// The same for R, C multiples of 64 with 8-byte global accesses (round 6: the transposed weight copies of the input gradients' NT
// GEMMs are rebuilt every step behind the optimiser pass — 1.8 GB per step of SiT-XL/2): a lane loads four consecutive columns of a
// row (16 lanes = one 128-byte row segment), the tile sits in LDS with 33-word rows, and a lane gathers four consecutive ROWS of one
// column (rows 4 rr .. 4 rr + 3: banks 4 rr + c / 2, conflict-free) into one 8-byte store (16 lanes = 128 bytes of a dst row).
__global__ __launch_bounds__(256) void transpose_bf16_v4_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int R,
                                                                int C) {
  __shared__ unsigned short t[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int chunk = threadIdx.x + 256 * pass, row = chunk >> 4, cq = chunk & 15;
    const uint2 v = *(const uint2*)(src + (long)(r0 + row) * C + c0 + cq * 4);
    *(unsigned*)&t[row][cq * 4] = v.x;
    *(unsigned*)&t[row][cq * 4 + 2] = v.y;
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int idx = threadIdx.x + 256 * pass, c = idx >> 4, rr = idx & 15;
    uint2 v;
    v.x = (unsigned)t[rr * 4][c] | ((unsigned)t[rr * 4 + 1][c] << 16);
    v.y = (unsigned)t[rr * 4 + 2][c] | ((unsigned)t[rr * 4 + 3][c] << 16);
    *(uint2*)(dst + (long)(c0 + c) * R + r0 + rr * 4) = v;
  }
}

// out[n] (+)= sum over rows of an f32 [R, N] partial buffer (fixed order)
__global__ __launch_bounds__(256) void rowsum_f32_kernel(const float* __restrict__ part, int R, float* __restrict__ out,
                                                         int N, int accumulate) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  float s = 0.f;
  if (col < N)
    for (int z = grp; z < R; z += 4) s += part[(long)z * N + col];
  red[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0 && col < N) {
    float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    out[col] = accumulate ? out[col] + t : t;
  }
}

struct PartList {
  const float* ptr[8];
  long stride[8];
  long off[8];
  int n;
};
__global__ void reduce_mod_parts_kernel(PartList pl, bf16* __restrict__ dmod, long ld, int D, int chunks) {
  const int b = blockIdx.y, j = blockIdx.z;
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const float* p = pl.ptr[j] + (long)b * chunks * pl.stride[j] + d;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += p[(long)c * pl.stride[j]];
  dmod[(long)b * ld + pl.off[j] + d] = f2bf(s);
}
// the same sums in the same order, four columns per thread (16-byte loads) and 8 chunks in flight: 43 -> 2x us at b = 256, where the
// one-column form moved 113 MB at 2.6 TB/s (D, the strides and the offsets multiples of 4: the host checks)
__global__ __launch_bounds__(64) void reduce_mod_parts4_kernel(PartList pl, bf16* __restrict__ dmod, long ld, int D, int chunks) {
  const int b = blockIdx.y, j = blockIdx.z;
  const int d = (blockIdx.x * 64 + threadIdx.x) * 4;
  if (d >= D) return;
  const long st = pl.stride[j];
  const float* p = pl.ptr[j] + (long)b * chunks * st + d;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int c = 0;
  for (; c + 8 <= chunks; c += 8) {
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *(const f32x4*)(p + (long)(c + k) * st);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  for (; c < chunks; ++c) s += *(const f32x4*)(p + (long)c * st);
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = f2bf(s[e]);
  *(bf16x4*)(dmod + (long)b * ld + pl.off[j] + d) = o;
}

__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const float* __restrict__ x, bf16* __restrict__ out,
                                                             int T, int D) {
  const int b = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const float* p = x + (long)b * T * D + d;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += p[(long)t * D];
  out[(long)b * D + d] = f2bf(s / T);
}
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const bf16* __restrict__ dmean, float* __restrict__ dx,
                                                             int T, int D) {
  const int b = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const float g = bf2f(dmean[(long)b * D + d]) / T;
  float* p = dx + (long)b * T * D + d;
  for (int t = 0; t < T; ++t) p[(long)t * D] += g;
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ s, bf16* __restrict__ d, long n4,
                                                        long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; i < n4; i += stride) st_bf4(d + i * 4, *(const f32x4*)(s + i * 4));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    long k = (n4 << 2) + threadIdx.x;
    d[k] = f2bf(s[k]);
  }
}

// column sums of a bf16 [M,N] matrix (bias gradients): stage 1 = partial per 256-row slice (8 independent 8-byte loads
// in flight per thread), stage 2 = ordered reduce of the slices (deterministic)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16* __restrict__ x, long ld, float* __restrict__ part,
                                                             int M, int N) {
  const int col = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (col >= N) return;
  const int r0 = blockIdx.y * 256;
  f32x4 s = {0, 0, 0, 0};
  const bf16* base = x + (long)r0 * ld + col;
  if (r0 + 256 <= M) {  // whole slice in range: unconditional loads, 8 in flight (a per-load bounds select would
                       // make hipcc branch + vmcnt(0) around every load: cdna_hip_programming.md trap (c))
#pragma unroll 1
    for (int rr = 0; rr < 256; rr += 8) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ld_bf4(base + (long)(rr + k) * ld);
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
  } else {
    for (int r = r0; r < M; ++r) s += ld_bf4(x + (long)r * ld + col);
  }
  *(f32x4*)(part + (long)blockIdx.y * N + col) = s;
}
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, int nsl, float* __restrict__ out,
                                                            int N, int accumulate) {
  // one wave per 64 columns would under-fill; instead 4 lanes-groups over slices then a fixed-order tree
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  float s = 0.f;
  if (col < N)
    for (int z = grp; z < nsl; z += 4) s += part[(long)z * N + col];
  red[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0 && col < N) {
    float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    out[col] = accumulate ? out[col] + t : t;
  }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, long stride, int ns,
                                                           float* __restrict__ out, long n, int accumulate) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? out[i] : 0.f;
  for (int z = 0; z < ns; ++z) s += slabs[(long)z * stride + i];
  out[i] = s;
}

}  // namespace

extern "C" int reed_ln_modulate_fwd(const float* x, const void* shift, const void* scale, int64_t ldmod,
                                    void* h, float* mean, float* rstd, int M, int D, int T, float eps,
                                    void* stream) {
  REED_CHECK_ARG(x && h, "ln_modulate_fwd: null pointer");
  REED_CHECK_ARG(D % 4 == 0 && D <= 256 * MAXV, "ln_modulate: D=%d unsupported (multiple of 4, <= %d)", D, 256 * MAXV);
  REED_CHECK_ARG(M > 0 && T > 0, "ln_modulate: bad M=%d T=%d", M, T);
  REED_CHECK_ARG((scale == nullptr) == (shift == nullptr), "ln_modulate: shift and scale must both be given or both NULL");
  if ((long)M * D * 4 >= (256l << 20))
    REED_KLAUNCH(ln_mod_fwd_kernel<true>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, (const bf16*)shift,
                 (const bf16*)scale, (long)ldmod, (bf16*)h, mean, rstd, M, D, T, eps);
  else
    REED_KLAUNCH(ln_mod_fwd_kernel<false>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, (const bf16*)shift,
                 (const bf16*)scale, (long)ldmod, (bf16*)h, mean, rstd, M, D, T, eps);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_ln_modulate_bwd(const void* dh, const float* x, const float* mean, const float* rstd,
                                    const void* scale, int64_t ldmod, float* dx, float* part, int M, int D,
                                    int T, void* stream) {
  REED_CHECK_ARG(dh && x && mean && rstd && scale && dx && part, "ln_modulate_bwd: null pointer");
  REED_CHECK_ARG(D % 4 == 0 && D <= 256 * MAXV, "ln_modulate: D=%d unsupported", D);
  REED_CHECK_ARG(T % 16 == 0 && M % 16 == 0, "ln_modulate_bwd: T=%d, M=%d must be multiples of 16", T, M);
  if (launch_ln_mod_bwd2<false>((hipStream_t)stream, (const bf16*)dh, x, mean, rstd, (const bf16*)scale, (long)ldmod, dx,
                                part, nullptr, nullptr, 0l, nullptr, nullptr, nullptr, M, D, T)) {
    REED_LAUNCH_CHECK();
    return REED_OK;
  }
    REED_KLAUNCH(ln_mod_bwd_kernel<false>, dim3(M / 16), dim3(256), 8 * D * sizeof(float), (hipStream_t)stream,
                 (const bf16*)dh, x, mean, rstd, (const bf16*)scale, (long)ldmod, dx, part, (const bf16*)nullptr,
                 (const bf16*)nullptr, 0l, (bf16*)nullptr, (float*)nullptr, (float*)nullptr, M, D, T);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_ln_modulate_bwd_gate(const void* dh, const float* x, const float* mean, const float* rstd,
                                         const void* scale, int64_t ldmod, float* dx, float* part, const void* y,
                                         const void* gate, int64_t ldgate, void* dy, float* part_g, float* part_dy,
                                         int M, int D, int T, void* stream) {
  REED_CHECK_ARG(dh && x && mean && rstd && scale && dx && part, "ln_modulate_bwd_gate: null pointer");
  REED_CHECK_ARG(y && gate && dy && part_g, "ln_modulate_bwd_gate: null gate operand");   // part_dy is optional
  REED_CHECK_ARG(D % 4 == 0 && D <= 256 * MAXV, "ln_modulate: D=%d unsupported", D);
  REED_CHECK_ARG(T % 16 == 0 && M % 16 == 0, "ln_modulate_bwd_gate: T=%d, M=%d must be multiples of 16", T, M);
  if (launch_ln_mod_bwd2<true>((hipStream_t)stream, (const bf16*)dh, x, mean, rstd, (const bf16*)scale, (long)ldmod, dx,
                               part, (const bf16*)y, (const bf16*)gate, (long)ldgate, (bf16*)dy, part_g, part_dy, M, D, T)) {
    REED_LAUNCH_CHECK();
    return REED_OK;
  }
    REED_KLAUNCH(ln_mod_bwd_kernel<true>, dim3(M / 16), dim3(256), 8 * D * sizeof(float), (hipStream_t)stream,
                 (const bf16*)dh, x, mean, rstd, (const bf16*)scale, (long)ldmod, dx, part, (const bf16*)y,
                 (const bf16*)gate, (long)ldgate, (bf16*)dy, part_g, part_dy, M, D, T);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_transpose_bf16(const void* src, void* dst, int R, int C, void* stream) {
  REED_CHECK_ARG(src && dst && R > 0 && C > 0, "transpose_bf16: bad args");
  if (R % 64 == 0 && C % 64 == 0 && ((uintptr_t)src % 8) == 0 && ((uintptr_t)dst % 8) == 0)
    REED_KLAUNCH(transpose_bf16_v4_kernel, dim3(C / 64, R / 64), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (bf16*)dst, R, C);
  else
    REED_KLAUNCH(transpose_bf16_kernel, dim3(cdiv(C, 64), cdiv(R, 64)), dim3(256), 0, (hipStream_t)stream,
                 (const bf16*)src, (bf16*)dst, R, C);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

// stage 1 of a tall reduce: part2[s, n] = sum of 64 consecutive rows (thread per column: coalesced, 8 loads in flight)
__global__ __launch_bounds__(256) void rowsum_stage1_kernel(const float* __restrict__ part, int R, float* __restrict__ part2,
                                                            int N) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= N) return;
  const int r0 = blockIdx.y * 64, r1 = min(R, r0 + 64);
  float s = 0.f;
  int r = r0;
  for (; r + 8 <= r1; r += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = part[(long)(r + k) * N + col];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  for (; r < r1; ++r) s += part[(long)r * N + col];
  part2[(long)blockIdx.y * N + col] = s;
}

extern "C" int reed_rowsum_f32(const float* part, int R, float* ws, float* out, int N, int accumulate, void* stream) {
  REED_CHECK_ARG(part && out && R > 0 && N > 0, "rowsum_f32: bad args");
  if (R > 256 && ws) {  // tall: two stages (ws: cdiv(R,64)*N floats), both in a fixed order
    const int R2 = cdiv(R, 64);
    REED_KLAUNCH(rowsum_stage1_kernel, dim3(cdiv(N, 256), R2), dim3(256), 0, (hipStream_t)stream, part, R, ws, N);
    REED_KLAUNCH(rowsum_f32_kernel, dim3(cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream, ws, R2, out, N, accumulate);
  } else {
    REED_KLAUNCH(rowsum_f32_kernel, dim3(cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream, part, R, out, N, accumulate);
  }
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_gate_bwd(const float* dx, const void* y, const void* gate, int64_t ldgate, void* dy,
                             float* part, float* part_dy, int M, int D, int T, void* stream) {
  REED_CHECK_ARG(dx && y && gate && dy && part, "gate_bwd: null pointer");
  REED_CHECK_ARG(D % 4 == 0 && D <= 256 * MAXV, "gate_bwd: D=%d unsupported", D);
  REED_CHECK_ARG(T % 16 == 0 && M % 16 == 0, "gate_bwd: T=%d, M=%d must be multiples of 16", T, M);
  REED_KLAUNCH(gate_bwd_kernel, dim3(M / 16), dim3(256), 4 * D * sizeof(float), (hipStream_t)stream, dx,
                     (const bf16*)y, (const bf16*)gate, (long)ldgate, (bf16*)dy, part, part_dy, M, D, T);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_reduce_mod_parts(const float* const* parts, const int64_t* strides, const int64_t* offs,
                                     int nparts, void* dmod, int64_t lddmod, int B, int D, int chunks,
                                     void* stream) {
  REED_CHECK_ARG(nparts >= 1 && nparts <= 8, "reduce_mod_parts: nparts=%d out of range", nparts);
  PartList pl;
  pl.n = nparts;
  for (int i = 0; i < nparts; ++i) { pl.ptr[i] = parts[i]; pl.stride[i] = strides[i]; pl.off[i] = offs[i]; }
  bool four = (D % 4) == 0 && (lddmod % 4) == 0 && ((uintptr_t)dmod % 8) == 0;
  for (int i = 0; i < nparts; ++i) four = four && (strides[i] % 4) == 0 && (offs[i] % 4) == 0 && ((uintptr_t)parts[i] % 16) == 0;
  if (four)
    REED_KLAUNCH(reduce_mod_parts4_kernel, dim3(cdiv(D, 256), B, nparts), dim3(64), 0, (hipStream_t)stream, pl,
                 (bf16*)dmod, (long)lddmod, D, chunks);
  else
    REED_KLAUNCH(reduce_mod_parts_kernel, dim3(cdiv(D, 256), B, nparts), dim3(256), 0, (hipStream_t)stream, pl,
                 (bf16*)dmod, (long)lddmod, D, chunks);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_token_mean_fwd(const float* x, void* out, int B, int T, int D, void* stream) {
  REED_KLAUNCH(token_mean_fwd_kernel, dim3(cdiv(D, 256), B), dim3(256), 0, (hipStream_t)stream, x, (bf16*)out, T, D);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
extern "C" int reed_token_mean_bwd(const void* dmean, float* dx, int B, int T, int D, void* stream) {
  REED_KLAUNCH(token_mean_bwd_kernel, dim3(cdiv(D, 256), B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dmean, dx, T, D);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (n <= 0) return REED_OK;
  REED_CHECK_ARG(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, "cast_bf16: misaligned");
  long n4 = n >> 2;
  int blocks = (int)(n4 / 256 + 1);
  if (blocks > 4096) blocks = 4096;
  REED_KLAUNCH(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, n4, (long)n);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int64_t reed_colsum_ws_floats(int M, int N) { return (int64_t)cdiv(M, 256) * N; }

extern "C" int reed_colsum_bf16(const void* x, int64_t ld, float* ws, float* out, int M, int N, int accumulate,
                                void* stream) {
  REED_CHECK_ARG(x && ws && out && M > 0 && N > 0 && N % 4 == 0, "colsum_bf16: bad args");
  const int nsl = cdiv(M, 256);
  REED_KLAUNCH(colsum_partial_kernel, dim3(cdiv(N, 1024), nsl), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
               (long)ld, ws, M, N);
  REED_KLAUNCH(colsum_reduce_kernel, dim3(cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream, ws, nsl, out, N, accumulate);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_reduce_slabs(const float* slabs, int64_t stride, int nslabs, float* out, int64_t n,
                                 int accumulate, void* stream) {
  if (n <= 0) return REED_OK;
  REED_KLAUNCH(reduce_slabs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, slabs, (long)stride,
                     nslabs, out, (long)n, accumulate);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
