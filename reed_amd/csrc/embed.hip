// Embedders and the final layer of SiT (image/models/sit.py): PatchEmbed (:198-200,279, timm conv k=s=P),
// TimestepEmbedder sinusoid (:45-64), LabelEmbedder + conditioning vector (:84-99,283-285),
// FinalLayer + unpatchify (:140-158,256-269), and the small-K weight-gradient reduction they share.
// All of it is HBM-bound bookkeeping around the MFMA GEMMs; the index maps (patchify order (c,pi,pj) on
// the way in, (pi,pj,c) on the way out, row-major tokens) are bit-exact restatements of the reference.
#include "../../include/reed_hip.h"
#include "common.hpp"

namespace {

constexpr int MAXV = 5;
constexpr int NSL = 256;  // token slices of the two-stage small-K wgrad (one wave each x Dw/128 column blocks)

__device__ __forceinline__ f32x4 ld_bf4(const bf16* p) {
  bf16x4 v = *(const bf16x4*)p;
  return f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
}

// ---- patchify: out bf16 [B*T, K], K = C*P*P. order 0: k = (c,pi,pj) (conv input); 1: k = (pi,pj,c) (unpatchify) ----
__device__ __forceinline__ long patch_src(int b, int t, int k, int C, int HW, int P, int order) {
  const int G = HW / P, ph = t / G, pw = t - ph * G;
  int c, pi, pj;
  if (order == 0) { c = k / (P * P); int r = k - c * P * P; pi = r / P; pj = r - pi * P; }
  else { c = k % C; int r = k / C; pi = r / P; pj = r - pi * P; }
  return (((long)b * C + c) * HW + ph * P + pi) * HW + pw * P + pj;
}
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, bf16* __restrict__ out, int B,
                                                       int C, int HW, int P, int order) {
  const int G = HW / P, T = G * G, K = C * P * P;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * T * K) return;
  int k = (int)(i % K);
  long bt = i / K;
  out[i] = f2bf(x[patch_src((int)(bt / T), (int)(bt % T), k, C, HW, P, order)]);
}

// ---- patch embed forward: 8 tokens per block ----
__global__ __launch_bounds__(256) void patch_embed_fwd_kernel(const float* __restrict__ x, const bf16* __restrict__ w,
                                                              const bf16* __restrict__ bias,
                                                              const float* __restrict__ pos, float* __restrict__ tok,
                                                              int B, int C, int HW, int P, int D) {
  extern __shared__ float xs[];  // [8][K]
  const int G = HW / P, T = G * G, K = C * P * P;
  const long bt0 = (long)blockIdx.x * 8, BT = (long)B * T;
  for (int i = threadIdx.x; i < 8 * K; i += 256) {
    long bt = bt0 + i / K;
    xs[i] = bt < BT ? bfround(x[patch_src((int)(bt / T), (int)(bt % T), i % K, C, HW, P, 0)]) : 0.f;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bf16* wr = w + (long)d * K;
    for (int k = 0; k < K; k += 8) {
      bf16x8 wv = *(const bf16x8*)(wr + k);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float wf = bf2f(wv[j]);
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[s] += wf * xs[s * K + k + j];
      }
    }
    const float bv = bias ? bf2f(bias[d]) : 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      long bt = bt0 + s;
      if (bt < BT) tok[bt * D + d] = bfround(acc[s] + bv) + pos[(bt % T) * D + d];
    }
  }
}

// The same with the roles turned: a thread keeps the weight rows of ITS four adjacent columns d = 4 tid .. 4 tid + 3 in registers
// (K = 16 at patch 2: 64 floats) and walks PT tokens whose K inputs sit in LDS — 4 broadcast reads of 16 bytes, 64 FMAs, one
// 16-byte load of the positional row and ONE 16-byte store per token, where the kernel above re-reads every input of its 8 tokens
// from LDS for every column (128 ds_read_b32 per column) and stores 4 bytes per lane.  Same accumulation order over k:
// bit-identical tokens.  320 threads (288 of them active for D = 1152).
constexpr int PT = 64, PK = 16, PNT = 320;
__global__ __launch_bounds__(PNT) void patch_embed_fwd16_kernel(const float* __restrict__ x, const bf16* __restrict__ w,
                                                                const bf16* __restrict__ bias,
                                                                const float* __restrict__ pos, float* __restrict__ tok,
                                                                int B, int C, int HW, int P, int D) {
  __shared__ __attribute__((aligned(16))) float xs[PT * PK];
  const int G = HW / P, T = G * G;
  const long bt0 = (long)blockIdx.x * PT, BT = (long)B * T;
  for (int i = threadIdx.x; i < PT * PK; i += PNT) {
    long bt = bt0 + i / PK;
    xs[i] = bt < BT ? bfround(x[patch_src((int)(bt / T), (int)(bt % T), i % PK, C, HW, P, 0)]) : 0.f;
  }
  const int d0 = 4 * threadIdx.x;
  const bool act = d0 < D;                       // D % 4 == 0
  float wr[4][PK], bv[4];
  if (act) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bf16x8 w0 = *(const bf16x8*)(w + (long)(d0 + e) * PK), w1 = *(const bf16x8*)(w + (long)(d0 + e) * PK + 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { wr[e][j] = bf2f(w0[j]); wr[e][8 + j] = bf2f(w1[j]); }
      bv[e] = bias ? bf2f(bias[d0 + e]) : 0.f;
    }
  }
  __syncthreads();
  if (!act) return;
  for (int s_ = 0; s_ < PT; ++s_) {
    const long bt = bt0 + s_;
    if (bt >= BT) break;
    f32x4 xv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xv[q] = *(const f32x4*)(xs + s_ * PK + 4 * q);
    const f32x4 pv = *(const f32x4*)(pos + (bt % T) * (long)D + d0);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < PK; ++k) acc += wr[e][k] * xv[k >> 2][k & 3];
      o[e] = bfround(acc + bv[e]) + pv[e];
    }
    *(f32x4*)(tok + bt * D + d0) = o;
  }
}

// ---- small-K weight gradient (final layer: KS = 2*C*p*p = 32, patch embed: KS = C*p*p = 16 at patch 2), stage 1 ----
// out[d][k] = sum_m wide[m][d] * small[m][k] over a token slice; ws[slice] holds the partial out (layout 0: d*KS+k,
// 1: k*Dw+d), then the partial colsum(wide)[Dw], then the partial colsum(small)[KS].
// One wave per block: a lane owns 2 adjacent columns d of `wide` (4- or 8-byte loads: a wave reads a contiguous
// 256/512-byte row segment) and KSP accumulators for each; small[m][k0 .. k0+KSP) is the same for every lane and comes
// in as 16-byte broadcast loads.  No LDS, no barriers.  blockIdx.z walks KS in chunks of KSP (KS % 8 == 0).
template <bool WIDE_F32, int KSP>
__global__ __launch_bounds__(64) void smallk_wgrad_kernel(const void* __restrict__ wide, const bf16* __restrict__ small,
                                                          float* __restrict__ ws, int M, int Dw, int KS, int layout) {
  const int lane = threadIdx.x;
  const int d = blockIdx.x * 128 + 2 * lane;
  const int slice = blockIdx.y;
  const int k0 = blockIdx.z * KSP;
  const int per = (M + NSL - 1) / NSL;
  const int mbeg = slice * per, mend = min(M, mbeg + per);
  float* wout = ws + (long)slice * ((long)KS * Dw + Dw + KS);
  const bool dok = d < Dw;   // Dw is even
  float acc0[KSP], acc1[KSP];
#pragma unroll
  for (int k = 0; k < KSP; ++k) acc0[k] = acc1[k] = 0.f;
  float ws0 = 0.f, ws1 = 0.f;
  const int dc = dok ? d : 0;
  const int nch = min(KSP, KS - k0) >> 3;   // valid 8-element chunks of this block's k range (block-uniform)
  // rows in groups of RG: all loads of a group are issued before its FMAs (a one-wave block has nothing else to hide
  // the load latency behind)
  constexpr int RG = KSP == 16 ? 8 : 4;
  // (the row of `small` is wave-uniform; an opaque zero keeps its address in a VGPR so that the 16-byte broadcast
  // loads land in VGPRs: as scalar loads RG rows would need 64-128 SGPRs and spill)
  int vz = 0;
  asm volatile("" : "+v"(vz));
  for (int m = mbeg; m < mend; m += RG) {
    float w0[RG], w1[RG];
    bf16x8 sv[RG][KSP / 8];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      const int mr = min(m + r, mend - 1);          // clamped: the tail rows are masked to zero below
      if (WIDE_F32) {
        const float2 v = *(const float2*)((const float*)wide + (long)mr * Dw + dc);
        w0[r] = bfround(v.x); w1[r] = bfround(v.y);
      } else {
        const bf16x2 v = *(const bf16x2*)((const bf16*)wide + (long)mr * Dw + dc);
        w0[r] = bf2f(v[0]); w1[r] = bf2f(v[1]);
      }
      if (m + r >= mend) w0[r] = w1[r] = 0.f;
      const bf16* srow = small + (long)mr * KS + k0 + vz;
#pragma unroll
      for (int c = 0; c < KSP / 8; ++c) {
        if (c < nch) sv[r][c] = *(const bf16x8*)(srow + 8 * c);
        else
#pragma unroll
          for (int j = 0; j < 8; ++j) sv[r][c][j] = (bf16)0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      ws0 += w0[r]; ws1 += w1[r];
#pragma unroll
      for (int c = 0; c < KSP / 8; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = bf2f(sv[r][c][j]);
          acc0[8 * c + j] = fmaf(w0[r], f, acc0[8 * c + j]);
          acc1[8 * c + j] = fmaf(w1[r], f, acc1[8 * c + j]);
        }
    }
  }
  if (dok) {
#pragma unroll
    for (int k = 0; k < KSP; ++k)
      if (k0 + k < KS) {
        const int kk = k0 + k;
        if (layout == 0) { wout[(long)d * KS + kk] = acc0[k]; wout[(long)(d + 1) * KS + kk] = acc1[k]; }
        else { wout[(long)kk * Dw + d] = acc0[k]; wout[(long)kk * Dw + d + 1] = acc1[k]; }
      }
    if (blockIdx.z == 0) {
      wout[(long)KS * Dw + d] = ws0;
      wout[(long)KS * Dw + d + 1] = ws1;
    }
  }
  if (blockIdx.x == 0 && blockIdx.z == 0) {
    for (int k = lane; k < KS; k += 64) {
      float ssum = 0.f;
      for (int m = mbeg; m < mend; ++m) ssum += bf2f(small[(long)m * KS + k]);
      wout[(long)KS * Dw + Dw + k] = ssum;
    }
  }
}
__global__ __launch_bounds__(256) void smallk_reduce_kernel(const float* __restrict__ ws, long stride,
                                                            float* __restrict__ o0, long n0, float* __restrict__ o1,
                                                            long n1, float* __restrict__ o2, long n2, int accumulate) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n0 + n1 + n2) return;
  float* dst;
  if (i < n0) dst = o0 ? o0 + i : nullptr;
  else if (i < n0 + n1) dst = o1 ? o1 + (i - n0) : nullptr;
  else dst = o2 ? o2 + (i - n0 - n1) : nullptr;
  if (!dst) return;
  float s = accumulate ? *dst : 0.f;
  for (int z = 0; z < NSL; ++z) s += ws[(long)z * stride + i];
  *dst = s;
}

// ---- timestep sinusoid ----
__global__ void sinusoid_kernel(const float* __restrict__ t, bf16* __restrict__ out, int B, int dim, float max_period) {
  const int half = dim / 2;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * half) return;
  int b = i / half, k = i - b * half;
  float a = (float)(-log((double)max_period)) * (float)k;
  float f = expf(a / (float)half);
  float arg = t[b] * f;
  out[(long)b * dim + k] = f2bf(cosf(arg));
  out[(long)b * dim + half + k] = f2bf(sinf(arg));
  if ((dim & 1) && k == 0) out[(long)b * dim + dim - 1] = f2bf(0.f);
}

// ---- label embedding + conditioning ----
// A label outside [0, table_rows) (nn.Embedding raises there, sit.py:98) must not index past the table: the table lives
// in the flat parameter arena, so the read would return other parameters and the backward scatter would add into their
// gradients. Such a label is replaced by row 0 and reported through the sticky device flag `err` (read by the host at
// its next synchronisation point: reed_amd/engine.py:Engine.check_errors).
__global__ void label_cond_kernel(const int64_t* __restrict__ labels, const uint8_t* __restrict__ drop, int num_classes,
                                  int table_rows, const float* __restrict__ table, const bf16* __restrict__ t_emb,
                                  int64_t* __restrict__ labels_out, float* __restrict__ c, bf16* __restrict__ silu_c,
                                  int* __restrict__ err, int D) {
  const int b = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
  int64_t lab = labels[b];
  if (drop && drop[b]) lab = num_classes;
  if (lab < 0 || lab >= table_rows) {
    if (d == 0 && err) *err = 1;
    lab = 0;
  }
  if (d == 0 && labels_out) labels_out[b] = lab;
  if (d >= D) return;
  float v = bf2f(t_emb[(long)b * D + d]) + table[lab * D + d];
  c[(long)b * D + d] = v;
  silu_c[(long)b * D + d] = f2bf(silu_f(v));
}
__global__ void label_cond_bwd_kernel(const float* __restrict__ dsilu, const float* __restrict__ c,
                                      const int64_t* __restrict__ labels_eff, bf16* __restrict__ dt_emb,
                                      float* __restrict__ dtable, int B, int D) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  for (int b = 0; b < B; ++b) {  // sequential over the batch: deterministic scatter-add
    float g = bfround(dsilu[(long)b * D + d]) * silu_grad_f(c[(long)b * D + d]);
    dt_emb[(long)b * D + d] = f2bf(g);
    dtable[labels_eff[b] * D + d] += g;
  }
}

// ---- final layer forward: wave per row ----
__global__ __launch_bounds__(256) void final_fwd_kernel(const float* __restrict__ x, const bf16* __restrict__ shift,
                                                        const bf16* __restrict__ scale, long ldmod,
                                                        const bf16* __restrict__ w, const bf16* __restrict__ bias,
                                                        float* __restrict__ out, float* __restrict__ mean,
                                                        float* __restrict__ rstd, int B, int T, int D, int C, int P,
                                                        float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int M = B * T;
  if (row >= M) return;
  const int lane = threadIdx.x & 63, nv = D >> 2;
  const float* xr = x + (long)row * D;
  f32x4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int idx = lane + 64 * k;
    if (idx < nv) { v[k] = *(const f32x4*)(xr + idx * 4); s += v[k][0] + v[k][1] + v[k][2] + v[k][3]; }
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
  const int b = row / T, t = row - b * T;
  const bf16* sc = scale + (long)b * ldmod;
  const bf16* sh = shift + (long)b * ldmod;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int idx = lane + 64 * k;
    if (idx < nv) {
      f32x4 a = ld_bf4(sc + idx * 4), bb = ld_bf4(sh + idx * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[k][j] = bfround((v[k][j] - mu) * r * bfround(1.f + a[j]) + bb[j]);
    }
  }
  const int NO = P * P * C, HW = (int)(sqrtf((float)T) + 0.5f) * P;
  for (int j = 0; j < NO; ++j) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      int idx = lane + 64 * k;
      if (idx < nv) {
        f32x4 wv = ld_bf4(w + (long)j * D + idx * 4);
        acc += v[k][0] * wv[0] + v[k][1] * wv[1] + v[k][2] * wv[2] + v[k][3] * wv[3];
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) out[patch_src(b, t, j, C, HW, P, 1)] = bfround(acc + (bias ? bf2f(bias[j]) : 0.f));
  }
}

// The same with the weight staged in LDS once per block of FR rows (a wave walks FR / 4 of them): the wave-per-row kernel above
// re-reads the [NO, D] weight (36 KiB for XL/2) from L1 / L2 for EVERY row — 2.4 GB at b = 256, 0.67 ms for a pass whose HBM
// traffic is 0.3 GB.  Same per-lane accumulation order and the same wave reduction: bit-identical outputs.
constexpr int FR = 64;
__global__ __launch_bounds__(256) void final_fwd_lds_kernel(const float* __restrict__ x, const bf16* __restrict__ shift,
                                                            const bf16* __restrict__ scale, long ldmod,
                                                            const bf16* __restrict__ w, const bf16* __restrict__ bias,
                                                            float* __restrict__ out, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int B, int T, int D, int C, int P,
                                                            float eps) {
  extern __shared__ __attribute__((aligned(16))) char wl_raw[];
  bf16* wl = (bf16*)wl_raw;                       // [NO][D]
  const int NO = P * P * C, M = B * T;
  for (int i = threadIdx.x * 8; i < NO * D; i += 256 * 8) *(bf16x8*)(wl + i) = *(const bf16x8*)(w + i);   // NO * D % 8 == 0
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nv = D >> 2;
  const int HW = (int)(sqrtf((float)T) + 0.5f) * P;
  // two rows per wave and pass, every stage issued for both before either is consumed (a wave's row is a chain of dependent
  // wave reductions; the second row fills its gaps); the arithmetic of a row is unchanged
  for (int rr = 2 * wave; rr < FR; rr += 8) {
    const int row0 = blockIdx.x * FR + rr;
    if (row0 >= M) break;
    const bool two = row0 + 1 < M;
    f32x4 v[2][MAXV];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float* xr = x + (long)(row0 + (u && two ? 1 : 0)) * D;
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        int idx = lane + 64 * k;
        if (idx < nv) { v[u][k] = *(const f32x4*)(xr + idx * 4); s[u] += v[u][k][0] + v[u][k][1] + v[u][k][2] + v[u][k][3]; }
      }
    }
    float mu[2], r[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) mu[u] = wave_sum(s[u]) / D;
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        int idx = lane + 64 * k;
        if (idx < nv) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { float d = v[u][k][j] - mu[u]; q[u] += d * d; }
        }
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) r[u] = rsqrtf(wave_sum(q[u]) / D + eps);
    int bb_[2], tt_[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = row0 + (u && two ? 1 : 0);
      if (lane == 0 && mean && (u == 0 || two)) { mean[row] = mu[u]; rstd[row] = r[u]; }
      bb_[u] = row / T;
      tt_[u] = row - bb_[u] * T;
      const bf16* sc = scale + (long)bb_[u] * ldmod;
      const bf16* sh = shift + (long)bb_[u] * ldmod;
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        int idx = lane + 64 * k;
        if (idx < nv) {
          f32x4 a = ld_bf4(sc + idx * 4), bb = ld_bf4(sh + idx * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[u][k][j] = bfround((v[u][k][j] - mu[u]) * r[u] * bfround(1.f + a[j]) + bb[j]);
        }
      }
    }
    for (int j = 0; j < NO; ++j) {
      float acc[2] = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        int idx = lane + 64 * k;
        if (idx < nv) {
          f32x4 wv = ld_bf4(wl + j * D + idx * 4);
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[u] += v[u][k][0] * wv[0] + v[u][k][1] * wv[1] + v[u][k][2] * wv[2] + v[u][k][3] * wv[3];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = wave_sum(acc[u]);
      if (lane == 0) {
        const float bj = bias ? bf2f(bias[j]) : 0.f;
        out[patch_src(bb_[0], tt_[0], j, C, HW, P, 1)] = bfround(acc[0] + bj);
        if (two) out[patch_src(bb_[1], tt_[1], j, C, HW, P, 1)] = bfround(acc[1] + bj);
      }
    }
  }
}

// ---- final layer backward, row part: h (bf16), dlin (bf16), dh (bf16) ----
__global__ __launch_bounds__(256) void final_bwd_rows_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const bf16* __restrict__ shift,
                                                             const bf16* __restrict__ scale, long ldmod,
                                                             const bf16* __restrict__ w, bf16* __restrict__ hbuf,
                                                             bf16* __restrict__ dlin, bf16* __restrict__ dh, int B,
                                                             int T, int D, int C, int P) {
  extern __shared__ float dl[];  // [4 waves][NO]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nv = D >> 2;
  const int row = blockIdx.x * 4 + wave;
  const int M = B * T, NO = P * P * C, HW = (int)(sqrtf((float)T) + 0.5f) * P;
  const bool ok = row < M;
  const int b = ok ? row / T : 0, t = ok ? row - b * T : 0;
  float* my = dl + wave * NO;
  if (ok) {
    for (int j = lane; j < NO; j += 64) {
      float g = bfround(dout[patch_src(b, t, j, C, HW, P, 1)]);
      my[j] = g;
      dlin[(long)row * NO + j] = f2bf(g);
    }
  }
  __syncthreads();
  if (!ok) return;
  const float mu = mean[row], r = rstd[row];
  const bf16* sc = scale + (long)b * ldmod;
  const bf16* sh = shift + (long)b * ldmod;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int idx = lane + 64 * k;
    if (idx < nv) {
      f32x4 xv = *(const f32x4*)(x + (long)row * D + idx * 4);
      f32x4 a = ld_bf4(sc + idx * 4), bb = ld_bf4(sh + idx * 4);
      bf16x4 hv, gv;
      f32x4 acc = {0, 0, 0, 0};
      for (int j = 0; j < NO; ++j) {
        f32x4 wv = ld_bf4(w + (long)j * D + idx * 4);
        float g = my[j];
        acc[0] += g * wv[0]; acc[1] += g * wv[1]; acc[2] += g * wv[2]; acc[3] += g * wv[3];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hv[j] = f2bf((xv[j] - mu) * r * bfround(1.f + a[j]) + bb[j]);
        gv[j] = f2bf(acc[j]);
      }
      *(bf16x4*)(hbuf + (long)row * D + idx * 4) = hv;
      *(bf16x4*)(dh + (long)row * D + idx * 4) = gv;
    }
  }
}

// The same with the weight in LDS per block of FR rows (see final_fwd_lds_kernel): bit-identical outputs.
__global__ __launch_bounds__(256) void final_bwd_rows_lds_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const bf16* __restrict__ shift, const bf16* __restrict__ scale,
                                                                 long ldmod, const bf16* __restrict__ w, bf16* __restrict__ hbuf,
                                                                 bf16* __restrict__ dlin, bf16* __restrict__ dh, int B, int T,
                                                                 int D, int C, int P) {
  extern __shared__ __attribute__((aligned(16))) char lraw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nv = D >> 2;
  const int M = B * T, NO = P * P * C, HW = (int)(sqrtf((float)T) + 0.5f) * P;
  bf16* wl = (bf16*)lraw;                                   // [NO][D]
  float* my = (float*)(lraw + (long)NO * D * sizeof(bf16)) + wave * NO;   // this wave's NO output gradients of its current row
  for (int i = threadIdx.x * 8; i < NO * D; i += 256 * 8) *(bf16x8*)(wl + i) = *(const bf16x8*)(w + i);
  __syncthreads();
  for (int rr = wave; rr < FR; rr += 4) {
    const int row = blockIdx.x * FR + rr;
    if (row >= M) break;
    const int b = row / T, t = row - b * T;
    for (int j = lane; j < NO; j += 64) {
      float g = bfround(dout[patch_src(b, t, j, C, HW, P, 1)]);
      my[j] = g;
      dlin[(long)row * NO + j] = f2bf(g);
    }
    __builtin_amdgcn_wave_barrier();           // `my` is this wave's own: LDS executes a wave's accesses in order
    const float mu = mean[row], r = rstd[row];
    const bf16* sc = scale + (long)b * ldmod;
    const bf16* sh = shift + (long)b * ldmod;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      int idx = lane + 64 * k;
      if (idx < nv) {
        f32x4 xv = *(const f32x4*)(x + (long)row * D + idx * 4);
        f32x4 a = ld_bf4(sc + idx * 4), bb = ld_bf4(sh + idx * 4);
        bf16x4 hv, gv;
        f32x4 acc = {0, 0, 0, 0};
        for (int j = 0; j < NO; ++j) {
          f32x4 wv = ld_bf4(wl + j * D + idx * 4);
          float g = my[j];
          acc[0] += g * wv[0]; acc[1] += g * wv[1]; acc[2] += g * wv[2]; acc[3] += g * wv[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hv[j] = f2bf((xv[j] - mu) * r * bfround(1.f + a[j]) + bb[j]);
          gv[j] = f2bf(acc[j]);
        }
        *(bf16x4*)(hbuf + (long)row * D + idx * 4) = hv;
        *(bf16x4*)(dh + (long)row * D + idx * 4) = gv;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

extern "C" int reed_patchify_bf16(const float* x, void* out, int B, int C, int HW, int P, int order, void* stream) {
  REED_CHECK_ARG(HW % P == 0, "patchify: HW=%d not divisible by P=%d", HW, P);
  long n = (long)B * C * HW * HW;
  REED_KLAUNCH(patchify_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16*)out, B, C, HW, P, order);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_patch_embed_fwd(const float* x, const void* w, const void* bias, const float* pos,
                                    float* tokens, int B, int C, int HW, int P, int D, void* stream) {
  REED_CHECK_ARG(x && w && pos && tokens, "patch_embed_fwd: null pointer");
  REED_CHECK_ARG(HW % P == 0 && (C * P * P) % 8 == 0, "patch_embed: HW=%d P=%d C=%d unsupported", HW, P, C);
  const int T = (HW / P) * (HW / P), K = C * P * P;
  if (K == PK && D % 4 == 0 && D <= 4 * PNT && ((uintptr_t)w % 16) == 0 && ((uintptr_t)pos % 16) == 0 &&
      ((uintptr_t)tokens % 16) == 0) {   // patch 2 on 4 channels (every SiT-*/2 preset)
    REED_KLAUNCH(patch_embed_fwd16_kernel, dim3(cdiv((long)B * T, PT)), dim3(PNT), 0, (hipStream_t)stream, x, (const bf16*)w,
                 (const bf16*)bias, pos, tokens, B, C, HW, P, D);
    REED_LAUNCH_CHECK();
    return REED_OK;
  }
  REED_KLAUNCH(patch_embed_fwd_kernel, dim3(cdiv((long)B * T, 8)), dim3(256), 8 * K * sizeof(float),
                     (hipStream_t)stream, x, (const bf16*)w, (const bf16*)bias, pos, tokens, B, C, HW, P, D);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int64_t reed_smallk_wgrad_ws_floats(int Dw, int KS) { return (int64_t)NSL * ((int64_t)KS * Dw + Dw + KS); }

extern "C" int reed_smallk_wgrad(const void* wide, int wide_is_f32, const void* small, float* ws, float* out,
                                 float* colsum_wide, float* colsum_small, int M, int Dw, int KS, int layout,
                                 int accumulate, void* stream) {
  REED_CHECK_ARG(wide && small && ws, "smallk_wgrad: null pointer");
  REED_CHECK_ARG(KS >= 8 && KS <= 512 && KS % 8 == 0, "smallk_wgrad: KS=%d must be a multiple of 8 in 8..512", KS);
  REED_CHECK_ARG(Dw > 0 && Dw % 2 == 0, "smallk_wgrad: Dw=%d must be even", Dw);
  if (KS <= 16) {
    dim3 grid(cdiv(Dw, 128), NSL, 1);
    if (wide_is_f32) REED_KLAUNCH((smallk_wgrad_kernel<true, 16>), grid, dim3(64), 0, (hipStream_t)stream, wide, (const bf16*)small, ws, M, Dw, KS, layout);
    else REED_KLAUNCH((smallk_wgrad_kernel<false, 16>), grid, dim3(64), 0, (hipStream_t)stream, wide, (const bf16*)small, ws, M, Dw, KS, layout);
  } else {
    dim3 grid(cdiv(Dw, 128), NSL, cdiv(KS, 32));
    if (wide_is_f32) REED_KLAUNCH((smallk_wgrad_kernel<true, 32>), grid, dim3(64), 0, (hipStream_t)stream, wide, (const bf16*)small, ws, M, Dw, KS, layout);
    else REED_KLAUNCH((smallk_wgrad_kernel<false, 32>), grid, dim3(64), 0, (hipStream_t)stream, wide, (const bf16*)small, ws, M, Dw, KS, layout);
  }
  REED_LAUNCH_CHECK();
  long n0 = (long)KS * Dw, n1 = Dw, n2 = KS;
  REED_KLAUNCH(smallk_reduce_kernel, dim3(cdiv(n0 + n1 + n2, 256)), dim3(256), 0, (hipStream_t)stream, ws,
                     n0 + n1 + n2, out, n0, colsum_wide, n1, colsum_small, n2, accumulate);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_timestep_sinusoid(const float* t, void* out, int B, int dim, float max_period, void* stream) {
  REED_CHECK_ARG(t && out && dim >= 2, "timestep_sinusoid: bad args");
  REED_KLAUNCH(sinusoid_kernel, dim3(cdiv((long)B * (dim / 2), 256)), dim3(256), 0, (hipStream_t)stream, t,
                     (bf16*)out, B, dim, max_period);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_label_cond(const int64_t* labels, const uint8_t* drop, int num_classes, int table_rows,
                               const float* table, const void* t_emb, int64_t* labels_out, float* c, void* silu_c,
                               int* err_flag, int B, int D, void* stream) {
  REED_CHECK_ARG(labels && table && t_emb && c && silu_c, "label_cond: null pointer");
  REED_CHECK_ARG(table_rows >= 1 && (!drop || num_classes < table_rows),
                 "label_cond: label dropout needs a null-class row (num_classes < table_rows)");
  REED_KLAUNCH(label_cond_kernel, dim3(cdiv(D, 256), B), dim3(256), 0, (hipStream_t)stream, labels, drop,
                     num_classes, table_rows, table, (const bf16*)t_emb, labels_out, c, (bf16*)silu_c, err_flag, D);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_label_cond_bwd(const float* dsilu_c, const float* c, const int64_t* labels_eff, void* dt_emb,
                                   float* dtable, int B, int D, void* stream) {
  REED_CHECK_ARG(dsilu_c && c && labels_eff && dt_emb && dtable, "label_cond_bwd: null pointer");
  REED_KLAUNCH(label_cond_bwd_kernel, dim3(cdiv(D, 256)), dim3(256), 0, (hipStream_t)stream, dsilu_c, c,
                     labels_eff, (bf16*)dt_emb, dtable, B, D);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_final_layer_fwd(const float* x, const void* shift, const void* scale, int64_t ldmod,
                                    const void* w, const void* bias, float* out, float* mean, float* rstd,
                                    int B, int T, int D, int C, int P, float eps, void* stream) {
  REED_CHECK_ARG(x && shift && scale && w && out, "final_layer_fwd: null pointer");
  REED_CHECK_ARG(D % 4 == 0 && D <= 256 * MAXV, "final_layer: D=%d unsupported", D);
  int G = (int)(sqrtf((float)T) + 0.5f);
  REED_CHECK_ARG(G * G == T, "final_layer: T=%d is not a square grid", T);
  const long wbytes = (long)P * P * C * D * sizeof(bf16);
  if (wbytes <= 64 * 1024 && (P * P * C * D) % 8 == 0 && ((uintptr_t)w % 16) == 0) {   // the weight fits a block's LDS
    REED_KLAUNCH(final_fwd_lds_kernel, dim3(cdiv((long)B * T, FR)), dim3(256), (size_t)wbytes, (hipStream_t)stream, x,
                 (const bf16*)shift, (const bf16*)scale, (long)ldmod, (const bf16*)w, (const bf16*)bias, out, mean, rstd, B, T, D, C, P,
                 eps);
    REED_LAUNCH_CHECK();
    return REED_OK;
  }
  REED_KLAUNCH(final_fwd_kernel, dim3(cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream, x,
                     (const bf16*)shift, (const bf16*)scale, (long)ldmod, (const bf16*)w, (const bf16*)bias, out, mean,
                     rstd, B, T, D, C, P, eps);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_final_layer_bwd_rows(const float* dout, const float* x, const float* mean, const float* rstd,
                                         const void* shift, const void* scale, int64_t ldmod, const void* w,
                                         void* hbuf, void* dlin, void* dh, int B, int T, int D, int C, int P,
                                         void* stream) {
  REED_CHECK_ARG(dout && x && mean && rstd && shift && scale && w && hbuf && dlin && dh, "final_layer_bwd_rows: null pointer");
  REED_CHECK_ARG(D % 4 == 0 && D <= 256 * MAXV, "final_layer: D=%d unsupported", D);
  const int NO = P * P * C;
  const long wb = (long)NO * D * sizeof(bf16);
  if (wb + 4 * NO * (long)sizeof(float) <= 64 * 1024 && (NO * D) % 8 == 0 && ((uintptr_t)w % 16) == 0) {
    REED_KLAUNCH(final_bwd_rows_lds_kernel, dim3(cdiv((long)B * T, FR)), dim3(256), (size_t)(wb + 4 * NO * sizeof(float)),
                 (hipStream_t)stream, dout, x, mean, rstd, (const bf16*)shift, (const bf16*)scale, (long)ldmod, (const bf16*)w,
                 (bf16*)hbuf, (bf16*)dlin, (bf16*)dh, B, T, D, C, P);
    REED_LAUNCH_CHECK();
    return REED_OK;
  }
  REED_KLAUNCH(final_bwd_rows_kernel, dim3(cdiv((long)B * T, 4)), dim3(256), 4 * NO * sizeof(float),
                     (hipStream_t)stream, dout, x, mean, rstd, (const bf16*)shift, (const bf16*)scale, (long)ldmod,
                     (const bf16*)w, (bf16*)hbuf, (bf16*)dlin, (bf16*)dh, B, T, D, C, P);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
