// Multi-head self-attention for the SiT block (timm Attention called at image/models/sit.py:114-118,134):
//   o = softmax(q k^T / sqrt(hd)) v      q,k,v = qkv.reshape(B,T,3,H,hd).permute(2,0,3,1,4)
// head_dim 64 (S/B/L) or 72 (XL: 1152/16), T = 256 tokens at 256x256 / patch 2.
//
// MI355X design: the whole K and V of one (batch, head) fit in LDS (256 x 72 bf16 = 36 KiB each,
// rows padded to 72 elements = 144 B for both head sizes, which also spreads the banks), so a
// workgroup of 8 waves loads them once and every wave owns 32 query rows.  Scores are computed
// "transposed" (S^T = K Q^T) so that one query's scores live in ONE lane's registers + its 3
// partner lanes (lane>>4): row max / sum are in-register plus two cross-lane steps, and the
// accumulator tile is already the B operand of the next MFMA (O^T = V^T P^T); the other operand
// (V^T, K^T, Q^T, dO^T) comes straight from row-major LDS tiles through ds_read_b64_tr_b16.
// No S x S matrix, no transposed copies, qkv is consumed in the layout the qkv GEMM wrote.
// head_dim 72 is handled on chip: the third k-step (cols 64..95) uses zeroed register fragments
// for the padded slots; the output's fifth 16-column tile is computed and only cols 64..71 stored.
//
// Backward recomputes P from the saved log-sum-exp, key-stationary (a wave owns 32 keys: dK / dV in registers, S and dP
// once) with dQ^T = K^T dS^T formed every 64 queries through an LDS tile of dS^T; deterministic, no atomics.  Kernels in
// this file: attn_fwd_kernel (any T) + attn_fwd_rows_kernel (a <= 16-row tail), attn_fwd256p_kernel (T <= 256, persistent:
// the engine's forward), attn_bwd_ks_kernel (one shot), attn_bwd_ksp_kernel (persistent, T < 256), attn_bwd_ring_kernel
// (persistent with Q / dO rings, T = 256: the engine's backward).  Forms that were measured and removed are named where
// they stood (round 2's three-barrier forward, round 1's two-phase backward, the half-workgroup stagger).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "../../include/reed_hip.h"
#include "common.hpp"

// cache policy (aux bits: 2 = nt) of the LDS-DMA tile loads: Q, K, V, dO are read exactly once per launch.  Build parameter for A/B.
#ifndef REED_ATTN_LD_AUX
#define REED_ATTN_LD_AUX 0
#endif
namespace {

constexpr int ROWB = 144;            // LDS row stride in bytes (72 bf16): the backward's four resident tiles
constexpr int TILE_B = 256 * ROWB;   // one 256-row tile
// The forward's two tiles use 160-byte rows (80 bf16).  With 144-byte rows the 16-lane groups of ds_read_b128
// ({0-3,12-15,20-27}, ... — MI355X_MICROARCH.md, LDS) put rows i and i' = i+8 (g = 0 / 1) on the same four banks and a
// 32-lane half of ds_read_b64_tr_b16 wraps row 7 onto row 0's banks: SQ_LDS_BANK_CONFLICT = 41 % of SQ_LDS_IDX_ACTIVE
// in the forward, 45 % in the backward (profiles/r1_pmc_gemm.txt).  At 40 dwords per row lane (i, g) of a b128 group
// lands on bank slot (10 i + g) mod 16 — even slots for one g, odd for the other — and the 8 rows of a transposed
// read on 8 disjoint 8-bank ranges (40 r mod 64 = 0, 40, 16, 56, 32, 8, 48, 24): conflict-free both ways.  Columns
// 72..79 of a row are zero; two tiles are exactly 80 KiB, i.e. two workgroups per CU still fit.
constexpr int ROWF = 160;
constexpr int TILE_F = 256 * ROWF;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <int HD>
struct Cfg {
  static constexpr int KS = (HD + 31) / 32;  // k-steps of 32 over head_dim
  static constexpr int DT = (HD + 15) / 16;  // 16-wide output tiles over head_dim
  static constexpr int NCH = HD / 8;         // 16-byte chunks per row
};

template <int HD, int RB = ROWB>
__device__ __forceinline__ void load_tile(char* lds, const bf16* src, long row_stride, int rows_valid,
                                          int tid, int nthreads) {
  // 256 rows x NCH 16-byte chunks. All loads are issued unconditionally on a clamped row (a per-load bounds branch
  // would serialise them: cdna_hip_programming.md trap (c)); rows >= rows_valid are zeroed by a select afterwards.
  constexpr int NCH = Cfg<HD>::NCH;
  constexpr int PER = (256 * NCH + 511) / 512;  // chunks per thread at 512 threads
  uint4 v[PER];
  const int last = rows_valid - 1;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int idx = tid + k * 512;
    int row = idx / NCH, c = idx - row * NCH;
    int rc = min(row, last);
    if (rc < 0) rc = 0;
    int cc = idx < 256 * NCH ? c : 0;
    v[k] = *(const uint4*)(src + (long)rc * row_stride + cc * 8);
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int idx = tid + k * 512;
    int row = idx / NCH, c = idx - row * NCH;
    if (idx < 256 * NCH) {
      const unsigned msk = row < rows_valid ? 0xFFFFFFFFu : 0u;  // component-wise mask (a select of two uint4
      uint4 w = v[k];                                            // aggregates is lowered through scratch memory)
      w.x &= msk; w.y &= msk; w.z &= msk; w.w &= msk;
      *(uint4*)(lds + row * RB + c * 16) = w;
    }
  }
  if (RB > HD * 2 + 8 && tid < 256) {   // padded row format: the chunk behind the head's columns is read (x 0) — keep it zero
#pragma unroll
    for (int c = Cfg<HD>::NCH; c < RB / 16; ++c) *(uint4*)(lds + tid * RB + c * 16) = make_uint4(0, 0, 0, 0);
  }
}

// lane (i = lane&15, g = lane>>4): X[row0 + i][ks*32 + 8g .. +7]
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int row0, int ks, int lane) {
  return *(const bf16x8*)(tile + (row0 + (lane & 15)) * ROWB + (ks * 32 + 8 * (lane >> 4)) * 2);
}
// the same on a 160-byte-row tile; WRAP: in the k-step that covers columns 64..95 of a 72-wide head the lanes g >= 2
// (columns 80..95, past the row) re-read columns 64..79 instead — their MFMA partner slots are zero either way
template <bool WRAP>
__device__ __forceinline__ bf16x8 frag_rows_f(const char* tile, int row0, int ks, int lane) {
  const int g = WRAP ? ((lane >> 4) & 1) : (lane >> 4);
  return *(const bf16x8*)(tile + (row0 + (lane & 15)) * ROWF + (ks * 32 + 8 * g) * 2);
}
__device__ __forceinline__ bf16x8 frag_trT_f(const char* tile, int rbase, int d0, int lane) {
  int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
  const char* a0 = tile + (rbase + 4 * g + q) * ROWF + (d0 + 4 * p) * 2;
  bf16x4 lo = REED_DS_READ_TR16_B64((bf16x4 __attribute__((address_space(3)))*)a0);
  bf16x4 hi = REED_DS_READ_TR16_B64((bf16x4 __attribute__((address_space(3)))*)(a0 + 16 * ROWF));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// lane (i, g): X[rbase + 16*(j>>2) + 4g + (j&3)][d0 + i], j = 0..7 — the k-slot order in which an
// MFMA accumulator tile pair (rows 4g+r of two stacked 16-row tiles) serves as the other operand.
__device__ __forceinline__ bf16x8 frag_trT(const char* tile, int rbase, int d0, int lane) {
  int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
  const char* a0 = tile + (rbase + 4 * g + q) * ROWB + (d0 + 4 * p) * 2;
  bf16x4 lo = REED_DS_READ_TR16_B64((bf16x4 __attribute__((address_space(3)))*)a0);
  bf16x4 hi = REED_DS_READ_TR16_B64((bf16x4 __attribute__((address_space(3)))*)(a0 + 16 * ROWB));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int HD>
__device__ __forceinline__ bf16x8 load_frag_global(const bf16* rowptr, bool row_ok, int ks, int lane) {
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
  int col = ks * 32 + 8 * (lane >> 4);
  if (row_ok && col < HD) z = *(const bf16x8*)(rowptr + col);
  return z;
}
template <int HD, int RB = ROWB>
__device__ __forceinline__ bf16x8 frag_rows_z(const char* tile, int row0, int ks, int lane) {
  // register-resident operand: padded k-slots (col >= HD) must be exact zeros
  bf16x8 f = *(const bf16x8*)(tile + (row0 + (lane & 15)) * RB + (ks * 32 + 8 * (lane >> 4)) * 2);
  if (HD % 32 != 0 && ks * 32 + 8 * (lane >> 4) >= HD) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16)0.f;
  }
  return f;
}
__device__ __forceinline__ bf16x8 pack2(f32x4 a, f32x4 b) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = f2bf(a[j]); r[4 + j] = f2bf(b[j]); }
  return r;
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
#define MFMA(a, b, c) REED_MFMA_16x16x32((a), (b), (c))

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  One (batch, head) reads 144-byte (hd 72) or
// 128-byte pieces of every token row of qkv, so the cache lines it touches are shared with the neighbouring heads of
// the same token: give each XCD a contiguous run of (batch, head) pairs, so that the 16 heads of a sample run on one
// XCD at about the same time and the shared lines (and the partial-line dqkv writes) meet in that L2.
__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
  const int xcd = bid & 7, q = n >> 3, r = n & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ------------------------------------------------------------------------------------------
// Forward for any T (512^2 sampling: T = 1024; the CLIP tower: T = 257): 256-key tiles with online softmax, one
// workgroup per (batch, head, 256-query block), register-staged tile loads.  T <= 256 runs attn_fwd256p_kernel below.
template <int HD>
__global__ __launch_bounds__(512, 4) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                       float* __restrict__ lse, int B, int T, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = Cfg<HD>::KS, DT = Cfg<HD>::DT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int D = H * HD;
  const int bh = xcd_contiguous(blockIdx.x, gridDim.x);
  const int b = bh / H, h = bh % H;
  const long tok = 3l * D;
  const bf16* base = qkv + (long)b * T * tok + h * HD;
  char* Kt = smem;
  char* Vt = smem + TILE_F;
  const int q0 = blockIdx.y * 256 + wave * 32;
  const bool active = q0 < T;
  const float sc2 = rsqrtf((float)HD) * LOG2E;

  bf16x8 qf[2][KS];
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  f32x4 ot[2][DT];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) ot[qt][dt] = zero4();

  for (int kv0 = 0; kv0 < T; kv0 += 256) {
    __syncthreads();
    const int rows = min(256, T - kv0);
    load_tile<HD, ROWF>(Kt, base + (long)kv0 * tok + D, tok, rows, tid, 512);
    load_tile<HD, ROWF>(Vt, base + (long)kv0 * tok + 2 * D, tok, rows, tid, 512);
    if (kv0 == 0) {  // Q fragments after the first tile loads: keeps the staging registers and Q from overlapping
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        int row = q0 + 16 * qt + i;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qt][ks] = load_frag_global<HD>(base + (long)row * tok, row < T, ks, lane);
      }
    }
    __syncthreads();
    if (!active) continue;
    const int nsub = (rows + 63) >> 6;
    for (int sub = 0; sub < nsub; ++sub) {
      const int kvs = sub * 64;
      f32x4 st[2][4];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) st[qt][kt] = zero4();
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 kf = (HD % 32 != 0 && ks == KS - 1) ? frag_rows_f<true>(Kt, kvs + 16 * kt, ks, lane)
                                                     : frag_rows_f<false>(Kt, kvs + 16 * kt, ks, lane);
          st[0][kt] = MFMA(kf, qf[0][ks], st[0][kt]);
          st[1][kt] = MFMA(kf, qf[1][ks], st[1][kt]);
        }
      bf16x8 pb[2][2];
      // keys past T exist only in a ragged last sub-block: mask them there (wave-uniform branch), nowhere else
      if (kv0 + kvs + 64 > T) {
        asm volatile("; ragged key sub-block" ::);   // keeps this a real (wave-uniform) branch: no if-conversion
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kv0 + kvs + 16 * kt + 4 * g + r >= T) st[qt][kt][r] = -INFINITY;
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        // running maximum on the raw scores (the scale is positive); the scale rides in the exponent's FMA
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[qt][kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(m[qt], mx * sc2);
        const float alpha = __builtin_amdgcn_exp2f(m[qt] - mnew);
        m[qt] = mnew;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qt][kt][r], sc2, -mnew));
            st[qt][kt][r] = p;
            sum += p;
          }
        l[qt] = l[qt] * alpha + sum;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ot[qt][dt] *= alpha;
        pb[qt][0] = pack2(st[qt][0], st[qt][1]);
        pb[qt][1] = pack2(st[qt][2], st[qt][3]);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          bf16x8 vf = frag_trT_f(Vt, kvs + 32 * s, 16 * dt, lane);
          ot[0][dt] = MFMA(vf, pb[0][s], ot[0][dt]);
          ot[1][dt] = MFMA(vf, pb[1][s], ot[1][dt]);
        }
    }
  }
  // O leaves through LDS (the wave's own 32 rows of the K tile, free once every wave is past its last key tile): in the
  // MFMA layout a wave-level store is sixteen 32-byte pieces; staged, it is whole 144-byte row pieces at 16 B per lane.
  __syncthreads();
  if (!active) return;
  char* stg = Kt + (wave * 32) * ROWF;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float lt = l[qt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = 1.f / lt;
    const int q = q0 + 16 * qt + i;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      int d = 16 * dt + 4 * g;
      if (d < HD) {
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = f2bf(ot[qt][dt][r] * inv);
        *(bf16x4*)(stg + (16 * qt + i) * ROWF + d * 2) = v;
      }
    }
    if (q < T && g == 0 && lse) lse[((long)b * H + h) * T + q] = m[qt] * LN2 + __logf(lt);
  }
  {
    constexpr int NCH = Cfg<HD>::NCH, NQ = 32 * NCH;
    bf16* obase = o + ((long)b * T + q0) * D + h * HD;
#pragma unroll
    for (int k = 0; k < (NQ + 63) / 64; ++k) {
      const int qi = lane + 64 * k;
      const int rr = qi / NCH, c = qi - rr * NCH;
      if (qi < NQ && q0 + rr < T) *(uint4*)(obase + (long)rr * D + c * 8) = *(const uint4*)(stg + rr * ROWF + c * 16);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Tail query rows of a sequence that is a few tokens longer than a multiple of 256 (round 4): T = 257 (a ViT tower's 256
// patches + CLS; 261 with DINOv2's registers).  attn_fwd_kernel gives every 256-query block its own workgroup, and the block that
// holds ONE query still loads every key and value tile of the head: 2 x the memory traffic and 2 x the workgroups of T = 256
// (8.6 % of the matrix peak in the CLIP tower).  Those rows (<= 16 per head) are done here on the vector ALUs instead — one
// wave per (batch, head, row), head_dim 64: scores with the key on the lane (q broadcast through scalar registers), exact
// softmax across the wave, then O with head_dim on the lane (p broadcast by v_readlane) — and the main kernel runs the full
// 256-query blocks only.  ~2 k instructions per wave; fp32 probabilities (the main kernel rounds them to 16 bits for the MFMA).
__global__ __launch_bounds__(256) void attn_fwd_rows_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                            float* __restrict__ lse, int T, int H, int row0, int nrows,
                                                            int nwaves) {
  constexpr int HD = 64, KPL = 8;                   // keys per lane: T <= 512
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= nwaves) return;
  const int lane = threadIdx.x & 63;
  const int bh = w / nrows, row = row0 + (w - bh * nrows);
  const int b = bh / H, h = bh - b * H;
  const long D = (long)H * HD, tok = 3 * D;
  const bf16* base = qkv + (long)b * T * tok + h * HD;
  const float scale = rsqrtf((float)HD);
  // q[d] * scale in 64 scalar registers
  const float qv = bf2f(base[(long)row * tok + lane]) * scale;
  float q[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) q[d] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qv), d));
  // scores of keys lane, lane + 64, ...
  float sc[KPL];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    const int key = lane + 64 * j;
    sc[j] = -INFINITY;
    if (64 * j < T) {                               // wave-uniform
      const bf16* kr = base + D + (long)min(key, T - 1) * tok;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) {
        const bf16x8 kk = *(const bf16x8*)(kr + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(bf2f(kk[e]), q[8 * c + e], acc);
      }
      if (key < T) sc[j] = acc;
      mx = fmaxf(mx, sc[j]);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    sc[j] = __builtin_amdgcn_exp2f((sc[j] - mx) * LOG2E);   // exp2(-inf) = 0 for the keys past T
    sum += sc[j];
  }
  sum = wave_sum(sum);
  // O[d = lane] = sum_key p[key] v[key][d]
  const bf16* vb = base + 2 * D + lane;
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    if (64 * j < T) {
      const int nk = min(64, T - 64 * j);
      for (int k0 = 0; k0 < nk; k0 += 8) {
        float vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = bf2f(vb[(long)min(64 * j + k0 + e, T - 1) * tok]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float p = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc[j]), (k0 + e) & 63));
          acc = fmaf((k0 + e < nk) ? p : 0.f, vv[e], acc);
        }
      }
    }
  }
  o[((long)b * T + row) * D + h * HD + lane] = f2bf(acc / sum);
  if (lse && lane == 0) lse[((long)b * H + h) * T + row] = mx + __logf(sum);
}

// ------------------------------------------------------------------------------------------
// LDS-DMA tile staging (buffer_load_dwordx4 ... lds): a [256][RB / 16]-chunk tile image is 256 * RB / 1024 wave
// instructions of 1 KiB; lane L of instruction I carries chunk c = 64 I + L = (row c / CPR, chunk c % CPR) from its own
// source address to the lane-linear LDS address tile + 16 c.  Chunks behind the head's columns (the pad of a padded row
// format) and rows >= rows_valid carry an offset outside the descriptor: the hardware range check writes zeros for them,
// so the tiles need no zero-fill code and ragged T needs no masks on the load side.
typedef void __attribute__((address_space(3))) * lds_ptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int DMA_OOB = 0x7FFFFFF0;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, long bytes) {
  if (!base || bytes < 0) bytes = 0;
  if (bytes > 0x7FFF0000l) bytes = 0x7FFF0000l;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (unsigned)bytes, 0x00020000);
}
template <int HD, int RB>
__device__ __forceinline__ int dma_voff(int c, int stride_bytes) {
  constexpr int CPR = RB / 16;
  const int row = c / CPR, col = c - row * CPR;
  return col < HD / 8 ? row * stride_bytes + col * 16 : DMA_OOB;
}
// bytes of a tile's source window: rows_valid rows of HD elements at stride_bytes
template <int HD>
__device__ __forceinline__ long tile_window(int rows_valid, int stride_bytes) {
  return rows_valid > 0 ? (long)(rows_valid - 1) * stride_bytes + HD * 2 : 0;
}
// raw barrier: no compiler-inserted vmcnt(0) (the LDS-DMA of the next tiles stays in flight across it)
#define ATTN_BARRIER()                                     \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("" ::: "memory");                         \
  } while (0)
#define ATTN_LDS_WAIT()                                    \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)
// ds_read_b64_tr_b16 through inline asm: with the builtin the compiler drains every pending LDS-DMA (s_waitcnt vmcnt(0))
// before the read (GEMM); the caller orders it (ATTN_LDS_WAIT before the first consumer).  ONE per-lane base
// address (tile + (4 g + q) RB + 8 p) and compile-time offsets in the DS immediate field: in a fully unrolled loop
// per-read address registers are loop invariants that get hoisted and spilled
template <int OFF>
__device__ __forceinline__ bf16x4 tr16_asm_off(const char* p) {
  bf16x4 r;
  const unsigned a = (unsigned)(size_t)(const char __attribute__((address_space(3)))*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(OFF));
  return r;
}
template <int RB, int OFF>
__device__ __forceinline__ bf16x8 frag_trT_off(const char* lane_base) {
  bf16x4 lo = tr16_asm_off<OFF>(lane_base);
  bf16x4 hi = tr16_asm_off<OFF + 16 * RB>(lane_base);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// lane (i, g): X[rbase + 16 (j >> 2) + 4 g + (j & 3)][d0 + i], j = 0..7 (frag_trT through the asm read, any row stride)
template <int RB>
__device__ __forceinline__ bf16x8 frag_trT_a(const char* tile, int rbase, int d0, int lane) {
  const int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
  const char* a0 = tile + (rbase + 4 * g + q) * RB + (d0 + 4 * p) * 2;
  bf16x4 lo = tr16_asm_off<0>(a0);
  bf16x4 hi = tr16_asm_off<16 * RB>(a0);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// O^T += V^T P^T over 32-key steps, software pipelined: the V^T fragments of step S + 1 are read (asm, unordered by the
// compiler) before the MFMAs of step S issue, and waited for behind them
template <int RB, int S, int NDT, int... DTS>
__device__ __forceinline__ void pv_load(const char* vb, bf16x8 (&vf)[NDT], std::integer_sequence<int, DTS...>) {
  ((vf[DTS] = frag_trT_off<RB, S * 32 * RB + DTS * 32>(vb)), ...);
}
template <int RB, int S, int NDT, int... DTS>
__device__ __forceinline__ void pv_step(const char* vb, int ns, const bf16x8 (&pb)[2][8], f32x4 (&ot)[2][NDT],
                                        bf16x8 (&cur)[NDT], bf16x8 (&nxt)[NDT], std::integer_sequence<int, DTS...> seq) {
  if (S < ns) {
    if constexpr (S + 1 < 8) {
      if (S + 1 < ns) pv_load<RB, S + 1, NDT>(vb, nxt, seq);
    }
    ((ot[0][DTS] = MFMA(cur[DTS], pb[0][S], ot[0][DTS]), ot[1][DTS] = MFMA(cur[DTS], pb[1][S], ot[1][DTS])), ...);
    ATTN_LDS_WAIT();
  }
}
template <int RB, int NDT>
__device__ __forceinline__ void pv_all(const char* vb, int ns, const bf16x8 (&pb)[2][8], f32x4 (&ot)[2][NDT]) {
  constexpr auto seq = std::make_integer_sequence<int, NDT>{};
  bf16x8 va[NDT], vb2[NDT];
  pv_load<RB, 0, NDT>(vb, va, seq);
  ATTN_LDS_WAIT();
  pv_step<RB, 0, NDT>(vb, ns, pb, ot, va, vb2, seq);
  pv_step<RB, 1, NDT>(vb, ns, pb, ot, vb2, va, seq);
  pv_step<RB, 2, NDT>(vb, ns, pb, ot, va, vb2, seq);
  pv_step<RB, 3, NDT>(vb, ns, pb, ot, vb2, va, seq);
  pv_step<RB, 4, NDT>(vb, ns, pb, ot, va, vb2, seq);
  pv_step<RB, 5, NDT>(vb, ns, pb, ot, vb2, va, seq);
  pv_step<RB, 6, NDT>(vb, ns, pb, ot, va, vb2, seq);
  pv_step<RB, 7, NDT>(vb, ns, pb, ot, vb2, va, seq);
}
__device__ __forceinline__ bf16x8 zero_frag() {
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
  return z;
}

// ------------------------------------------------------------------------------------------
// Round 4: the T <= 256 forward with every operand at least one phase further ahead (attn_fwd256p_kernel; the engine's form).
//
// Round 2's persistent kernel (three LDS tiles K | V | Q + a staging patch per wave; removed in round 5, 199-203 us at b = 256 against
// this kernel's 174-180) met at three barriers per item and could only ask for K(n+1), Q(n+1) once every wave was past
// S(n), and for V(n+1) once every wave is past PV(n): each tile has less than one item's time to arrive, the waves wait
// for it at the next barrier, and the per-CU vector-memory path (the kernel's floor: 180 KiB per item at ~10 B/clk) idles
// whenever a compute phase runs long.  Its instruction stream also carried a third of dead weight (ISA census, hd 72:
// 84 v_mov_b64 zeroing accumulators, 90 v_perm_b32 re-packing the 16-bit elements of every inline-asm transposed read, 132
// v_cndmask + 65 v_cmp of the ragged-key masks and ~90 scalar spill moves around the T-dependent branches, 128 v_add of
// the row sums: ~1200 vector instructions per wave and item for 176 MFMAs).  Here:
//   * LDS = K x 2 | V | Q: the K tile is double-buffered, K(n+2) is issued when every wave is past softmax(n) and is
//     first read 1.7 items later;
//   * Q is per wave: a wave DMAs its own 32 query rows into its own 5 KiB patch and reads them back itself (no barrier
//     for Q at all); Q(n+1) is issued as soon as Q(n)'s fragments are in registers, a whole item ahead;
//   * the output staging patch is the wave's own 32 rows of the V tile (dead after the barrier that ends PV), and a
//     wave's five V pieces are exactly those rows: V(n+1) is issued by each wave right after its own read-back, in FRONT
//     of its output stores, and is first read after S(n+1) AND softmax(n+1) (the V wait moved behind the softmax);
//   * two barriers per item (V landed + all past S; all past PV + K(n+1) landed) instead of three;
//   * T == 256 (FULL) is a separate instantiation without masks or T-dependent branches; the first MFMA of every
//     accumulator takes C = 0; the transposed reads return 32-bit pairs (no v_perm); for hd 72 the row sums come out of
//     the PV product itself: column 72 of the V tile (pad of the fifth 16-column output tile) is set to 1.0 by the wave
//     that owns the rows, so O^T row 72 = sum_k P[q][k] — the sum of the SAME bf16-rounded P the output is built from.
// Every vector-memory instruction is issued unconditionally (items past the end use an empty buffer descriptor: the
// range check turns them into zero fills and dropped stores), so every wait is one counted s_waitcnt vmcnt(N) with the
// same N for every wave and item.  Issue order per wave and item n, [count]:
//   start: Q(n+1)[5] | after softmax + barrier: K(n+2)[5] | end: V(n+1)[5], O rows + lse [5 + 2]
//   Q(n) landed  <=> at most K(n+1) 5 + V(n) 5 + stores(n-1) 7 = 17 younger
//   V(n) landed  <=> at most stores(n-1) 7 + Q(n+1) 5 = 12 younger
//   K(n+1) landed <=> at most V(n) 5 + stores(n-1) 7 + Q(n+1) 5 + K(n+2) 5 = 22 younger
#ifndef REED_ATTN_FWD_VDB_DEFAULT
#define REED_ATTN_FWD_VDB_DEFAULT 1
#endif
#ifndef REED_ATTN_KSPREAD   // 1: K(n + 2)'s pieces ride in the PV product's first steps — measured equal to the burst (profiles/r6_attn_fwd_kspread.txt)
#define REED_ATTN_KSPREAD 0
#endif
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ u32x2 tr16_u(unsigned a) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(OFF));
  return r;
}
template <int RB, int OFF>
__device__ __forceinline__ bf16x8 frag_trT_u(unsigned lane_base) {
  const u32x2 lo = tr16_u<OFF>(lane_base), hi = tr16_u<OFF + 16 * RB>(lane_base);
  const u32x4 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 lds_read128_asm(unsigned a) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(a) : "memory");
  return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(size_t)(const char __attribute__((address_space(3)))*)p;
}
template <int RB, int S, int NDT, int... DTS>
__device__ __forceinline__ void pvu_load(unsigned vb, bf16x8 (&vf)[NDT], std::integer_sequence<int, DTS...>) {
  ((vf[DTS] = frag_trT_u<RB, S * 32 * RB + DTS * 32>(vb)), ...);
}
// one 32-key step of O^T += V^T P^T: the V^T fragments of step S + 1 are read before the MFMAs of step S issue
template <int RB, int S, int NS, int NDT, int... DTS>
__device__ __forceinline__ void pvu_step(unsigned vb, const bf16x8 (&pb)[2][8], f32x4 (&ot)[2][NDT], bf16x8 (&cur)[NDT],
                                         bf16x8 (&nxt)[NDT], std::integer_sequence<int, DTS...> seq) {
  if constexpr (S < NS) {
    if constexpr (S + 1 < NS) pvu_load<RB, S + 1, NDT>(vb, nxt, seq);
    if constexpr (S == 0)
      ((ot[0][DTS] = MFMA(cur[DTS], pb[0][S], zero4()), ot[1][DTS] = MFMA(cur[DTS], pb[1][S], zero4())), ...);
    else
      ((ot[0][DTS] = MFMA(cur[DTS], pb[0][S], ot[0][DTS]), ot[1][DTS] = MFMA(cur[DTS], pb[1][S], ot[1][DTS])), ...);
    ATTN_LDS_WAIT();
  }
}
template <int RB, int NS, int NDT>
__device__ __forceinline__ void pvu_all(unsigned vb, const bf16x8 (&pb)[2][8], f32x4 (&ot)[2][NDT]) {
  constexpr auto seq = std::make_integer_sequence<int, NDT>{};
  bf16x8 va[NDT], vb2[NDT];
  pvu_load<RB, 0, NDT>(vb, va, seq);
  ATTN_LDS_WAIT();
  pvu_step<RB, 0, NS, NDT>(vb, pb, ot, va, vb2, seq);
  pvu_step<RB, 1, NS, NDT>(vb, pb, ot, vb2, va, seq);
  pvu_step<RB, 2, NS, NDT>(vb, pb, ot, va, vb2, seq);
  pvu_step<RB, 3, NS, NDT>(vb, pb, ot, vb2, va, seq);
  pvu_step<RB, 4, NS, NDT>(vb, pb, ot, va, vb2, seq);
  pvu_step<RB, 5, NS, NDT>(vb, pb, ot, vb2, va, seq);
  pvu_step<RB, 6, NS, NDT>(vb, pb, ot, va, vb2, seq);
  pvu_step<RB, 7, NS, NDT>(vb, pb, ot, vb2, va, seq);
}

// PV with the exponentials inside (FULL T = 256): P of 32-key step S + 1 is formed (16 v_exp, 16 v_fma, 8 v_cvt_pk per wave) and the
// V^T fragments of step S + 1 are read while the matrix pipe works on step S — vector ALU beside MFMA inside one wave, and the
// fp32 scores die step by step instead of all before the product (the allocator's peak).  SUM: row sums accumulated here (head
// sizes without a spare V column).
template <bool SUM>
__device__ __forceinline__ void px_make(const f32x4 (&st)[2][16], int s, const float (&mneg)[2], float sc2, bf16x8 (&pb)[2],
                                        float (&sum)[2], bool noexp) {
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    f32x4 a, b;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = __builtin_fmaf(st[qt][2 * s][r], sc2, -mneg[qt]), y = __builtin_fmaf(st[qt][2 * s + 1][r], sc2, -mneg[qt]);
      if (!noexp) { x = __builtin_amdgcn_exp2f(x); y = __builtin_amdgcn_exp2f(y); }
      a[r] = x;
      b[r] = y;
      if (SUM) sum[qt] += x + y;
    }
    pb[qt] = pack2(a, b);
  }
}
// HOOK: called once per 32-key step with the step's index as an integral_constant, in front of the step's reads (round 6: the five
// DMA pieces of K(n + 2) ride in steps 0..4 instead of going out as a burst in front of the product)
template <int RB, int S, int NDT, bool SUM, typename HOOK, int... DTS>
__device__ __forceinline__ void pvx_step(unsigned vb, const f32x4 (&st)[2][16], const float (&mneg)[2], float sc2, float (&sum)[2],
                                         bool noexp, f32x4 (&ot)[2][NDT], bf16x8 (&pc)[2], bf16x8 (&pn)[2], bf16x8 (&cur)[NDT],
                                         bf16x8 (&nxt)[NDT], HOOK& hook, std::integer_sequence<int, DTS...> seq) {
  hook(std::integral_constant<int, S>{});
  if constexpr (S + 1 < 8) {
    pvu_load<RB, S + 1, NDT>(vb, nxt, seq);
    px_make<SUM>(st, S + 1, mneg, sc2, pn, sum, noexp);
  }
  if constexpr (S == 0)
    ((ot[0][DTS] = MFMA(cur[DTS], pc[0], zero4()), ot[1][DTS] = MFMA(cur[DTS], pc[1], zero4())), ...);
  else
    ((ot[0][DTS] = MFMA(cur[DTS], pc[0], ot[0][DTS]), ot[1][DTS] = MFMA(cur[DTS], pc[1], ot[1][DTS])), ...);
  ATTN_LDS_WAIT();
}
template <int RB, int NDT, bool SUM, typename HOOK>
__device__ __forceinline__ void pvx_all(unsigned vb, const f32x4 (&st)[2][16], const float (&mneg)[2], float sc2, float (&sum)[2],
                                        bool noexp, f32x4 (&ot)[2][NDT], HOOK hook) {
  constexpr auto seq = std::make_integer_sequence<int, NDT>{};
  bf16x8 va[NDT], vb2[NDT], pa[2], pb2[2];
  pvu_load<RB, 0, NDT>(vb, va, seq);
  px_make<SUM>(st, 0, mneg, sc2, pa, sum, noexp);
  ATTN_LDS_WAIT();
  pvx_step<RB, 0, NDT, SUM>(vb, st, mneg, sc2, sum, noexp, ot, pa, pb2, va, vb2, hook, seq);
  pvx_step<RB, 1, NDT, SUM>(vb, st, mneg, sc2, sum, noexp, ot, pb2, pa, vb2, va, hook, seq);
  pvx_step<RB, 2, NDT, SUM>(vb, st, mneg, sc2, sum, noexp, ot, pa, pb2, va, vb2, hook, seq);
  pvx_step<RB, 3, NDT, SUM>(vb, st, mneg, sc2, sum, noexp, ot, pb2, pa, vb2, va, hook, seq);
  pvx_step<RB, 4, NDT, SUM>(vb, st, mneg, sc2, sum, noexp, ot, pa, pb2, va, vb2, hook, seq);
  pvx_step<RB, 5, NDT, SUM>(vb, st, mneg, sc2, sum, noexp, ot, pb2, pa, vb2, va, hook, seq);
  pvx_step<RB, 6, NDT, SUM>(vb, st, mneg, sc2, sum, noexp, ot, pa, pb2, va, vb2, hook, seq);
  pvx_step<RB, 7, NDT, SUM>(vb, st, mneg, sc2, sum, noexp, ot, pb2, pa, vb2, va, hook, seq);
}

template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128_off(unsigned a) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(OFF));
  return __builtin_bit_cast(bf16x8, r);
}
// S^T = K Q^T one 16-key tile at a time, software pipelined like the PV product: the K row fragments of tile KT + 1 are read
// (asm: unordered by the compiler, no wait of its own) before the MFMAs of tile KT issue, and waited for behind them.
// ka = tile + i * RB + 16 g (lane base), kaw = the same with lanes g >= 2 moved back two chunks (the k-step that covers columns
// 64..95 of a 72- / 80-wide head: columns 80..95 do not exist, those lanes re-read 64..79; their Q slots are zero)
template <int HD, int RB, int KT>
__device__ __forceinline__ void sk_load(unsigned ka, unsigned kaw, bf16x8 (&kf)[Cfg<HD>::KS]) {
  constexpr int KS = Cfg<HD>::KS;
  kf[0] = lds_read128_off<KT * 16 * RB>(ka);
  kf[1] = lds_read128_off<KT * 16 * RB + 64>(ka);
  if constexpr (KS == 3) kf[2] = lds_read128_off<KT * 16 * RB + 128>(HD % 32 != 0 ? kaw : ka);
}
// HOOK: called with the tile's index as an integral_constant in front of the tile's work (attn_fwd256v_kernel: the read-backs,
// DMA pieces and stores that ride in S)
struct SkNoHook {
  template <typename C>
  __device__ __forceinline__ void operator()(C) const {}
};
template <int HD, int RB, int KT, int NKT, typename HOOK>
__device__ __forceinline__ void sk_step(unsigned ka, unsigned kaw, const bf16x8 (&qf)[2][Cfg<HD>::KS], f32x4 (&st)[2][16],
                                        bf16x8 (&cur)[Cfg<HD>::KS], bf16x8 (&nxt)[Cfg<HD>::KS], HOOK& hook) {
  constexpr int KS = Cfg<HD>::KS;
  if constexpr (KT < NKT) {
    hook(std::integral_constant<int, KT>{});
    if constexpr (KT + 1 < NKT) sk_load<HD, RB, KT + 1>(ka, kaw, nxt);
    st[0][KT] = MFMA(cur[0], qf[0][0], zero4());
    st[1][KT] = MFMA(cur[0], qf[1][0], zero4());
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) {
      st[0][KT] = MFMA(cur[ks], qf[0][ks], st[0][KT]);
      st[1][KT] = MFMA(cur[ks], qf[1][ks], st[1][KT]);
    }
    ATTN_LDS_WAIT();
  }
}
template <int HD, int RB, typename HOOK = SkNoHook>
__device__ __forceinline__ void sk_all(unsigned ka, unsigned kaw, const bf16x8 (&qf)[2][Cfg<HD>::KS], f32x4 (&st)[2][16],
                                       HOOK hook = HOOK{}) {
  bf16x8 fa[Cfg<HD>::KS], fb[Cfg<HD>::KS];
  sk_load<HD, RB, 0>(ka, kaw, fa);
  ATTN_LDS_WAIT();
  sk_step<HD, RB, 0, 16>(ka, kaw, qf, st, fa, fb, hook);
  sk_step<HD, RB, 1, 16>(ka, kaw, qf, st, fb, fa, hook);
  sk_step<HD, RB, 2, 16>(ka, kaw, qf, st, fa, fb, hook);
  sk_step<HD, RB, 3, 16>(ka, kaw, qf, st, fb, fa, hook);
  sk_step<HD, RB, 4, 16>(ka, kaw, qf, st, fa, fb, hook);
  sk_step<HD, RB, 5, 16>(ka, kaw, qf, st, fb, fa, hook);
  sk_step<HD, RB, 6, 16>(ka, kaw, qf, st, fa, fb, hook);
  sk_step<HD, RB, 7, 16>(ka, kaw, qf, st, fb, fa, hook);
  sk_step<HD, RB, 8, 16>(ka, kaw, qf, st, fa, fb, hook);
  sk_step<HD, RB, 9, 16>(ka, kaw, qf, st, fb, fa, hook);
  sk_step<HD, RB, 10, 16>(ka, kaw, qf, st, fa, fb, hook);
  sk_step<HD, RB, 11, 16>(ka, kaw, qf, st, fb, fa, hook);
  sk_step<HD, RB, 12, 16>(ka, kaw, qf, st, fa, fb, hook);
  sk_step<HD, RB, 13, 16>(ka, kaw, qf, st, fb, fa, hook);
  sk_step<HD, RB, 14, 16>(ka, kaw, qf, st, fa, fb, hook);
  sk_step<HD, RB, 15, 16>(ka, kaw, qf, st, fb, fa, hook);
}
// two floats -> one dword of two 16-bit operands (v_cvt_pk_bf16_f32 / v_cvt_pkrtz... of the build's operand type)
__device__ __forceinline__ unsigned pk2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// DBGK: a diagnosis instantiation that takes `dbg` (bit 0: no S products, 1: no exponentials, 2: no PV products, 3: the output
// stores dropped by the range check, 4: every tile load an empty descriptor); the product instantiations ignore it
template <int HD, bool FULL, bool DBGK>
__global__ __launch_bounds__(512, 2) void attn_fwd256p_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                             float* __restrict__ lse, int T, int H, int nitems, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = Cfg<HD>::KS, DT = Cfg<HD>::DT, NCH = Cfg<HD>::NCH;
  constexpr bool ONES = (HD == 72);   // d = 72..79 is the zero pad of the fifth output tile: column 72 carries the row sums
  constexpr bool PVX = FULL && HD != 80;   // exponentials inside the PV product (pvx_all); hd 80's allocation spills with it
  if (!DBGK) dbg = 0;
  // dbg bit 5 (DBGK only): shader-clock time per phase, summed over the wave's items in registers and written over the start
  // of `lse` when the wave is done ([workgroup][wave][10] x u64; tools/attn_fwd_stamps.py) — no memory instruction inside the loop
  unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define ATTN_STAMP(k)                                               \
  do {                                                              \
    if (DBGK && (dbg & 32)) {                                       \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
      tacc[k] += t_ - tprev;                                        \
      tprev = t_;                                                   \
    }                                                               \
  } while (0)
  const int tid = threadIdx.x;
  const int lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (Round 4 measured a half-workgroup stagger — waves 4..7 form S(n+1) before storing item n, so that one wave of each SIMD
  // multiplies while its partner sits in the vector-memory queue: 20.5 k against 21.4 k cycles per item, kernel time equal within
  // the box noise; removed in round 5, profiles/r4_time_attn_stag_ab.txt.)
  if (DBGK && (dbg & 64) && wave >= 4) __builtin_amdgcn_s_setprio(1);    // experiment: static priority for the younger half
  if (DBGK && (dbg & 128) && wave < 4) __builtin_amdgcn_s_setprio(1);    // (control: for the older half)
  const int D = H * HD;
  const long tok = 3l * D;
  const int tokb = (int)(tok * 2);
  char* Kb = smem;                                     // two K tiles
  char* Vt = smem + 2 * TILE_F;
  char* Vw = Vt + wave * 32 * ROWF;                    // this wave's rows of the V tile = its five DMA pieces = its output patch
  char* Qw = smem + 3 * TILE_F + wave * 32 * ROWF;     // this wave's 32 query rows
  const int q0 = wave * 32;
  const float sc2 = rsqrtf((float)HD) * LOG2E;
  int voff[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) voff[j] = dma_voff<HD, ROWF>((wave * 5 + j) * 64 + lane0, tokb);
  const long win = (DBGK && (dbg & 16)) ? 0 : tile_window<HD>(T, tokb);
  // the wave's five pieces (its 32 rows) of a tile whose row 0 is `base`; base == nullptr: an empty descriptor (zero fill)
  auto issue = [&](char* wave_rows, const bf16* base) {
    const __amdgpu_buffer_rsrc_t rs = mk_rsrc(base, win);
#pragma unroll
    for (int j = 0; j < 5; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(wave_rows + j * 1024), 16, voff[j], 0, 0, REED_ATTN_LD_AUX);
  };
  auto issue_piece = [&](char* wave_rows, __amdgpu_buffer_rsrc_t rs, int j) {   // j: a literal after inlining
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(wave_rows + j * 1024), 16, voff[j], 0, 0, REED_ATTN_LD_AUX);
  };
  auto base_of = [&](int item) -> const bf16* {
    if (item >= nitems) return nullptr;
    const int b = item / H, h = item - b * H;
    return qkv + (long)b * T * tok + h * HD;
  };
  auto plus = [](const bf16* p, long n) -> const bf16* { return p ? p + n : nullptr; };
  const int G = gridDim.x;
  const int nsub = FULL ? 4 : (T + 63) >> 6;
  // "stores(-1)" + V of the first item: seven dropped stores, so that the counted waits hold from item 0 (distinct,
  // non-adjacent offsets and values: identical ones are merged into one instruction — tools/check_attn_isa.py counts them)
  auto first_v_and_stores = [&](const bf16* b0) {
    issue(Vw, plus(b0, 2 * D));
    const __amdgpu_buffer_rsrc_t none = mk_rsrc(nullptr, 0);
#pragma unroll
    for (int k = 0; k < 7; ++k) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(k + lane0), none, DMA_OOB - 256 * k, 0, 0);
  };
  f32x4 st[2][16];   // S^T of the item in hand
  // ---- Q(n): own patch, own wait; then S^T(n) = K(n) Q(n)^T from K buffer `kpar` (landed: the barrier that ended item n-1).
  auto q_and_s = [&](int n, int kpar) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));   // per-item copy of the lane id: hoisted lane-derived values are spilled around the loop
    const int i = lane & 15, g = lane >> 4;
    const char* Kt = Kb + kpar * TILE_F;
    asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(0);
    bf16x8 qf[2][KS];
    {
      const unsigned qa = lds_addr(Qw + i * ROWF + 16 * g);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          // columns 64..95 of a 72-wide head: 64..79 exist (72..79 = zero pad); lanes g >= 2 re-read them and are zeroed
          const bool wrap = HD % 32 != 0 && ks == KS - 1;
          const unsigned a = qa + qt * 16 * ROWF + ks * 64 - ((wrap && g >= 2) ? 32 : 0);
          qf[qt][ks] = lds_read128_asm(a);
        }
      ATTN_LDS_WAIT();
      if (HD % 32 != 0 && g >= 2) { qf[0][KS - 1] = zero_frag(); qf[1][KS - 1] = zero_frag(); }
    }
    issue(Qw, base_of(n + G));
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(1);
    if (DBGK && (dbg & 1)) {
#pragma unroll
      for (int kt = 0; kt < 16; ++kt) st[0][kt] = st[1][kt] = zero4();
    } else if (FULL) {
      const unsigned ka = lds_addr(Kt + i * ROWF + 16 * g);
      sk_all<HD, ROWF>(ka, ka - (g >= 2 ? 32 : 0), qf, st);
    } else {
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        if (sub < nsub) {
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const int kt = sub * 4 + k4;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              const bf16x8 kf = (HD % 32 != 0 && ks == KS - 1) ? frag_rows_f<true>(Kt, 16 * kt, ks, lane)
                                                               : frag_rows_f<false>(Kt, 16 * kt, ks, lane);
              if (ks == 0) {
                st[0][kt] = MFMA(kf, qf[0][ks], zero4());
                st[1][kt] = MFMA(kf, qf[1][ks], zero4());
              } else {
                st[0][kt] = MFMA(kf, qf[0][ks], st[0][kt]);
                st[1][kt] = MFMA(kf, qf[1][ks], st[1][kt]);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) st[0][sub * 4 + k4] = st[1][sub * 4 + k4] = zero4();
        }
      }
    }
    ATTN_STAMP(2);
  };
  // ---- output of item n in two parts.  stage_item: normalise, round and write O into the wave's own rows of the V tile (dead
  // after the barrier that ends PV) — the accumulators are free afterwards; store_item: whole 144-byte row pieces read back and
  // stored, V(n+1) into the rows just read back, in FRONT of the stores.
  // (round 5) the staging image inside the wave's 32 V rows uses 144-byte rows where the head fits them: the 16 rows of an 8-byte
  // write then share the write banks two by two instead of four by four (160-byte rows: 160 i mod 128 has four values), and the
  // read-back of hd 72's nine 16-byte chunks per row is one contiguous run
  constexpr int RST = HD <= 72 ? ROWB : ROWF;
  float lsev[2];
  auto stage_item = [&](const f32x4 (&ot)[2][DT], const float (&mrow)[2], const float (&lrow)[2]) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float l = lrow[qt];
      if (ONES) l = __shfl(ot[qt][DT - 1][0], 32 + i, 64);   // O^T row 72 (lanes g = 2, element 0) = sum_k P[q][k]
      const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = 16 * dt + 4 * g;
        const u32x2 v = {pk2(ot[qt][dt][0] * inv, ot[qt][dt][1] * inv), pk2(ot[qt][dt][2] * inv, ot[qt][dt][3] * inv)};
        if (16 * dt + 16 <= HD || d < HD) *(u32x2*)(Vw + (16 * qt + i) * RST + d * 2) = v;   // (only hd 72's fifth tile is partial)
      }
      lsev[qt] = mrow[qt] * LN2 + __logf(l);
    }
  };
  auto store_item = [&](int n) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int i = lane & 15, g = lane >> 4;
    const int b = n / H, h = n - b * H;
    // exactly 5 + 2 buffer stores per wave and item (counted by the waits); rows >= T and the lanes past the 32 x NCH chunks
    // fall outside the descriptors and are dropped by the range check
    const bool drop = DBGK && (dbg & 8);
    const __amdgpu_buffer_rsrc_t rsO = mk_rsrc(drop ? nullptr : o + (long)b * T * D + h * HD, tile_window<HD>(T, D * 2));
    const __amdgpu_buffer_rsrc_t rsL = mk_rsrc((lse && !drop) ? lse + ((long)b * H + h) * T : nullptr, (long)T * 4);
    constexpr int NQ = 32 * NCH;
    u32x4 piece[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int qi = min(lane + 64 * k, NQ - 1);
      const int rr = qi / NCH, c = qi - rr * NCH;
      piece[k] = __builtin_bit_cast(u32x4, lds_read128_asm(lds_addr(Vw + rr * RST + c * 16)));
    }
    ATTN_LDS_WAIT();
    issue(Vw, plus(base_of(n + G), 2 * D));
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int qi = lane + 64 * k;
      const int rr = qi / NCH, c = qi - rr * NCH;
      __builtin_amdgcn_raw_buffer_store_b128(piece[k], rsO, qi < NQ ? (q0 + rr) * (D * 2) + c * 16 : DMA_OOB, 0, 0);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lsev[qt]), rsL,
                                            g == 0 ? (q0 + 16 * qt + i) * 4 : DMA_OOB, 0, 0);
    ATTN_STAMP(9);
  };

  int it = xcd_contiguous(blockIdx.x, G);   // the 16 heads of a sample run on one XCD at about the same time
  {
    const bf16* b0 = base_of(it);
    issue(Qw, b0);
    issue(Kb + wave * 32 * ROWF, plus(b0, D));
    issue(Kb + TILE_F + wave * 32 * ROWF, plus(base_of(it + G), D));
    first_v_and_stores(b0);
    asm volatile("s_waitcnt vmcnt(17)" ::: "memory");   // Q, K of the first item (younger: K of the second, V, the 7 stores)
    ATTN_BARRIER();
  }
  if (DBGK && (dbg & 32)) tprev = __builtin_amdgcn_s_memtime();
  int par = 0;
  for (; it < nitems; it += G, par ^= 1) {
    q_and_s(it, par);
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int i = lane & 15, g = lane >> 4;
    // ---------------- softmax over the keys, in registers ----------------
    if (!FULL) {
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        if (64 * sub + 64 > T) {   // keys past T: a ragged or absent sub-block (wave-uniform)
          int gg = g;
          asm volatile("" : "+v"(gg));   // per-item value: keeps the 64 lane masks from being hoisted out of the item loop
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (64 * sub + 16 * k4 + 4 * gg + r >= T) st[qt][sub * 4 + k4][r] = -INFINITY;
        }
      }
    }
    float mrow[2], lrow[2];
    bf16x8 pb[2][8];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 16; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[qt][kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mneg = mx * sc2;   // the scale is positive: max of the scaled scores
      float sum = 0.f;
      mrow[qt] = mneg;
      lrow[qt] = 0.f;
      if (PVX) continue;             // the exponentials are formed inside the PV product (pvx_all)
#pragma unroll
      for (int kt = 0; kt < 16; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_fmaf(st[qt][kt][r], sc2, -mneg);
          if (!(DBGK && (dbg & 2))) p = __builtin_amdgcn_exp2f(p);
          st[qt][kt][r] = p;
          if (!ONES) sum += p;
        }
        if (kt & 1) pb[qt][kt >> 1] = pack2(st[qt][kt - 1], st[qt][kt]);
      }
      if (!ONES) {
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
      }
      mrow[qt] = mneg;
      lrow[qt] = sum;
      __builtin_amdgcn_sched_barrier(0);
    }
    ATTN_STAMP(3);
    // ---------------- V(n) landed; every wave is past S(n): K(n+2) may overwrite K(n) ----------------
    // younger than V(n): stores(n-1) 7 + Q(n+1) 5
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(4);
    if (ONES && lane < 32) *(bf16*)(Vw + lane * ROWF + HD * 2) = (bf16)1.0f;   // own rows, behind own pieces: the ones column
    ATTN_BARRIER();
    ATTN_STAMP(5);
    // K(n + 2): with the exponentials inside the PV product (PVX) its five pieces ride in the product's first five 32-key steps
    // — as a burst here the forty pieces of the eight waves queued in front of every wave's first MFMAs (REED_ATTN_KSPREAD=0: A/B)
    char* const k2rows = Kb + par * TILE_F + wave * 32 * ROWF;
    const __amdgpu_buffer_rsrc_t k2rs = mk_rsrc(plus(base_of(it + 2 * G), D), win);
    constexpr bool KSPREAD = PVX && REED_ATTN_KSPREAD;
    if constexpr (!KSPREAD) {
#pragma unroll
      for (int j = 0; j < 5; ++j) issue_piece(k2rows, k2rs, j);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- O^T = V^T P^T ----------------
    f32x4 ot[2][DT];
    {
      const unsigned vb = lds_addr(Vt + (4 * g + (i >> 2)) * ROWF + (i & 3) * 8);
      if (DBGK && (dbg & 4)) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ot[0][dt] = ot[1][dt] = f32x4{1.f, 1.f, 1.f, 1.f};
      } else if (FULL && !PVX) {
        pvu_all<ROWF, 8, DT>(vb, pb, ot);
      } else if (FULL) {
        pvx_all<ROWF, DT, !ONES>(vb, st, mrow, sc2, lrow, DBGK && (dbg & 2), ot, [&](auto s_c) {
          constexpr int S = decltype(s_c)::value;
          if constexpr (KSPREAD && S < 5) issue_piece(k2rows, k2rs, S);
        });
        if (!ONES) {
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            lrow[qt] += __shfl_xor(lrow[qt], 16, 64);
            lrow[qt] += __shfl_xor(lrow[qt], 32, 64);
          }
        }
      } else {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) ot[qt][dt] = zero4();
        pv_all<ROWF, DT>(Vt + (4 * g + (i >> 2)) * ROWF + (i & 3) * 8, (T + 31) >> 5, pb, ot);
      }
    }
    ATTN_STAMP(6);
    // K(n+1) landed (this wave's pieces); after the barrier: every wave's, and every wave is past its V reads
    asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP(7);
    ATTN_BARRIER();
    ATTN_STAMP(8);
    stage_item(ot, mrow, lrow);
    __builtin_amdgcn_sched_barrier(0);
    store_item(it);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero fills of the items past the end: LDS stays allocated until they land
  if (DBGK && (dbg & 32) && lse && lane0 == 0) {
    unsigned long long* dst = (unsigned long long*)lse + ((long)blockIdx.x * 8 + wave) * 10;
#pragma unroll
    for (int k = 0; k < 10; ++k) dst[k] = tacc[k];
  }
#undef ATTN_STAMP
}

// ------------------------------------------------------------------------------------------
// Round 6: the T = 256 forward with the V tile DOUBLE-BUFFERED (LDS = K | V x 2 | Q instead of K x 2 | V | Q; verdict item 2: spread
// the operand issue along the item).  attn_fwd256p_kernel ends every item with a burst — V(n+1) (five pieces) and the output of item
// n (five row pieces + lse) per wave: 12 of the item's 22 vector-memory instructions, issued by all eight waves behind the same
// barrier, through a memory path that takes ~12 B/clk per CU: 8.1 k of the younger waves' 21.4 k cycles per item with the matrix
// pipes idle (profiles/r4_attn_fwd_stamps.txt).  Here nothing is issued at the end of an item:
//   * O(n) is staged (normalised, rounded) into the wave's own 32 rows of V[n & 1] behind the barrier that ends PV(n), as before;
//   * its read-back and stores ride in S(n+1), one piece per two key tiles, and V(n+2) piece k is issued right behind read-back k
//     into the very bytes it freed (read-back piece k = DMA piece k: both are the k-th KiB of the wave's rows) — V(n+2) is first
//     read two PV products later;
//   * K is single-buffered: K(n+1) is issued behind the barrier that ends S(n) (every wave has read K(n)) and waited for behind
//     PV(n) — a whole PV product ahead of its first read.
// Issue order per wave and item n, [count]:  start: Q(n+1)[5] | in S(n): V(n+1)[5] interleaved with O(n-1) rows[5], lse(n-1)[2] |
// behind barrier 2: K(n+1)[5].  Counted waits (every vector-memory instruction is issued unconditionally; past the end: empty
// descriptors):  Q(n) landed <=> at most 12 + K(n) 5 = 17 younger;  V(n) landed <=> at most [store 1 + lse 2] + K(n) 5 + Q(n+1) 5 +
// 12 = 25 younger;  K(n+1) landed <=> 0 younger.  The prologue issues Q, K, V of the first item and EIGHT dropped stores so that
// the counts hold from the first item on.  T == 256, head sizes with the exponentials inside the PV product (64, 72) only.
// KSP: K(n+1)'s pieces ride in the PV product's first five steps; QSP: Q(n+1)'s in S(n)'s last five key tiles — with both the
// steady-state waits for Q and V are no-ops (everything was drained behind PV(n-1)); they order the first item.
// Measured at b = 256, hd 72 (profiles/r6_attn_fwd_vdb.txt, one box, alternating): attn_fwd256p_kernel 164-165 us; this kernel with
// the K and Q bursts kept 154-159; K spread 151-157; K and Q spread 148-150 us = 4.0-4.1 TB/s (the form launched).
template <int HD, bool KSP = true, bool QSP = true>
__global__ __launch_bounds__(512, 2) void attn_fwd256v_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                             float* __restrict__ lse, int H, int nitems) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int T = 256;
  constexpr int KS = Cfg<HD>::KS, DT = Cfg<HD>::DT, NCH = Cfg<HD>::NCH;
  constexpr bool ONES = (HD == 72);
  static_assert(HD == 64 || HD == 72, "attn_fwd256v_kernel: head_dim 64 or 72");
  const int tid = threadIdx.x;
  const int lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * HD;
  const long tok = 3l * D;
  const int tokb = (int)(tok * 2);
  char* const Kw = smem + wave * 32 * ROWF;                    // this wave's rows of the K tile
  char* const Vt = smem + TILE_F;                              // V[0], V[1]
  char* const Qw = smem + 3 * TILE_F + wave * 32 * ROWF;       // this wave's 32 query rows
  const int q0 = wave * 32;
  const float sc2 = rsqrtf((float)HD) * LOG2E;
  int voff[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) voff[j] = dma_voff<HD, ROWF>((wave * 5 + j) * 64 + lane0, tokb);
  const long win = tile_window<HD>(T, tokb);
  auto piece = [&](char* wave_rows, __amdgpu_buffer_rsrc_t rs, int j) {   // j: a literal after inlining
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(wave_rows + j * 1024), 16, voff[j], 0, 0, REED_ATTN_LD_AUX);
  };
  auto issue = [&](char* wave_rows, const bf16* base) {
    const __amdgpu_buffer_rsrc_t rs = mk_rsrc(base, win);
#pragma unroll
    for (int j = 0; j < 5; ++j) piece(wave_rows, rs, j);
  };
  auto base_of = [&](int item) -> const bf16* {
    if (item >= nitems) return nullptr;
    const int b = item / H, h = item - b * H;
    return qkv + (long)b * T * tok + h * HD;
  };
  auto plus = [](const bf16* p_, long n) -> const bf16* { return p_ ? p_ + n : nullptr; };
  const int G = gridDim.x;
  constexpr int RST = ROWB;             // staging rows inside the wave's 32 V rows (attn_fwd256p_kernel: round 5)
  constexpr int NQ = 32 * NCH;
  float lsev[2] = {0.f, 0.f};
  f32x4 st[2][16];

  int it = xcd_contiguous(blockIdx.x, G);
  {
    const bf16* b0 = base_of(it);
    issue(Qw, b0);
    issue(Kw, plus(b0, D));
    issue(Vt + wave * 32 * ROWF, plus(b0, 2 * D));
    const __amdgpu_buffer_rsrc_t none = mk_rsrc(nullptr, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(k + lane0), none, DMA_OOB - 256 * k, 0, 0);
    asm volatile("s_waitcnt vmcnt(13)" ::: "memory");   // Q, K of the first item (younger: its V, the 8 stores)
    ATTN_BARRIER();
  }
  int par = 0, pit = -1;   // V buffer of the item in hand; the previous item of this workgroup (its output is staged in V[par ^ 1])
  for (; it < nitems; it += G, par ^= 1) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int i = lane & 15, g = lane >> 4;
    char* const Vcur = Vt + par * TILE_F;
    char* const Vprev_w = Vt + (par ^ 1) * TILE_F + wave * 32 * ROWF;   // own rows: O(pit) staged; V(n+1) lands here
    // ---- Q(n): own patch, own wait
    asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 qf[2][KS];
    {
      const unsigned qa = lds_addr(Qw + i * ROWF + 16 * g);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bool wrap = HD % 32 != 0 && ks == KS - 1;
          const unsigned a = qa + qt * 16 * ROWF + ks * 64 - ((wrap && g >= 2) ? 32 : 0);
          qf[qt][ks] = lds_read128_asm(a);
        }
      ATTN_LDS_WAIT();
      if (HD % 32 != 0 && g >= 2) { qf[0][KS - 1] = zero_frag(); qf[1][KS - 1] = zero_frag(); }
    }
    const __amdgpu_buffer_rsrc_t rsQ = mk_rsrc(base_of(it + G), win);
    if constexpr (!QSP) {
#pragma unroll
      for (int j = 0; j < 5; ++j) piece(Qw, rsQ, j);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- S(n) = K(n) Q(n)^T; in it: the previous item's output out, V(n+1) in
    {
      const int pb_ = pit < 0 ? 0 : pit / H, ph = pit < 0 ? 0 : pit - pb_ * H;
      const __amdgpu_buffer_rsrc_t rsO = mk_rsrc(pit < 0 ? nullptr : o + (long)pb_ * T * D + ph * HD, tile_window<HD>(T, D * 2));
      const __amdgpu_buffer_rsrc_t rsL = mk_rsrc((lse && pit >= 0) ? lse + ((long)pb_ * H + ph) * T : nullptr, (long)T * 4);
      const __amdgpu_buffer_rsrc_t rsV = mk_rsrc(plus(base_of(it + G), 2 * D), win);
      u32x4 pc = {0u, 0u, 0u, 0u};
      const unsigned ka = lds_addr(smem + i * ROWF + 16 * g);
      sk_all<HD, ROWF>(ka, ka - (g >= 2 ? 32 : 0), qf, st, [&](auto kt_c) {
        constexpr int KT = decltype(kt_c)::value;
        if constexpr (KT < 10 && (KT & 1) == 0) {          // read-back of row piece k (waited for at the end of this tile)
          constexpr int k = KT / 2;
          const int qi = min(lane + 64 * k, NQ - 1);
          const int rr = qi / NCH, c = qi - rr * NCH;
          pc = __builtin_bit_cast(u32x4, lds_read128_asm(lds_addr(Vprev_w + rr * RST + c * 16)));
        } else if constexpr (KT < 10) {                    // V(n+1) piece k into the bytes just read; the row piece out
          constexpr int k = KT / 2;
          piece(Vprev_w, rsV, k);
          const int qi = lane + 64 * k;
          const int rr = qi / NCH, c = qi - rr * NCH;
          __builtin_amdgcn_raw_buffer_store_b128(pc, rsO, qi < NQ ? (q0 + rr) * (D * 2) + c * 16 : DMA_OOB, 0, 0);
        } else if constexpr (KT < 12) {
          constexpr int qt = KT - 10;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lsev[qt]), rsL,
                                                g == 0 ? (q0 + 16 * qt + i) * 4 : DMA_OOB, 0, 0);
        }
        if constexpr (QSP && KT >= 11) piece(Qw, rsQ, KT - 11);
      });
    }
    // ---- row maxima (the exponentials are formed inside the PV product)
    float mrow[2], lrow[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 16; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[qt][kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      mrow[qt] = mx * sc2;
      lrow[qt] = 0.f;
    }
    // ---- V(n) landed; every wave is past S(n): K(n+1) may overwrite K(n)
    asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (ONES && lane < 32) *(bf16*)(Vcur + (wave * 32 + lane) * ROWF + HD * 2) = (bf16)1.0f;   // own rows, behind own pieces
    ATTN_BARRIER();
    const __amdgpu_buffer_rsrc_t rsK = mk_rsrc(plus(base_of(it + G), D), win);
    if constexpr (!KSP) {
#pragma unroll
      for (int j = 0; j < 5; ++j) piece(Kw, rsK, j);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- O^T = V^T P^T
    f32x4 ot[2][DT];
    {
      const unsigned vb = lds_addr(Vcur + (4 * g + (i >> 2)) * ROWF + (i & 3) * 8);
      pvx_all<ROWF, DT, !ONES>(vb, st, mrow, sc2, lrow, false, ot, [&](auto s_c) {
        constexpr int S = decltype(s_c)::value;
        if constexpr (KSP && S < 5) piece(Kw, rsK, S);
      });
      if (!ONES) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          lrow[qt] += __shfl_xor(lrow[qt], 16, 64);
          lrow[qt] += __shfl_xor(lrow[qt], 32, 64);
        }
      }
    }
    // ---- K(n+1) landed (this wave's pieces); behind the barrier: every wave's, and every wave is past its reads of V[par]
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ATTN_BARRIER();
    // ---- O(n): normalised, rounded, into the wave's own rows of V[par]; it leaves in S of this workgroup's next item
    {
      char* const Vw = Vcur + wave * 32 * ROWF;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        float l = lrow[qt];
        if (ONES) l = __shfl(ot[qt][DT - 1][0], 32 + i, 64);
        const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d = 16 * dt + 4 * g;
          const u32x2 v = {pk2(ot[qt][dt][0] * inv, ot[qt][dt][1] * inv), pk2(ot[qt][dt][2] * inv, ot[qt][dt][3] * inv)};
          if (16 * dt + 16 <= HD || d < HD) *(u32x2*)(Vw + (16 * qt + i) * RST + d * 2) = v;
        }
        lsev[qt] = mrow[qt] * LN2 + __logf(l);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    pit = it;
  }
  // the last item's output (staged in V[par ^ 1] after the loop's last flip)
  if (pit >= 0) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int i = lane & 15, g = lane >> 4;
    const char* const Vw = Vt + (par ^ 1) * TILE_F + wave * 32 * ROWF;
    const int pb_ = pit / H, ph = pit - pb_ * H;
    const __amdgpu_buffer_rsrc_t rsO = mk_rsrc(o + (long)pb_ * T * D + ph * HD, tile_window<HD>(T, D * 2));
    const __amdgpu_buffer_rsrc_t rsL = mk_rsrc(lse ? lse + ((long)pb_ * H + ph) * T : nullptr, (long)T * 4);
    u32x4 pcs[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int qi = min(lane + 64 * k, NQ - 1);
      const int rr = qi / NCH, c = qi - rr * NCH;
      pcs[k] = __builtin_bit_cast(u32x4, lds_read128_asm(lds_addr(Vw + rr * RST + c * 16)));
    }
    ATTN_LDS_WAIT();
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int qi = lane + 64 * k;
      const int rr = qi / NCH, c = qi - rr * NCH;
      __builtin_amdgcn_raw_buffer_store_b128(pcs[k], rsO, qi < NQ ? (q0 + rr) * (D * 2) + c * 16 : DMA_OOB, 0, 0);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lsev[qt]), rsL, g == 0 ? (q0 + 16 * qt + i) * 4 : DMA_OOB, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero fills of the items past the end: LDS stays allocated until they land
}

// ------------------------------------------------------------------------------------------
// Backward, round 3: key-stationary, S and dP computed ONCE.
//
// The two-phase kernel above recomputes S = Q K^T, P and dP = dO V^T in both of its phases (7 matrix products and
// 2 x 65536 exponentials per head: phase 1 queries-on-wave for dQ, phase 2 keys-on-wave for dK / dV).  Here a wave owns 32
// keys for the whole item (K / V row fragments and the dK / dV accumulators in registers, exactly the old phase 2) and walks
// the queries 32 at a time; the dS block it forms on the way — [32 queries x its 32 keys], already rounded to bf16 for the
// dK product — is ALSO written to an LDS tile dS^T[key][query] (8 bytes per lane and MFMA tile: a lane owns 4 consecutive
// queries of one key).  After every 64 queries the workgroup meets at a barrier and forms dQ^T = K^T dS^T for those 64
// queries with the matrix pipe (A = K^T and B = dS^T both through ds_read_b64_tr_b16 with the same k-slot permutation;
// 20 (16-query tile, 16-column tile) pairs over 8 waves: 3 or 2 per wave, 8 k-steps each in one accumulator, i.e. the old
// phase 1's summation order over the keys).  5 products and 65536 exponentials per head: 448 MFMAs per wave and item instead
// of 624, half the softmax VALU work, 108 KiB of tiles per item by LDS-DMA instead of 144 (V never goes through LDS: a wave
// reads its own 32 rows from global memory straight into fragments).  dQ leaves through the Q tile's rows of the finished
// chunk (dead after the barrier), dK / dV through the wave's own rows of the K / dO tiles at the end: whole 144-byte row
// pieces.  Deterministic (no atomics; every output element has one owner and a fixed summation order).
// LDS: Q | K | dS^T (64 queries wide) | dO tiles of 144-byte rows + lse / delta = the same 147 KiB.  T <= 256.
template <int HD>
__global__ __launch_bounds__(512, 2) void attn_bwd_ks_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                             const bf16* __restrict__ d_o, const float* __restrict__ lse,
                                                             bf16* __restrict__ dqkv, int B, int T, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = Cfg<HD>::KS, DT = Cfg<HD>::DT, NCH = Cfg<HD>::NCH;
  constexpr int DTA = (DT + 1) / 2;          // 16-column tiles of dQ^T a wave of the first / second group forms (3 + 2 of 5)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int D = H * HD;
  const int bh = xcd_contiguous(blockIdx.x, gridDim.x);
  const int b = bh / H, h = bh % H;
  const long tok = 3l * D;
  const bf16* base = qkv + (long)b * T * tok + h * HD;
  bf16* dbase = dqkv + (long)b * T * tok + h * HD;
  const bf16* gbase = d_o + (long)b * T * D + h * HD;
  char* Qt = smem;
  char* Kt = smem + TILE_B;
  char* St = smem + 2 * TILE_B;   // dS^T[key][query - 64 chunk], bf16, ROWB-byte rows (64 of the 72 columns used)
  char* Gt = smem + 3 * TILE_B;   // dO
  float* lse2 = (float*)(smem + 4 * TILE_B + 256);
  float* dlt = lse2 + 256;
  // the 256 bytes behind the dO tile are read (x 0) by the k-step that covers columns 64..95 of the tile's last row: keep them
  // zero — LDS is not cleared between kernels, and a NaN bit pattern left there by another kernel times zero is a NaN
  if (tid < 16) *(uint4*)(smem + 4 * TILE_B + tid * 16) = make_uint4(0, 0, 0, 0);
  const float scale = rsqrtf((float)HD);
  const float sc2 = scale * LOG2E;
  const int r0 = wave * 32;       // this wave's rows: its keys, and the queries whose delta / log-sum-exp it prepares

  // ---- load phase: K, Q, dO tiles by LDS-DMA (36 pieces of 1 KiB each: waves 0..3 issue 5 of a tile, 4..7 issue 4) ----
  auto issue_tile = [&](char* t, const bf16* s, int sb) {
    const __amdgpu_buffer_rsrc_t rs = mk_rsrc(s, tile_window<HD>(T, sb));
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int I = wave + 8 * j;
      const int vo = dma_voff<HD, ROWB>(I * 64 + lane, sb);
      if (I < 36) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(t + I * 1024), 16, vo, 0, 0, REED_ATTN_LD_AUX);
    }
  };
  issue_tile(Kt, base + D, (int)(tok * 2));
  issue_tile(Qt, base, (int)(tok * 2));
  issue_tile(Gt, gbase, D * 2);
  // the wave's own rows from global memory: V fragments (its keys) and, for its 32 QUERY rows, dO and O -> delta, lse
  bf16x8 kf[2][KS], vf[2][KS];
  {
    bf16x8 gf[2][KS], of[2][KS];
    float lq[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int row = r0 + 16 * t2 + i;
      const bool ok = row < T;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        vf[t2][ks] = load_frag_global<HD>(base + 2 * D + (long)row * tok, ok, ks, lane);
        gf[t2][ks] = load_frag_global<HD>(gbase + (long)row * D, ok, ks, lane);
        of[t2][ks] = load_frag_global<HD>(o + ((long)b * T + row) * D + h * HD, ok, ks, lane);
      }
      lq[t2] = lse[((long)b * H + h) * T + min(row, T - 1)];
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      float acc = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += bf2f(gf[t2][ks][e]) * bf2f(of[t2][ks][e]);
      acc += __shfl_xor(acc, 16, 64);
      acc += __shfl_xor(acc, 32, 64);
      if (g == 0) {
        dlt[r0 + 16 * t2 + i] = acc;
        lse2[r0 + 16 * t2 + i] = (r0 + 16 * t2 + i < T) ? lq[t2] * LOG2E : INFINITY;   // rows >= T: p = exp2(-inf) = 0
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kf[ct][ks] = frag_rows_z<HD>(Kt, r0 + 16 * ct, ks, lane);
  bool kvalid[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) kvalid[ct] = r0 + 16 * ct + i < T;

  // whole 144-byte row pieces of `nrows` tile rows from row0 on -> global rows (16 bytes per lane)
  auto store_rows = [&](const char* tile, bf16* gb, int row0, int nrows) {
    const int nq = nrows * NCH;
    for (int k = 0; k * 64 < nq; ++k) {
      const int qi = lane + 64 * k;
      const int rr = qi / NCH, c = qi - rr * NCH;
      if (qi < nq && row0 + rr < T)
        *(uint4*)(gb + (long)(row0 + rr) * tok + c * 8) = *(const uint4*)(tile + (row0 + rr) * ROWB + c * 16);
    }
  };

  f32x4 dk[2][DT], dv[2][DT];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dk[ct][dt] = zero4(); dv[ct][dt] = zero4(); }
  const int nblk = (T + 31) >> 5;          // 32-query blocks (and 32-key steps) that hold rows < T
  const int nchunk = (nblk + 1) >> 1;
  for (int ch = 0; ch < nchunk; ++ch) {
    // ---------------- phase A: this wave's 32 keys x the chunk's 64 queries ----------------
#pragma unroll 1
    for (int qh = 0; qh < 2; ++qh) {
      const int qq0 = ch * 64 + qh * 32;
      f32x4 st[2][2], dp[2][2];  // [qt][ct]: rows q = qq0+16qt+4g+r, col kv = r0+16ct+i
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) { st[qt][ct] = zero4(); dp[qt][ct] = zero4(); }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 qa = frag_rows(Qt, qq0 + 16 * qt, ks, lane);
          bf16x8 ga = frag_rows(Gt, qq0 + 16 * qt, ks, lane);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            st[qt][ct] = MFMA(qa, kf[ct][ks], st[qt][ct]);
            dp[qt][ct] = MFMA(ga, vf[ct][ks], dp[qt][ct]);
          }
        }
      f32x4 lq4[2], dl4[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        lq4[qt] = *(const f32x4*)(lse2 + qq0 + 16 * qt + 4 * g);
        dl4[qt] = *(const f32x4*)(dlt + qq0 + 16 * qt + 4 * g);
      }
      bf16x8 pb[2], dsb[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x4 p0, p1, s0, s1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[0][ct][r], sc2, -lq4[0][r]));
          p1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[1][ct][r], sc2, -lq4[1][r]));
          s0[r] = kvalid[ct] ? p0[r] * (dp[0][ct][r] - dl4[0][r]) : 0.f;   // a key past T contributes nothing to dQ
          s1[r] = kvalid[ct] ? p1[r] * (dp[1][ct][r] - dl4[1][r]) : 0.f;
        }
        pb[ct] = pack2(p0, p1);
        dsb[ct] = pack2(s0, s1);
        // dS^T[key = r0+16ct+i][query = qq0 + 16 qt + 4g + r]: the lane's 4 consecutive queries of each 16-query tile
        char* sp = St + (r0 + 16 * ct + i) * ROWB + (qh * 32 + 4 * g) * 2;
        *(bf16x4*)sp = __builtin_shufflevector(dsb[ct], dsb[ct], 0, 1, 2, 3);
        *(bf16x4*)(sp + 32) = __builtin_shufflevector(dsb[ct], dsb[ct], 4, 5, 6, 7);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16x8 gtf = frag_trT(Gt, qq0, 16 * dt, lane);
        bf16x8 qtf = frag_trT(Qt, qq0, 16 * dt, lane);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          dv[ct][dt] = MFMA(gtf, pb[ct], dv[ct][dt]);
          dk[ct][dt] = MFMA(qtf, dsb[ct], dk[ct][dt]);
        }
      }
    }
    __syncthreads();   // dS^T of the chunk is complete; the chunk's Q rows are dead (phase A of later chunks reads other rows)
    // ---------------- phase B: dQ^T = K^T dS^T for the chunk's 64 queries ----------------
    {
      const int qtile = wave >> 1;                       // 16 queries of the chunk
      const int dt0 = (wave & 1) ? DTA : 0;              // this wave's 16-column tiles of head_dim: [dt0, dt0 + ndt)
      f32x4 dq[DTA];
#pragma unroll
      for (int k = 0; k < DTA; ++k) dq[k] = zero4();
#pragma unroll 1
      for (int ks = 0; ks < nblk; ++ks) {
        const bf16x8 dsf = frag_trT(St, 32 * ks, 16 * qtile, lane);
#pragma unroll
        for (int k = 0; k < DTA; ++k) {
          if (dt0 + k < DT) {
            const bf16x8 ktf = frag_trT(Kt, 32 * ks, 16 * (dt0 + k), lane);
            dq[k] = MFMA(ktf, dsf, dq[k]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < DTA; ++k) {
        const int d = 16 * (dt0 + k) + 4 * g;
        if (dt0 + k < DT && d < HD) {
          bf16x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = f2bf(dq[k][r] * scale);
          *(bf16x4*)(Qt + (ch * 64 + 16 * qtile + i) * ROWB + d * 2) = v;
        }
      }
    }
    __syncthreads();   // dQ rows of the chunk are staged; dS^T is free for the next chunk
    store_rows(Qt, dbase, ch * 64 + 8 * wave, 8);
  }
  // ---------------- dK, dV of the wave's keys leave through its own rows of the K / dO tiles ----------------
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = 16 * dt + 4 * g;
      if (d < HD) {
        bf16x4 a, c;
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[r] = f2bf(dk[ct][dt][r] * scale); c[r] = f2bf(dv[ct][dt][r]); }
        *(bf16x4*)(Kt + (r0 + 16 * ct + i) * ROWB + d * 2) = a;
        *(bf16x4*)(Gt + (r0 + 16 * ct + i) * ROWB + d * 2) = c;
      }
    }
  store_rows(Kt, dbase + D, r0, 32);
  store_rows(Gt, dbase + 2 * D, r0, 32);
}

// ------------------------------------------------------------------------------------------
// The key-stationary backward, persistent: one workgroup per CU walks the (batch, head) items and the NEXT item's operands
// arrive while the current one computes.  Measured on the one-shot form above (and on the two-phase kernel before it): with the
// compute skipped the kernel still takes half its time — every workgroup loads its 180 KiB in one burst at ~11 B/clk per CU
// and waits, then computes with the memory system idle (one 147-KiB workgroup per CU: nothing to switch to).
// What makes the overlap possible here is that in the key-stationary order the Q and dO tiles are STREAMS — the 64 rows of
// chunk c are read in phase A of chunk c and never again — and only K is resident:
//   * the 64 rows of chunk c of Q(n+1), dO(n+1) are issued (LDS-DMA, 18 pieces of 1 KiB) after the barrier that ends phase A of
//     chunk c of item n: each lands a whole item before it is read;
//   * the wave's own key rows of item n+1 — K and V as MFMA fragments — and the (log-sum-exp, delta) pairs of its 32 query
//     rows are plain global loads into registers (inline asm: the compiler never waits for them) issued at that same point,
//     into the registers of the current item's fragments (dead: phase A is over);
//   * K(n+1) (the K^T operand of phase B) is issued when phase B of the last chunk is over and is first needed after
//     phase A of chunk 0 of item n+1;
//   * dQ leaves straight from the MFMA layout (8 bytes per lane; the Q tile's rows belong to the next item by then),
//     dK / dV through the wave's own rows of the dS^T tile (dead after the last phase B), one after the other.
// delta = rowsum(dO * O) comes from a row kernel in front (attn_delta_kernel: it reads the bytes of dO and O this kernel no
// longer reads — the traffic is the same — and keeps 48 fragment registers per lane out of the pipeline).
// Two waits per item, both counted when T = 256 (s_waitcnt vmcnt(N): loads, stores and LDS-DMA retire in issue order): at the
// end of the item — everything older than the K(n+1) issue has landed, K itself and the dK / dV row stores behind it stay in
// flight — and before the first phase B of the next item (K landed; the stores stay in flight).  Ragged T: the same code with
// both waits as vmcnt(0).
__device__ __forceinline__ bf16x8 gload128_asm(const void* p) {
  bf16x8 r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ float gload32_asm(const void* p) {
  float r;
  asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}

// delta[b, h, t] = sum_d dO[b, t, h, d] * O[b, t, h, d]  (fp32): one thread per (token, head) segment of HD elements
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ o, const bf16* __restrict__ d_o,
                                                         float* __restrict__ delta, long nseg, int T, int H) {
  const long seg = (long)blockIdx.x * 256 + threadIdx.x;   // (b * T + t) * H + h
  if (seg >= nseg) return;
  bf16x8 a[HD / 8], c[HD / 8];
#pragma unroll
  for (int k = 0; k < HD / 8; ++k) {
    a[k] = *(const bf16x8*)(o + seg * HD + 8 * k);
    c[k] = *(const bf16x8*)(d_o + seg * HD + 8 * k);
  }
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < HD / 8; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += bf2f(a[k][e]) * bf2f(c[k][e]);
  const long tokn = seg / H;
  const int h = (int)(seg - tokn * H);
  const long b = tokn / T;
  const int t = (int)(tokn - b * T);
  delta[(b * H + h) * T + t] = acc;
}

// delta[b, h, t] from the partial dot products the dO-producing GEMM left (reed_gemm epilogue 13): dpart f32 [H, S, B * T]
__global__ __launch_bounds__(256) void attn_delta_combine_kernel(const float* __restrict__ dpart, float* __restrict__ delta,
                                                                 long ntok, int T, int H, int S) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // h * ntok + (b * T + t)
  if (i >= ntok * H) return;
  const int h = (int)(i / ntok);
  const long tokn = i - h * ntok;
  float acc = dpart[(long)h * S * ntok + tokn];
  if (S == 2) acc += dpart[((long)h * S + 1) * ntok + tokn];
  const long b = tokn / T;
  const int t = (int)(tokn - b * T);
  delta[(b * H + h) * T + t] = acc;
}

template <int HD>
__global__ __launch_bounds__(512, 2) void attn_bwd_ksp_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              bf16* __restrict__ dqkv, int T, int H, int nitems, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = Cfg<HD>::KS, DT = Cfg<HD>::DT, NCH = Cfg<HD>::NCH;
  constexpr int DTA = (DT + 1) / 2;
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * HD;
  const long tok = 3l * D;
  const int tokb = (int)(tok * 2), db = D * 2;
  char* Qt = smem;
  char* Kt = smem + TILE_B;
  char* St = smem + 2 * TILE_B;
  char* Gt = smem + 3 * TILE_B;
  float* lse2 = (float*)(smem + 4 * TILE_B + 256);
  float* dlt = lse2 + 256;
  if (tid < 16) *(uint4*)(smem + 4 * TILE_B + tid * 16) = make_uint4(0, 0, 0, 0);   // read (x 0) behind the dO tile's last row
  const float scale = rsqrtf((float)HD);
  const float sc2 = scale * LOG2E;
  const int r0 = wave * 32;
  const bool full = T == 256;      // counted waits (the store counts below assume every row exists)
  const int nblk = (T + 31) >> 5, nchunk = (nblk + 1) >> 1;

  // 36 pieces of 1 KiB: rows 0..127 (half 0) / 128..255 (half 1) of two tiles, or (whole) one whole tile; waves 0..3 issue 5
  // pieces, waves 4..7 issue 4
  auto issue36 = [&](char* t0, const bf16* s0, int sb0, char* t1, const bf16* s1, int sb1, int half, bool whole, int ln) {
    const __amdgpu_buffer_rsrc_t rs0 = mk_rsrc(s0, tile_window<HD>(T, sb0)), rs1 = mk_rsrc(s1, tile_window<HD>(T, sb1));
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int pp = wave + 8 * j;                 // 0..35 (36..39: not issued)
      const bool second = !whole && pp >= 18;
      const int I = whole ? pp : (second ? pp - 18 : pp) + 18 * half;
      const int vo = dma_voff<HD, ROWB>(I * 64 + ln, second ? sb1 : sb0);
      if (pp < 36) __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? rs1 : rs0, (lds_ptr_t)((second ? t1 : t0) + I * 1024), 16, vo, 0, 0, REED_ATTN_LD_AUX);
    }
  };
  // the 64 rows of chunk c of the Q and dO tiles (9 + 9 pieces): wave w issues pieces w and w + 8, waves 0 / 1 also 16 / 17
  auto issue_chunk = [&](const bf16* qb, const bf16* gb2, int c, int ln) {
    const __amdgpu_buffer_rsrc_t rs0 = mk_rsrc(qb, tile_window<HD>(T, tokb)), rs1 = mk_rsrc(gb2, tile_window<HD>(T, db));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pp = wave + 8 * j;                 // 0..17 (18..23: not issued)
      const bool second = pp >= 9;
      const int I = 9 * c + (second ? pp - 9 : pp);
      const int vo = dma_voff<HD, ROWB>(I * 64 + ln, second ? db : tokb);
      if (pp < 18) __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? rs1 : rs0, (lds_ptr_t)((second ? Gt : Qt) + I * 1024), 16, vo, 0, 0, REED_ATTN_LD_AUX);
    }
  };
  auto bases = [&](int item, const bf16*& base, const bf16*& gbase, const float*& lbase, const float*& dlbase, bf16*& dbase) {
    const int b = item / H, h = item - b * H;
    base = qkv + (long)b * T * tok + h * HD;
    dbase = dqkv + (long)b * T * tok + h * HD;
    gbase = d_o + (long)b * T * D + h * HD;
    lbase = lse + ((long)b * H + h) * T;
    dlbase = delta + ((long)b * H + h) * T;
  };
  // the wave's own rows of an item: K, V fragments of its 32 keys, and (lse, delta) of its 32 query rows (lane i + 16 g' of
  // the first 32 lanes holds row r0 + i + 16 g') -> registers
  bf16x8 kf[2][KS], vf[2][KS];
  float lq, dq_;
  auto own_rows = [&](const bf16* base, const float* lbase, const float* dlbase, int ln) {
    const int i = ln & 15, g = ln >> 4;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int row = min(r0 + 16 * t2 + i, T - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int col = min(ks * 32 + 8 * g, HD - 8);      // clamped: the fragment is zeroed below when it does not exist
        kf[t2][ks] = gload128_asm(base + D + (long)row * tok + col);
        vf[t2][ks] = gload128_asm(base + 2 * D + (long)row * tok + col);
      }
    }
    const int row = min(r0 + (ln & 31), T - 1);
    lq = gload32_asm(lbase + row);
    dq_ = gload32_asm(dlbase + row);
  };
  // after the loads landed: zero what does not exist; lse (in log2 units) and delta of the wave's 32 query rows -> LDS
  auto own_rows_finish = [&](int ln) {
    const int i = ln & 15, g = ln >> 4;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const bool rok = r0 + 16 * t2 + i < T;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        if (!(rok && ks * 32 + 8 * g < HD)) { kf[t2][ks] = zero_frag(); vf[t2][ks] = zero_frag(); }
    }
    if (ln < 32) {
      const int row = r0 + ln;
      dlt[row] = dq_;
      lse2[row] = row < T ? lq * LOG2E : INFINITY;    // rows >= T: p = exp2(-inf) = 0
    }
  };

  int it = xcd_contiguous(blockIdx.x, gridDim.x);
  const bf16 *base, *gbase;
  const float *lbase, *dlbase;
  bf16* dbase;
  if (it < nitems) {
    bases(it, base, gbase, lbase, dlbase, dbase);
    issue36(Kt, base + D, tokb, Kt, base + D, tokb, 0, true, lane0);
    issue36(Qt, base, tokb, Gt, gbase, db, 0, false, lane0);
    issue36(Qt, base, tokb, Gt, gbase, db, 1, false, lane0);
    own_rows(base, lbase, dlbase, lane0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    own_rows_finish(lane0);
  }
  for (; it < nitems; it += gridDim.x) {
    const int nxt = it + gridDim.x;
    const bool has_next = nxt < nitems;
    const bf16 *nbase = nullptr, *ngbase = nullptr;
    const float *nlbase = nullptr, *ndlbase = nullptr;
    bf16* ndbase = nullptr;
    if (has_next) bases(nxt, nbase, ngbase, nlbase, ndlbase, ndbase);
    int lane = lane0;
    asm volatile("" : "+v"(lane));   // per-item copy: lane-derived addresses are not hoisted out of the item loop (and spilled)
    const int i = lane & 15, g = lane >> 4;
    bool kvalid[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) kvalid[ct] = r0 + 16 * ct + i < T;
    ATTN_BARRIER();    // item boundary: lse2 / dlt of this item are written, every wave is done with the previous item's tiles
    f32x4 dk[2][DT], dv[2][DT];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { dk[ct][dt] = zero4(); dv[ct][dt] = zero4(); }
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ++ch) {
      // ---------------- phase A: this wave's 32 keys x the chunk's 64 queries ----------------
#pragma unroll 1
      for (int qh = 0; qh < ((dbg & 1) ? 0 : 2); ++qh) {   // dbg bit 0 (diagnosis only, tools/time_attn.py): skip phase A
        const int qq0 = ch * 64 + qh * 32;
        f32x4 st[2][2], dp[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) { st[qt][ct] = zero4(); dp[qt][ct] = zero4(); }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            bf16x8 qa = frag_rows(Qt, qq0 + 16 * qt, ks, lane);
            bf16x8 ga = frag_rows(Gt, qq0 + 16 * qt, ks, lane);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
              st[qt][ct] = MFMA(qa, kf[ct][ks], st[qt][ct]);
              dp[qt][ct] = MFMA(ga, vf[ct][ks], dp[qt][ct]);
            }
          }
        f32x4 lq4[2], dl4[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          lq4[qt] = *(const f32x4*)(lse2 + qq0 + 16 * qt + 4 * g);
          dl4[qt] = *(const f32x4*)(dlt + qq0 + 16 * qt + 4 * g);
        }
        bf16x8 pb[2], dsb[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          f32x4 p0, p1, s0, s1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[0][ct][r], sc2, -lq4[0][r]));
            p1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[1][ct][r], sc2, -lq4[1][r]));
            s0[r] = kvalid[ct] ? p0[r] * (dp[0][ct][r] - dl4[0][r]) : 0.f;
            s1[r] = kvalid[ct] ? p1[r] * (dp[1][ct][r] - dl4[1][r]) : 0.f;
          }
          pb[ct] = pack2(p0, p1);
          dsb[ct] = pack2(s0, s1);
          char* sp = St + (r0 + 16 * ct + i) * ROWB + (qh * 32 + 4 * g) * 2;
          *(bf16x4*)sp = __builtin_shufflevector(dsb[ct], dsb[ct], 0, 1, 2, 3);
          *(bf16x4*)(sp + 32) = __builtin_shufflevector(dsb[ct], dsb[ct], 4, 5, 6, 7);
        }
        // dV^T += dO^T P, dK^T += Q^T dS: the transposing reads are inline asm (the builtin would drain the LDS-DMA queue),
        // two groups of 16-column tiles, each waited for as a whole
#pragma unroll
        for (int d0 = 0; d0 < DT; d0 += 3) {
          bf16x8 gtf[3], qtf[3];
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (d0 + k < DT) {
              gtf[k] = frag_trT_a<ROWB>(Gt, qq0, 16 * (d0 + k), lane);
              qtf[k] = frag_trT_a<ROWB>(Qt, qq0, 16 * (d0 + k), lane);
            }
          ATTN_LDS_WAIT();
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (d0 + k < DT) {
#pragma unroll
              for (int ct = 0; ct < 2; ++ct) {
                dv[ct][d0 + k] = MFMA(gtf[k], pb[ct], dv[ct][d0 + k]);
                dk[ct][d0 + k] = MFMA(qtf[k], dsb[ct], dk[ct][d0 + k]);
              }
            }
        }
      }
      if (ch == 0) {   // K of this item (the K^T operand of phase B) landed; younger: the previous item's dK / dV row stores
        if (full) {
          if constexpr (2 * ((32 * NCH + 63) / 64) == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // hd 72: 2 x 5
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                   // hd 64: 2 x 4
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      ATTN_BARRIER();   // dS^T of the chunk complete; rows [64 ch, 64 ch + 64) of the Q and dO tiles are dead
      const bool last = ch == nchunk - 1;
      if (has_next) {
        // the next item's Q / dO rows of this chunk (dead from here on) — a whole item before they are read; at the last chunk
        // also whatever rows a short T never reached, and the wave's own rows (kf / vf of this item are dead: phase A is over)
        int ln = lane0;
        asm volatile("" : "+v"(ln));   // fresh copy: the piece offsets and row addresses are computed here, not kept in registers
        issue_chunk(nbase, ngbase, ch, ln);
        if (last) {
          for (int c2 = ch + 1; c2 < 4; ++c2) issue_chunk(nbase, ngbase, c2, ln);
          own_rows(nbase, nlbase, ndlbase, ln);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---------------- phase B: dQ^T = K^T dS^T for the chunk's 64 queries ----------------
      {
        const int qtile = wave >> 1;
        const int dt0 = (wave & 1) ? DTA : 0;
        f32x4 dq[DTA];
#pragma unroll
        for (int k = 0; k < DTA; ++k) dq[k] = zero4();
#pragma unroll 1
        for (int ks = 0; ks < ((dbg & 2) ? 0 : nblk); ++ks) {   // dbg bit 1: skip phase B's products
          const bf16x8 dsf = frag_trT_a<ROWB>(St, 32 * ks, 16 * qtile, lane);
          bf16x8 ktf[DTA];
#pragma unroll
          for (int k = 0; k < DTA; ++k) ktf[k] = frag_trT_a<ROWB>(Kt, 32 * ks, 16 * min(dt0 + k, DT - 1), lane);
          ATTN_LDS_WAIT();
#pragma unroll
          for (int k = 0; k < DTA; ++k)
            if (dt0 + k < DT) dq[k] = MFMA(ktf[k], dsf, dq[k]);
        }
        // straight from the MFMA layout: rows d = 16 dt + 4 g + r, column q = i -> 8 bytes per lane.  A wave whose group has
        // one tile fewer stores its last tile twice (same bytes): every wave issues DTA stores per chunk
        const int q = ch * 64 + 16 * qtile + i;
#pragma unroll
        for (int k = 0; k < DTA; ++k) {
          const int kk = (dt0 + k < DT) ? k : DT - 1 - dt0;
          const int d = 16 * (dt0 + kk) + 4 * g;
          bf16x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = f2bf(dq[kk][r] * scale);
          if (d < HD && q < T) *(bf16x4*)(dbase + (long)q * tok + d) = v;
        }
      }
      ATTN_BARRIER();   // dS^T is free for the next chunk (after the last chunk: K and dS^T tiles are dead)
    }
    int le = lane0;
    asm volatile("" : "+v"(le));
    if (has_next) {
      issue36(Kt, nbase + D, tokb, Kt, nbase + D, tokb, 0, true, le);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---------------- dK, then dV, of the wave's keys through its own rows of the dS^T tile ----------------
    auto stage_store = [&](const f32x4 (&acc)[2][DT], float mul, bf16* gb) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d = 16 * dt + 4 * g;
          if (d < HD) {
            bf16x4 a;
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = f2bf(acc[ct][dt][r] * mul);
            *(bf16x4*)(St + (r0 + 16 * ct + i) * ROWB + d * 2) = a;
          }
        }
      // (asm LDS reads: behind a plain read the compiler drains the LDS-DMA of the next item's K issued just above)
      constexpr int NQ = 32 * NCH, NI = (NQ + 63) / 64;
      uint4 piece[NI];
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int qi = min(le + 64 * k, NQ - 1);
        const int rr = qi / NCH, c = qi - rr * NCH;
        const unsigned a = (unsigned)(size_t)(const char __attribute__((address_space(3)))*)(St + (r0 + rr) * ROWB + c * 16);
        asm volatile("ds_read_b128 %0, %1" : "=v"(piece[k]) : "v"(a) : "memory");
      }
      ATTN_LDS_WAIT();
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int qi = le + 64 * k;
        const int rr = qi / NCH, c = qi - rr * NCH;
        if (qi < NQ && r0 + rr < T) *(uint4*)(gb + (long)(r0 + rr) * tok + c * 8) = piece[k];
      }
    };
    stage_store(dk, scale, dbase + D);
    stage_store(dv, 1.f, dbase + 2 * D);
    if (has_next) {
      // everything older than the K issue has landed — the next item's Q / dO chunks and the wave's own rows; younger and still
      // in flight: K's pieces (5 for waves 0..3, 4 for 4..7) and the dK / dV row stores issued since
      __builtin_amdgcn_sched_barrier(0);
      if (full) {
        constexpr int NST = 2 * ((32 * NCH + 63) / 64);
        if constexpr (NST == 10) {
          if (wave < 4) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        } else {
          if (wave < 4) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      own_rows_finish(le);    // -> kf, vf, lse2, dlt of the next item
      base = nbase; gbase = ngbase; lbase = nlbase; dlbase = ndlbase; dbase = ndbase;
    }
  }
}

// Phase A of the ring kernel (round 4).  ring_a_load<HD, QO>: the Q and dO row fragments of the 16-query tile that starts at row QO
// of the ring slot (lane (i, g): row QO + i, columns ks * 32 + 8 g .. + 7; the k-step that covers columns 64..95 of a 72-wide head
// reads 64..79 again in lanes g >= 2 — qbw / gbw — whose partner slots in kf / vf are zero).
template <int HD, int QO>
__device__ __forceinline__ void ring_a_load(unsigned qb, unsigned qbw, unsigned gb, unsigned gbw, bf16x8 (&qa)[Cfg<HD>::KS],
                                            bf16x8 (&ga)[Cfg<HD>::KS]) {
  constexpr int KS = Cfg<HD>::KS;
  qa[0] = lds_read128_off<QO * ROWF>(qb);
  ga[0] = lds_read128_off<QO * ROWF>(gb);
  qa[1] = lds_read128_off<QO * ROWF + 64>(qb);
  ga[1] = lds_read128_off<QO * ROWF + 64>(gb);
  if constexpr (KS == 3) {
    qa[2] = lds_read128_off<QO * ROWF + 128>(HD % 32 != 0 ? qbw : qb);
    ga[2] = lds_read128_off<QO * ROWF + 128>(HD % 32 != 0 ? gbw : gb);
  }
}
template <int HD>
__device__ __forceinline__ void ring_a_sdp(const bf16x8 (&qa)[Cfg<HD>::KS], const bf16x8 (&ga)[Cfg<HD>::KS],
                                           const bf16x8 (&kf)[2][Cfg<HD>::KS], const bf16x8 (&vf)[2][Cfg<HD>::KS], f32x4 (&st)[2],
                                           f32x4 (&dp)[2]) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    st[ct] = MFMA(qa[0], kf[ct][0], zero4());
    dp[ct] = MFMA(ga[0], vf[ct][0], zero4());
  }
#pragma unroll
  for (int ks = 1; ks < Cfg<HD>::KS; ++ks)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      st[ct] = MFMA(qa[ks], kf[ct][ks], st[ct]);
      dp[ct] = MFMA(ga[ks], vf[ct][ks], dp[ct]);
    }
}
// one 32-query half (QH = 0 / 1) of a chunk: on entry qa[0] / ga[0] hold its first tile's fragments (read and waited for)
template <int HD, int QH>
__device__ __forceinline__ void ring_a_half(unsigned qb, unsigned qbw, unsigned gb, unsigned gbw, bf16x8 (&qa)[2][Cfg<HD>::KS],
                                            bf16x8 (&ga)[2][Cfg<HD>::KS], const bf16x8 (&kf)[2][Cfg<HD>::KS],
                                            const bf16x8 (&vf)[2][Cfg<HD>::KS], f32x4 (&dk)[2][Cfg<HD>::DT], f32x4 (&dv)[2][Cfg<HD>::DT],
                                            char* St, const char* Qs, const char* Gs, const float* lse2, const float* dlt, float sc2,
                                            int q0c, int r0, int lane) {
  constexpr int DT = Cfg<HD>::DT;
  const int i = lane & 15, g = lane >> 4;
  const int ql = QH * 32, qq0 = q0c + ql;
  f32x4 st[2][2], dp[2][2];
  ring_a_load<HD, QH * 32 + 16>(qb, qbw, gb, gbw, qa[1], ga[1]);   // the second tile's fragments fly under the first tile's MFMAs
  ring_a_sdp<HD>(qa[0], ga[0], kf, vf, st[0], dp[0]);
  ATTN_LDS_WAIT();
  f32x4 lq4[2], dl4[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    lq4[qt] = *(const f32x4*)(lse2 + qq0 + 16 * qt + 4 * g);
    dl4[qt] = *(const f32x4*)(dlt + qq0 + 16 * qt + 4 * g);
  }
  ring_a_sdp<HD>(qa[1], ga[1], kf, vf, st[1], dp[1]);
  bf16x8 pb[2], dsb[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    f32x4 p0, p1, s0, s1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[0][ct][r], sc2, -lq4[0][r]));
      p1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[1][ct][r], sc2, -lq4[1][r]));
      s0[r] = p0[r] * (dp[0][ct][r] - dl4[0][r]);
      s1[r] = p1[r] * (dp[1][ct][r] - dl4[1][r]);
    }
    pb[ct] = pack2(p0, p1);
    dsb[ct] = pack2(s0, s1);
    // dS^T tile (round 5): 128-byte rows (64 queries).  Logical (row r, 32-byte segment s, 8-byte slot p) lives at segment
    // s ^ ((r >> 1) & 3), slot p ^ ((r & 1) | ((r >> 3) & 1) << 1): the 8 rows a half-wave of a transposing read touches (banks
    // mod 64) lie on 8 disjoint 8-bank ranges, and the 16 rows a 16-lane group of this 8-byte write touches (banks mod 32:
    // MI355X_MICROARCH.md, LDS) on 16 distinct slots of the 128-byte bank space.  (144-byte rows: a read's eighth row wrapped onto
    // its first row's banks and a write's rows i, i + 8 shared a slot; profiles/r5_attn_bwd_lds.txt)
    const int f = (i >> 1) & 3, hsl = (i & 1) | (((i >> 3) & 1) << 1);   // (r0 + 16 ct is a multiple of 16)
    char* srow = St + (r0 + 16 * ct + i) * 128 + ((g ^ hsl) << 3);
    *(bf16x4*)(srow + (((2 * QH) ^ f) << 5)) = __builtin_shufflevector(dsb[ct], dsb[ct], 0, 1, 2, 3);
    *(bf16x4*)(srow + (((2 * QH + 1) ^ f) << 5)) = __builtin_shufflevector(dsb[ct], dsb[ct], 4, 5, 6, 7);
  }
  // dO^T / Q^T fragments of the half (transposed reads), then — for the first half — the next half's first tile, all in flight
  // under the MFMAs of the previous group
#pragma unroll
  for (int d0 = 0; d0 < DT; d0 += 3) {
    bf16x8 gtf[3], qtf[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (d0 + k < DT) {
        gtf[k] = frag_trT_a<ROWF>(Gs, ql, 16 * (d0 + k), lane);
        qtf[k] = frag_trT_a<ROWF>(Qs, ql, 16 * (d0 + k), lane);
      }
    if (QH == 0 && d0 + 3 >= DT) ring_a_load<HD, 32>(qb, qbw, gb, gbw, qa[0], ga[0]);
    ATTN_LDS_WAIT();
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (d0 + k < DT) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          dv[ct][d0 + k] = MFMA(gtf[k], pb[ct], dv[ct][d0 + k]);
          dk[ct][d0 + k] = MFMA(qtf[k], dsb[ct], dk[ct][d0 + k]);
        }
      }
  }
}

// s_waitcnt vmcnt(n) for a wave-uniform n that is only known per wave class (the instruction takes an immediate)
__device__ __forceinline__ void vmcnt_wait(int n) {
  switch (n) {
#define REED_VMC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    REED_VMC(1) REED_VMC(2) REED_VMC(3) REED_VMC(4) REED_VMC(5) REED_VMC(6) REED_VMC(7) REED_VMC(8) REED_VMC(9) REED_VMC(10)
    REED_VMC(11) REED_VMC(12) REED_VMC(13) REED_VMC(14) REED_VMC(15) REED_VMC(16) REED_VMC(17) REED_VMC(18) REED_VMC(19) REED_VMC(20)
#undef REED_VMC
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// Phase B of the ring kernel, round 4: dQ^T = K^T dS^T for one 16-query tile and this wave's column tiles (even waves: the first
// DTA 16-column tiles, odd waves: the rest), software pipelined like the forward's PV product — the transposed fragments of
// key step ks + 1 (inline asm, 32-bit pairs, compile-time offsets from ONE lane base per tile) are read before the MFMAs of step
// ks issue and waited for behind them; the loop of round 3 read, waited and multiplied in turn (8 exposed LDS round trips).
// dQ then leaves in 16-byte lanes: v_permlane16_swap exchanges the odd 16-lane rows of tile k with the even rows of tile k + 1,
// after which lane row g holds 8 consecutive columns (16 (g & 1) + 8 (g >> 1) of the tile pair) of its query — one dwordx4
// store covers 64 contiguous bytes of each of 16 query rows, where the MFMA layout gave two stores of 32-byte pieces.
template <int HD, bool ODD>
struct RingB {
  static constexpr int DT = Cfg<HD>::DT, DTA = (DT + 1) / 2;
  static constexpr int NK = ODD ? DT - DTA : DTA;      // column tiles of this wave class (hd 72: 3 | 2, hd 64: 2 | 2)
  static constexpr int D0 = ODD ? DTA : 0;
  static constexpr int NST = (NK + 1) / 2;             // store instructions per chunk: pairs as dwordx4, a last odd tile as dwordx2
  template <int KS>
  static __device__ __forceinline__ void load(unsigned sb, unsigned kb, bf16x8& dsf, bf16x8 (&ktf)[NK]) {
    dsf = frag_trT_u<128, KS * 32 * 128>(sb);
    ktf[0] = frag_trT_u<ROWF, KS * 32 * ROWF>(kb);
    if constexpr (NK > 1) ktf[1] = frag_trT_u<ROWF, KS * 32 * ROWF + 32>(kb);
    if constexpr (NK > 2) ktf[2] = frag_trT_u<ROWF, KS * 32 * ROWF + 64>(kb);
  }
  template <int KS>
  static __device__ __forceinline__ void step(unsigned sb, unsigned kb, f32x4 (&dq)[NK], bf16x8& dsc, bf16x8 (&kc)[NK], bf16x8& dsn,
                                              bf16x8 (&kn)[NK]) {
    if constexpr (KS + 1 < 8) load<KS + 1>(sb, kb, dsn, kn);
#pragma unroll
    for (int k = 0; k < NK; ++k) dq[k] = (KS == 0) ? MFMA(kc[k], dsc, zero4()) : MFMA(kc[k], dsc, dq[k]);
    ATTN_LDS_WAIT();
  }
  // St / Kt: the dS^T tile (128-byte rows, swizzled 32-byte segments) and the K tile (160-byte rows); qrow0: dqkv row of the tile's first query (q part); returns after the stores
  static __device__ __forceinline__ void run(const char* St, const char* Kt, int qtile, int lane, float scale, bf16* qrow0, long tok,
                                             bool skip) {
    const int i = lane & 15, g = lane >> 4;
    const unsigned sb = lds_addr(St + (4 * g + (i >> 2)) * 128 + ((qtile ^ ((2 * g + (i >> 3)) & 3)) << 5) +
                                 (((i & 3) ^ (((i >> 2) & 1) | ((g >> 1) << 1))) << 3));
    const unsigned kb = lds_addr(Kt + (4 * g + (i >> 2)) * ROWF + (16 * D0 + 4 * (i & 3)) * 2);
    f32x4 dq[NK];
    if (skip) {   // diagnosis: no products
#pragma unroll
      for (int k = 0; k < NK; ++k) dq[k] = zero4();
    } else {
      bf16x8 da, db, ka[NK], kb2[NK];
      load<0>(sb, kb, da, ka);
      ATTN_LDS_WAIT();
      step<0>(sb, kb, dq, da, ka, db, kb2);
      step<1>(sb, kb, dq, db, kb2, da, ka);
      step<2>(sb, kb, dq, da, ka, db, kb2);
      step<3>(sb, kb, dq, db, kb2, da, ka);
      step<4>(sb, kb, dq, da, ka, db, kb2);
      step<5>(sb, kb, dq, db, kb2, da, ka);
      step<6>(sb, kb, dq, da, ka, db, kb2);
      step<7>(sb, kb, dq, db, kb2, da, ka);
    }
    u32x2 v[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) v[k] = u32x2{pk2(dq[k][0] * scale, dq[k][1] * scale), pk2(dq[k][2] * scale, dq[k][3] * scale)};
    bf16* row = qrow0 + (long)i * tok;
#pragma unroll
    for (int k = 0; k + 1 < NK; k += 2) {
      const auto r0 = __builtin_amdgcn_permlane16_swap(v[k][0], v[k + 1][0], false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(v[k][1], v[k + 1][1], false, false);
      const int d = 16 * (D0 + k) + 16 * (g & 1) + 8 * (g >> 1);
      const u32x4 w = {r0[0], r1[0], r0[1], r1[1]};
      if (16 * (D0 + k) + 32 <= HD || d < HD) *(u32x4*)(row + d) = w;   // (hd 72: columns 72..79 of the last pair do not exist)
    }
    if constexpr (NK & 1) {
      const int d = 16 * (D0 + NK - 1) + 4 * g;
      if (16 * (D0 + NK) <= HD || d < HD) *(u32x2*)(row + d) = v[NK - 1];
    }
  }
};

// ------------------------------------------------------------------------------------------
// The persistent key-stationary backward for T = 256 with the operand traffic SPREAD over the item (the form the engine runs).
// attn_bwd_ksp_kernel above still asks for half of an item's bytes at its very end — the wave's own K / V rows (72 KiB per
// CU), the last Q / dO rows and K(n+1) can only be issued when phase A of the last chunk has released their registers and
// tile rows — and the memory system serves that burst at ~3.5 TB/s while every CU waits: with the products skipped the kernel
// takes 340 us, and the products' 190 us ADD to it (REED_ATTN_KSP_DBG).  Here nothing but the wave's own 32 V
// rows is asked for late:
//   * Q and dO are STREAMS (the 64 rows of a chunk are read in its phase A and never again): two 64-row ring slots each
//     instead of whole tiles, chunk G + 2 issued into the slot phase A of chunk G has just released — across item boundaries;
//   * the 36 KiB that frees hold a second K tile: K(n+1) is issued during chunk 0 of item n and serves both as the K^T operand
//     of phase B and, through plain fragment reads at the start of item n+1, as the wave's own K rows;
//   * only the V fragments of the wave's 32 keys (and the lse / delta scalars of its 32 query rows) come straight from global
//     memory into registers, issued when phase A of the last chunk is over.
// Every wait of the item is a counted s_waitcnt vmcnt(N) (loads, stores and LDS-DMA retire in issue order).  A chunk of Q + dO is
// 20 pieces of 1 KiB (64 rows of 160 bytes, twice) and a K tile 36: waves 0..3 issue 3 of a chunk and 5 of a K tile, waves 4..7
// issue 2 and 4, so N depends on the wave's half.  Per item and wave, in issue order:
//   c=0: X0[3|2] K(n+1)[5|4] dQ[NQ] | c=1: X1[3|2] dQ[NQ] | c=2: X2[3|2] dQ[NQ] | c=3: X3[3|2] own[8] dQ[NQ] | dK, dV rows[2 NI]
//   (X0, X1 = chunks 2, 3 of this item; X2, X3 = chunks 0, 1 of the next; NQ = the wave's dQ store instructions per chunk, RingB::NST:
//   hd 72 even waves 2, odd 1; hd 64: 1).  Chunk 2 is needed when chunk 1 ends: X0 landed <=> at most K + X + 2 NQ younger
//   operations outstanding; chunk 3 when chunk 2 ends: X1 <=> X + 2 NQ; the item-end wait (the own rows landed: at most
//   NQ + 2 NI younger) covers X2, X3 and K(n+1).  (Round 3 hard-coded the hd-72 counts 14 / 12, 9 / 8 and 3 + 2 NI for both
//   head sizes: with hd 64's two dQ stores per chunk its waits allowed two operations too many — fixed with the counts computed.)
// The ring slots use 160-byte rows (ROWF: conflict-free ds_read_b128 and transposing reads; with 144-byte rows 7 of the 8 rows
// a ds_read_b128 lane group takes at g = 1 share banks with its g = 0 rows).  Round 5: the two K tiles too (the 8 KiB the
// kernel had left: 40 pieces per tile, five per wave — the transposing reads of phase B's K^T operand and the waves' own K row
// fragments no longer wrap a row onto the first row's banks); the dS^T tile keeps 144-byte rows: 160 would need 4 KiB the CU
// does not have — so it went the other way: 128-byte rows (its payload: 64 queries) with the 32-byte segments XOR-swizzled by the
// row pair, and the dK / dV staging moved to the item's dead K tile.  LDS: Q ring 20 | dO ring 20 | K x 2 80 | dS^T 32 | lse, delta x 2 4 = 156 KiB.
template <int HD, bool STAMPS = false>
__global__ __launch_bounds__(512, 2) void attn_bwd_ring_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               bf16* __restrict__ dqkv, int H, int nitems, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int T = 256;
  constexpr int KS = Cfg<HD>::KS, DT = Cfg<HD>::DT, NCH = Cfg<HD>::NCH;
  constexpr int DTA = (DT + 1) / 2;
  constexpr int CHB = 64 * ROWF;                    // one ring slot: 64 rows of 160 bytes (conflict-free ds_read_b128, see ROWF)
  constexpr int NI = (32 * NCH + 63) / 64;          // row-store instructions per wave for dK (and for dV)
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * HD;
  const long tok = 3l * D;
  const int tokb = (int)(tok * 2), db = D * 2;
  char* Qr = smem;
  char* Gr = smem + 2 * CHB;
  char* Kb = smem + 4 * CHB;                        // two K tiles
  char* St = Kb + 2 * TILE_F;
  float* ld2 = (float*)(St + 256 * 128);            // [2 items][lse2[256] | dlt[256]]
  const float scale = rsqrtf((float)HD);
  const float sc2 = scale * LOG2E;
  const int r0 = wave * 32;
  if ((dbg & 8) && wave >= 4) __builtin_amdgcn_s_setprio(1);    // experiment (REED_ATTN_KSP_DBG bit 3): static priority for waves 4..7
  if ((dbg & 16) && wave < 4) __builtin_amdgcn_s_setprio(1);    // (bit 4: for waves 0..3)
  // STAMPS (diagnosis instantiation, dbg bit 2): shader-clock time per phase summed over the wave's items in registers, written over
  // the start of dqkv when the wave is done ([workgroup][wave][8] x u64; tools/attn_bwd_stamps.py); no memory instruction in the loops
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define RING_STAMP(k)                                               \
  do {                                                              \
    if (STAMPS && (dbg & 4)) {                                      \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
      tacc[k] += t_ - tprev;                                        \
      tprev = t_;                                                   \
    }                                                               \
  } while (0)

  // chunk `c` (64 rows) of the Q and dO tiles of an item -> ring slot c & 1: 10 + 10 pieces of 1 KiB (16-byte lanes; a 12-byte
  // lane — buffer_load_dwordx3 ... lds — would give every wave the same piece count, but the hardware puts each lane's 12 bytes
  // in a 16-byte LDS slot: tools/micro/dma12.hip).  Wave w issues pieces w and w + 8, waves 0..3 also 16..19
  auto issue_chunk = [&](const bf16* qb, const bf16* gb2, int c, int ln) {
    const __amdgpu_buffer_rsrc_t rs0 = mk_rsrc(qb, tile_window<HD>(T, tokb)), rs1 = mk_rsrc(gb2, tile_window<HD>(T, db));
    char* q = Qr + (c & 1) * CHB;
    char* g2 = Gr + (c & 1) * CHB;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pp = wave + 8 * j;                  // 0..19 (20..23: not issued)
      const bool second = pp >= 10;
      const int I = second ? pp - 10 : pp;          // piece inside the chunk
      const int sb = second ? db : tokb;
      const int vo = dma_voff<HD, ROWF>(I * 64 + ln, sb);
      const int v2 = vo == DMA_OOB ? DMA_OOB : vo + c * 64 * sb;
      if (pp < 20) __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? rs1 : rs0, (lds_ptr_t)((second ? g2 : q) + I * 1024), 16, v2, 0, 0, REED_ATTN_LD_AUX);
    }
  };
  // a whole K tile (160-byte rows): 40 pieces, every wave issues 5
  auto issue_k = [&](const bf16* kb, char* dst, int ln) {
    const __amdgpu_buffer_rsrc_t rs = mk_rsrc(kb, tile_window<HD>(T, tokb));
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int I = wave + 8 * j;
      const int vo = dma_voff<HD, ROWF>(I * 64 + ln, tokb);   // (not inside the builtin's argument list: clang's host pass then drops the kernel)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + I * 1024), 16, vo, 0, 0, REED_ATTN_LD_AUX);
    }
  };
  auto bases = [&](int item, const bf16*& base, const bf16*& gbase, const float*& lbase, const float*& dlbase, bf16*& dbase) {
    const int b = item / H, h = item - b * H;
    base = qkv + (long)b * T * tok + h * HD;
    dbase = dqkv + (long)b * T * tok + h * HD;
    gbase = d_o + (long)b * T * D + h * HD;
    lbase = lse + ((long)b * H + h) * T;
    dlbase = delta + ((long)b * H + h) * T;
  };
  bf16x8 kf[2][KS], vf[2][KS];
  float lq, dq_;
  // the wave's own rows of an item that do not come through LDS: V fragments of its 32 keys, (lse, delta) of its 32 query rows
  auto own_rows = [&](const bf16* base, const float* lbase, const float* dlbase, int ln) {
    const int i = ln & 15, g = ln >> 4;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        vf[t2][ks] = gload128_asm(base + 2 * D + (long)(r0 + 16 * t2 + i) * tok + min(ks * 32 + 8 * g, HD - 8));
    lq = gload32_asm(lbase + r0 + (ln & 31));
    dq_ = gload32_asm(dlbase + r0 + (ln & 31));
  };
  auto own_rows_finish = [&](int ln, float* ld) {
    const int g = ln >> 4;
    if (HD % 32 != 0) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
        if ((KS - 1) * 32 + 8 * g >= HD) vf[t2][KS - 1] = zero_frag();
    }
    if (ln < 32) {
      ld[r0 + ln] = lq * LOG2E;
      ld[256 + r0 + ln] = dq_;
    }
  };

  int it = xcd_contiguous(blockIdx.x, gridDim.x);
  const bf16 *base, *gbase;
  const float *lbase, *dlbase;
  bf16* dbase;
  if (it < nitems) {
    bases(it, base, gbase, lbase, dlbase, dbase);
    issue_k(base + D, Kb, lane0);
    issue_chunk(base, gbase, 0, lane0);
    issue_chunk(base, gbase, 1, lane0);
    own_rows(base, lbase, dlbase, lane0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    own_rows_finish(lane0, ld2);
  }
  int par = 0;     // item parity: K tile and lse / delta set of the current item
  for (; it < nitems; it += gridDim.x, par ^= 1) {
    const int nxt = it + gridDim.x;
    const bool has_next = nxt < nitems;
    const bf16 *nbase = nullptr, *ngbase = nullptr;
    const float *nlbase = nullptr, *ndlbase = nullptr;
    bf16* ndbase = nullptr;
    if (has_next) bases(nxt, nbase, ngbase, nlbase, ndlbase, ndbase);
    int lane = lane0;
    asm volatile("" : "+v"(lane));   // per-item copy: lane-derived addresses are not hoisted out of the item loop (and spilled)
    const int i = lane & 15, g = lane >> 4;
    const char* Kt = Kb + par * TILE_F;
    const float* lse2 = ld2 + par * 512;
    const float* dlt = lse2 + 256;
    if (STAMPS && (dbg & 4)) tprev = __builtin_amdgcn_s_memtime();
    ATTN_BARRIER();    // item boundary: K, lse / delta of this item are in LDS (every wave waited for its pieces and wrote its rows)
    RING_STAMP(0);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kf[ct][ks] = frag_rows_z<HD, ROWF>(Kt, r0 + 16 * ct, ks, lane);
    f32x4 dk[2][DT], dv[2][DT];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { dk[ct][dt] = zero4(); dv[ct][dt] = zero4(); }
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
      const char* Qs = Qr + (ch & 1) * CHB;
      const char* Gs = Gr + (ch & 1) * CHB;
      // ---------------- phase A: this wave's 32 keys x the chunk's 64 queries ----------------
      // Round 4: the two 32-query halves are unrolled and their LDS reads run ahead of the matrix work — the Q / dO row fragments
      // of the second 16-query tile are read (inline asm, compile-time offsets from one lane base) while the 12 S / dP MFMAs of
      // the first issue, and the first tile's fragments of the NEXT half are read before the 20 dV / dK MFMAs of this one.  In round
      // 3 every group of four fragment reads was waited for right before its four MFMAs (four exposed LDS round trips per half).
      if (!(dbg & 1)) {   // dbg bit 0 (diagnosis only): skip phase A
        const unsigned qb = lds_addr(Qs + i * ROWF + 16 * g), gb = lds_addr(Gs + i * ROWF + 16 * g);
        const unsigned qbw = qb - ((HD % 32 != 0 && g >= 2) ? 32 : 0), gbw = gb - ((HD % 32 != 0 && g >= 2) ? 32 : 0);
        bf16x8 qa[2][KS], ga[2][KS];   // [set][ks]: set 0 = the 16-query tile in the matrix pipe, set 1 = the one being read
        ring_a_load<HD, 0>(qb, qbw, gb, gbw, qa[0], ga[0]);
        ATTN_LDS_WAIT();
        ring_a_half<HD, 0>(qb, qbw, gb, gbw, qa, ga, kf, vf, dk, dv, St, Qs, Gs, lse2, dlt, sc2, ch * 64, r0, lane);
        ring_a_half<HD, 1>(qb, qbw, gb, gbw, qa, ga, kf, vf, dk, dv, St, Qs, Gs, lse2, dlt, sc2, ch * 64, r0, lane);
      }
      RING_STAMP(1);
      ATTN_BARRIER();   // dS^T of the chunk complete; the ring slot is released
      RING_STAMP(2);
      if (ch < 2 || has_next) {
        int ln = lane0;
        asm volatile("" : "+v"(ln));   // fresh copy: piece offsets and row addresses are computed here, not kept in registers
        if (ch < 2) issue_chunk(base, gbase, ch + 2, ln);
        else issue_chunk(nbase, ngbase, ch - 2, ln);
        if (has_next && ch == 0) issue_k(nbase + D, Kb + (par ^ 1) * TILE_F, ln);
        if (has_next && ch == 3) own_rows(nbase, nlbase, ndlbase, ln);   // vf of this item is dead: phase A is over
        __builtin_amdgcn_sched_barrier(0);
      }
      // (Round 6 moved the issue block above INTO phase B — one piece per key step through a hook of RingB::step, same instructions
      // in the same order, the own rows kept as a burst at ch = 3 — and measured the kernel and the step equal: 395 / 400 / 405 us
      // burst, 395 / 406 / 392 spread; the 14-17 k cycles of "issue + phase B" are the product's transposing LDS reads, not the
      // issue.  profiles/r6_attn_bwd_spread.txt, commit a468645; removed.)
      // ---------------- phase B: dQ^T = K^T dS^T for the chunk's 64 queries ----------------
      {
        bf16* qrow0 = dbase + (long)(ch * 64 + 16 * (wave >> 1)) * tok;
        if (wave & 1) RingB<HD, true>::run(St, Kt, wave >> 1, lane, scale, qrow0, tok, (dbg & 2) != 0);   // dbg bit 1: no products
        else RingB<HD, false>::run(St, Kt, wave >> 1, lane, scale, qrow0, tok, (dbg & 2) != 0);
      }
      // the next chunk's Q / dO rows (issued one chunk ago) have landed: counted, see the table in the header
      __builtin_amdgcn_sched_barrier(0);
      RING_STAMP(3);
      if (has_next) {
        // X = this wave's pieces of a chunk (3 | 2 by half), Kp = of a K tile (5), NQ = its dQ stores per chunk (by parity)
        const int X = wave < 4 ? 3 : 2, Kp = 5;
        const int NQ = (wave & 1) ? RingB<HD, true>::NST : RingB<HD, false>::NST;
        if (ch == 1) vmcnt_wait(Kp + X + 2 * NQ);      // chunk 2 (X0) landed; younger: K(n+1), dQ, X1, dQ
        else if (ch == 2) vmcnt_wait(X + 2 * NQ);      // chunk 3 (X1) landed; younger: dQ, X2, dQ
      } else if (ch == 1 || ch == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the last item: nothing is issued behind it
      }
      RING_STAMP(4);
      ATTN_BARRIER();   // dS^T is free for the next chunk; the next chunk's ring slot is complete
      RING_STAMP(5);
    }
    // ---------------- dK, then dV, of the wave's keys through its own rows of this item's K tile ----------------
    // (dead since the barrier that ended the last chunk's phase B; K(n+2) is issued into it behind the next item-boundary barrier.
    // Until round 5 the dS^T tile served here, which kept its rows at 144 bytes: a head's 72 columns.  The staging image keeps
    // 144-byte rows inside the 40 KiB buffer: with 160 the 16 rows of an 8-byte write share four slots of the write banks)
    char* Ks = Kb + par * TILE_F;
    int le = lane0;
    asm volatile("" : "+v"(le));
    auto stage_store = [&](const f32x4 (&acc)[2][DT], float mul, bf16* gb) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d = 16 * dt + 4 * g;
          if (d < HD) {
            bf16x4 a;
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = f2bf(acc[ct][dt][r] * mul);
            *(bf16x4*)(Ks + (r0 + 16 * ct + i) * ROWB + d * 2) = a;
          }
        }
      constexpr int NQ = 32 * NCH;
      uint4 piece[NI];
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int qi = min(le + 64 * k, NQ - 1);
        const int rr = qi / NCH, c = qi - rr * NCH;
        const unsigned a = (unsigned)(size_t)(const char __attribute__((address_space(3)))*)(Ks + (r0 + rr) * ROWB + c * 16);
        asm volatile("ds_read_b128 %0, %1" : "=v"(piece[k]) : "v"(a) : "memory");
      }
      ATTN_LDS_WAIT();
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int qi = le + 64 * k;
        const int rr = qi / NCH, c = qi - rr * NCH;
        if (qi < NQ) *(uint4*)(gb + (long)(r0 + rr) * tok + c * 8) = piece[k];
      }
    };
    stage_store(dk, scale, dbase + D);
    stage_store(dv, 1.f, dbase + 2 * D);
    RING_STAMP(6);
    if (has_next) {
      // the wave's own rows of the next item have landed (and with them everything older: X3, K(n+1)); younger and still in
      // flight: the last chunk's dQ stores and the dK / dV row stores
      __builtin_amdgcn_sched_barrier(0);
      vmcnt_wait(((wave & 1) ? RingB<HD, true>::NST : RingB<HD, false>::NST) + 2 * NI);
      __builtin_amdgcn_sched_barrier(0);
      own_rows_finish(le, ld2 + (par ^ 1) * 512);
      base = nbase; gbase = ngbase; lbase = nlbase; dlbase = ndlbase; dbase = ndbase;
    }
    RING_STAMP(7);
  }
  if (STAMPS && (dbg & 4) && lane0 == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* dst = (unsigned long long*)dqkv + ((long)blockIdx.x * 8 + wave) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = tacc[k];
  }
#undef RING_STAMP
}

template <typename K>
int set_lds(K kernel, int bytes) {
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { reed_set_error("hipFuncSetAttribute(LDS=%d): %s", bytes, hipGetErrorString(e)); return (int)e; }
  return 0;
}

}  // namespace
int reed_concurrent_comm();   // gemm256.hip
int reed_num_cus();           // gemm256.hip: the device's CUs minus the reserve (reed_set_cu_reserve) — the persistent kernels' grids
                              // leave the reserved CUs to whatever the caller runs beside them (RCCL channels; a co-running kernel)
namespace {
int num_cus() { return reed_num_cus(); }

}  // namespace

extern "C" int reed_attention_fwd(const void* qkv, void* o, float* lse, int B, int T, int H, int hd,
                                  void* stream) {
  REED_CHECK_ARG(qkv && o, "attention_fwd: null pointer");
  REED_CHECK_ARG(hd == 64 || hd == 72 || hd == 80, "attention: head_dim %d unsupported (64, 72 or 80)", hd);
  REED_CHECK_ARG(hd != 80 || T <= 256, "attention: head_dim 80 (the I-JEPA ViT-H tower) is built for T <= 256 only (T=%d)", T);
  REED_CHECK_ARG(B > 0 && T > 0 && H > 0, "attention: bad dims B=%d T=%d H=%d", B, T, H);
  if (T <= 256) {
    const int lds = 4 * TILE_F, nitems = B * H;
    int ncu = num_cus();
    ncu -= ncu % 8;                       // whole XCD rounds: the item -> XCD map of xcd_contiguous
    const dim3 grid(nitems < ncu ? nitems : ncu);
#ifdef REED_ATTN_DIAG   // diagnosis build: REED_ATTN_FWD_DBG=<bits> runs the instantiation with parts of the work switched off
    static const int fdbg = getenv("REED_ATTN_FWD_DBG") ? atoi(getenv("REED_ATTN_FWD_DBG")) : 0;
#else
    constexpr int fdbg = 0;
#endif
    // T == 256, head_dim 64 / 72 (the training path): the forward with the V tile double-buffered (attn_fwd256v_kernel, round 6);
    // -DREED_ATTN_FWD_VDB_DEFAULT=0 builds keep attn_fwd256p_kernel there (A/B), which stays the kernel of T < 256 and head_dim 80
    if (REED_ATTN_FWD_VDB_DEFAULT && T == 256 && (hd == 64 || hd == 72) && !fdbg) {
      if (hd == 64) {
        static int once = set_lds(attn_fwd256v_kernel<64>, lds);
        if (once) return once;
        REED_KLAUNCH((attn_fwd256v_kernel<64>), grid, dim3(512), lds, (hipStream_t)stream, (const bf16*)qkv, (bf16*)o, lse, H, nitems);
      } else {
        static int once = set_lds(attn_fwd256v_kernel<72>, lds);
        if (once) return once;
        REED_KLAUNCH((attn_fwd256v_kernel<72>), grid, dim3(512), lds, (hipStream_t)stream, (const bf16*)qkv, (bf16*)o, lse, H, nitems);
      }
      REED_LAUNCH_CHECK();
      return REED_OK;
    }
#define LAUNCH_FWD256P(HD, FULL, DBGK)                                                                                 \
    do {                                                                                                               \
      static int once = set_lds(attn_fwd256p_kernel<HD, FULL, DBGK>, lds);                                             \
      if (once) return once;                                                                                           \
      REED_KLAUNCH((attn_fwd256p_kernel<HD, FULL, DBGK>), grid, dim3(512), lds, (hipStream_t)stream,                    \
                   (const bf16*)qkv, (bf16*)o, lse, T, H, nitems, fdbg);                                               \
    } while (0)
#ifdef REED_ATTN_DIAG
    if (fdbg && hd == 72 && T == 256) LAUNCH_FWD256P(72, true, true);
    else
#endif
    if (hd == 64) { if (T == 256) LAUNCH_FWD256P(64, true, false); else LAUNCH_FWD256P(64, false, false); }
    else if (hd == 72) { if (T == 256) LAUNCH_FWD256P(72, true, false); else LAUNCH_FWD256P(72, false, false); }
    else { if (T == 256) LAUNCH_FWD256P(80, true, false); else LAUNCH_FWD256P(80, false, false); }
#undef LAUNCH_FWD256P
    REED_LAUNCH_CHECK();
    return REED_OK;
  }
  const int lds = 2 * TILE_F;
  dim3 grid(B * H, (T + 255) / 256);
  // a last query block of at most 16 rows (T = 257, 261: the ViT towers) goes to the row kernel
  const int tail = T % 256;
  if (hd == 64 && T > 256 && T <= 512 && tail >= 1 && tail <= 16) {
    grid.y = T / 256;
    const int nwaves = B * H * tail;
    REED_KLAUNCH(attn_fwd_rows_kernel, dim3(cdiv(nwaves, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv, (bf16*)o, lse, T, H,
                 T - tail, tail, nwaves);
    REED_LAUNCH_CHECK();
  }
  if (hd == 64) {
    static int once = set_lds(attn_fwd_kernel<64>, lds);
    if (once) return once;
    REED_KLAUNCH(attn_fwd_kernel<64>, grid, dim3(512), lds, (hipStream_t)stream, (const bf16*)qkv, (bf16*)o, lse, B, T, H);
  } else {
    static int once = set_lds(attn_fwd_kernel<72>, lds);
    if (once) return once;
    REED_KLAUNCH(attn_fwd_kernel<72>, grid, dim3(512), lds, (hipStream_t)stream, (const bf16*)qkv, (bf16*)o, lse, B, T, H);
  }
  REED_LAUNCH_CHECK();
  return REED_OK;
}

extern "C" int reed_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse,
                                  void* dqkv, int B, int T, int H, int hd, void* stream) {
  REED_CHECK_ARG(qkv && o && d_o && lse && dqkv, "attention_bwd: null pointer");
  REED_CHECK_ARG(hd == 64 || hd == 72, "attention: head_dim %d unsupported (64 or 72)", hd);
  REED_CHECK_ARG(T > 0 && T <= 256, "attention_bwd: T=%d unsupported (training path is T <= 256)", T);
  const int lds = 4 * TILE_B + 256 + 2048;
  dim3 grid(B * H);
  // the one-shot key-stationary kernel (S and dP once); the engine's path is reed_attention_bwd_ws / _dp below.  (Round 1's
  // two-phase kernel — dQ by query rows, then dK / dV by key rows, S and dP twice — and a four-wave 64-row variant of it were
  // removed in round 5: round 3 measured 595 us isolated / 586 in-step at b = 256 for the two-phase kernel against 572 for this
  // one and 408 in-step for the persistent ring kernel, DESIGN_HISTORY.md.)
  if (hd == 64) {
    static int once = set_lds(attn_bwd_ks_kernel<64>, lds);
    if (once) return once;
    REED_KLAUNCH((attn_bwd_ks_kernel<64>), grid, dim3(512), lds, (hipStream_t)stream, (const bf16*)qkv, (const bf16*)o,
                 (const bf16*)d_o, lse, (bf16*)dqkv, B, T, H);
  } else {
    static int once = set_lds(attn_bwd_ks_kernel<72>, lds);
    if (once) return once;
    REED_KLAUNCH((attn_bwd_ks_kernel<72>), grid, dim3(512), lds, (hipStream_t)stream, (const bf16*)qkv, (const bf16*)o,
                 (const bf16*)d_o, lse, (bf16*)dqkv, B, T, H);
  }
  REED_LAUNCH_CHECK();
  return REED_OK;
}

// ------------------------------------------------------------------------------------------

// Backward with a workspace (ws: reed_attention_bwd_ws_floats(B, T, H) floats, caller-owned): delta = rowsum(dO * O) by a row
// kernel, then the persistent key-stationary kernel (attn_bwd_ring_kernel at T = 256, attn_bwd_ksp_kernel below it).
extern "C" int64_t reed_attention_bwd_ws_floats(int B, int T, int H) { return (int64_t)B * T * H; }

// dpart != NULL: delta comes from the partial dot products of reed_gemm's epilogue 13 (o is not read)
static int attention_bwd_persistent(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, float* ws,
                                    const float* dpart, int B, int T, int H, int hd, void* stream) {
  REED_CHECK_ARG(qkv && (o || dpart) && d_o && lse && dqkv && ws, "attention_bwd: null pointer");
  REED_CHECK_ARG(hd == 64 || hd == 72, "attention: head_dim %d unsupported (64 or 72)", hd);
  REED_CHECK_ARG(B > 0 && H > 0 && T > 0 && T <= 256, "attention_bwd: B=%d H=%d T=%d unsupported (training path is T <= 256)", B, H, T);
  const int lds = 4 * TILE_B + 256 + 2048;
  const int nitems = B * H;
  const long nseg = (long)B * T * H;
  int ncu = num_cus();
  ncu -= ncu % 8;                       // whole XCD rounds: the item -> XCD map of xcd_contiguous
  // Beside a collective (reed_set_concurrent_comm: the data-parallel backward) RCCL's channels hold CUs, and a grid of one workgroup
  // per CU with the items in a static stride then waits for the workgroups that found no CU to run their WHOLE lists after the
  // others (+ 42 % with 8-32 CUs held: profiles/r4_kernels_under_cu_hog.txt).  Several short lists per CU instead: the dispatcher
  // hands the next workgroup to whichever CU is free.
  const int gmult = reed_concurrent_comm() ? 4 : 1;
  const long gwant = (long)ncu * gmult;
  const dim3 pgrid((unsigned)(nitems < gwant ? nitems : gwant));
  hipStream_t s = (hipStream_t)stream;
#ifdef REED_ATTN_DIAG
  static const int dbg = getenv("REED_ATTN_KSP_DBG") ? atoi(getenv("REED_ATTN_KSP_DBG")) : 0;   // diagnosis build: skip parts of the work
#else
  constexpr int dbg = 0;
#endif
#define REED_DELTA(HD)                                                                                                    \
  do {                                                                                                                    \
    if (dpart)                                                                                                            \
      REED_KLAUNCH(attn_delta_combine_kernel, dim3(cdiv(nseg, 256)), dim3(256), 0, s, dpart, ws, (long)B * T, T, H, (HD) == 64 ? 1 : 2); \
    else                                                                                                                  \
      REED_KLAUNCH(attn_delta_kernel<HD>, dim3(cdiv(nseg, 256)), dim3(256), 0, s, (const bf16*)o, (const bf16*)d_o, ws, nseg, T, H); \
  } while (0)
#define REED_BWD_KSP(HD)                                                                                                  \
  do {                                                                                                                    \
    static int once = set_lds(attn_bwd_ksp_kernel<HD>, lds);                                                              \
    if (once) return once;                                                                                                \
    REED_DELTA(HD);                                                                                                       \
    REED_LAUNCH_CHECK();                                                                                                  \
    REED_KLAUNCH(attn_bwd_ksp_kernel<HD>, pgrid, dim3(512), lds, s, (const bf16*)qkv, (const bf16*)d_o, lse, (const float*)ws, \
                 (bf16*)dqkv, T, H, nitems, dbg);                                                                              \
  } while (0)
  if (T == 256) {
    const int rlds = 4 * 64 * ROWF + 2 * TILE_F + 256 * 128 + 4096;   // 156 KiB
#define REED_BWD_RING(HD)                                                                                                 \
  do {                                                                                                                    \
    static int once = set_lds(attn_bwd_ring_kernel<HD>, rlds);                                                            \
    if (once) return once;                                                                                                \
    REED_DELTA(HD);                                                                                                       \
    REED_LAUNCH_CHECK();                                                                                                  \
    REED_KLAUNCH(attn_bwd_ring_kernel<HD>, pgrid, dim3(512), rlds, s, (const bf16*)qkv, (const bf16*)d_o, lse, (const float*)ws, \
                 (bf16*)dqkv, H, nitems, dbg);                                                                            \
  } while (0)
#ifdef REED_ATTN_DIAG
    if (hd == 72 && (dbg & 4)) {   // the stamped instantiation (diagnosis)
      static int once = set_lds(attn_bwd_ring_kernel<72, true>, rlds);
      if (once) return once;
      REED_DELTA(72);
      REED_LAUNCH_CHECK();
      REED_KLAUNCH((attn_bwd_ring_kernel<72, true>), pgrid, dim3(512), rlds, s, (const bf16*)qkv, (const bf16*)d_o, lse, (const float*)ws,
                   (bf16*)dqkv, H, nitems, dbg);
    } else
#endif
    if (hd == 64) REED_BWD_RING(64);
    else REED_BWD_RING(72);
#undef REED_BWD_RING
    REED_LAUNCH_CHECK();
    return REED_OK;
  }
  if (hd == 64) REED_BWD_KSP(64);
  else REED_BWD_KSP(72);
#undef REED_BWD_KSP
  REED_LAUNCH_CHECK();
  return REED_OK;
}
#undef REED_DELTA

extern "C" int reed_attention_bwd_ws(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, float* ws,
                                     int B, int T, int H, int hd, void* stream) {
  if (!ws) return reed_attention_bwd(qkv, o, d_o, lse, dqkv, B, T, H, hd, stream);
  return attention_bwd_persistent(qkv, o, d_o, lse, dqkv, ws, nullptr, B, T, H, hd, stream);
}

extern "C" int reed_attention_bwd_dp(const void* qkv, const void* d_o, const float* lse, const float* dpart, void* dqkv, float* ws,
                                     int B, int T, int H, int hd, void* stream) {
  REED_CHECK_ARG(dpart != nullptr, "attention_bwd_dp: the partial dot products of reed_gemm epilogue 13 are required");
  return attention_bwd_persistent(qkv, nullptr, d_o, lse, dqkv, ws, dpart, B, T, H, hd, stream);
}
