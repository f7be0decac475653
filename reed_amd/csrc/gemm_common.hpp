// Shared pieces of the bf16 MFMA GEMM kernels (gemm.hip: 128x128 tile; gemm256.hip: 256x256 tile):
// buffer descriptors, LDS tile formats (swizzles), fragment reads and the fused epilogues.
#pragma once
#include "common.hpp"
#include "gemm.h"

namespace gemm_detail {

typedef void __attribute__((address_space(3))) * lds_ptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long bytes) {
  if (bytes < 0) bytes = 0;
  if (bytes > 0xFFFFFFFFl) bytes = 0xFFFFFFFFl;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (unsigned)bytes, 0x00020000);
}

typedef void __attribute__((address_space(3))) * lds_ptr_t;

__device__ __forceinline__ int tr_sw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
// ---- fragment reads -----------------------------------------------------------
// lane (i = lane&15, g = lane>>4) gets X[rowbase+i][ks*32 + 8g .. +7]
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int rowbase, int ks, int lane) {
  int row = rowbase + (lane & 15);
  int c = ks * 4 + (lane >> 4);
  return *(const bf16x8*)(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
}
// ds_read_b64_tr_b16 through inline asm.  With the builtin, hipcc (ROCm 7.2) puts an `s_waitcnt vmcnt(0)` between any
// pending LDS-DMA (buffer_load ... lds) and the transposing read, which drains the whole staging pipeline every
// K step (the plain ds_read_b128 path is not affected).  The asm form is invisible to that logic; the CALLER owns
// the ordering: an `s_waitcnt lgkmcnt(0)` + `__builtin_amdgcn_sched_barrier(0)` (REED_LDS_WAIT) before the first
// consumer, and the usual DMA-landed wait + barrier before the read.
#define REED_LDS_WAIT()                                    \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)
__device__ __forceinline__ bf16x4 ds_read_tr16(const char* p) {
  bf16x4 r;
  const unsigned a = (unsigned)(size_t)(const char __attribute__((address_space(3)))*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(a));
  return r;
}
template <int OFF>
__device__ __forceinline__ bf16x4 ds_read_tr16_off(const char* p) {
  bf16x4 r;
  const unsigned a = (unsigned)(size_t)(const char __attribute__((address_space(3)))*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(OFF));
  return r;
}
// lane (i, g) gets X[k = ks*32 + 8g + j][colbase + i], j = 0..7 (two transposing reads)
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int colbase, int ks, int lane) {
  int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
  int row0 = ks * 32 + 8 * g + q, row1 = row0 + 4;
  int ch = (colbase >> 3) + (p >> 1);
  const char* a0 = tile + row0 * 256 + ((ch ^ tr_sw(row0)) << 4) + ((p & 1) << 3);
  const char* a1 = tile + row1 * 256 + ((ch ^ tr_sw(row1)) << 4) + ((p & 1) << 3);
  bf16x4 lo = ds_read_tr16(a0);
  bf16x4 hi = ds_read_tr16(a1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- epilogue -------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ void epilogue(const GemmArgs& a, f32x4 acc, int m, int n, int z) {
  if (m >= a.M || n >= a.N) return;
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (a.bias) {
    bf16x4 b = *(const bf16x4*)(a.bias + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += bf2f(b[j]);
  }
  if constexpr (EPI == EPI_BF16) {
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
    *(bf16x4*)((bf16*)a.C + (long)m * a.ldc + n) = o;
  } else if constexpr (EPI == EPI_GELU || EPI == EPI_SILU) {
    bf16x4 pre, act;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pre[j] = f2bf(v[j]);
      float x = bf2f(pre[j]);
      act[j] = f2bf(EPI == EPI_GELU ? gelu_tanh_f(x) : silu_f(x));
    }
    if (a.C) *(bf16x4*)((bf16*)a.C + (long)m * a.ldc + n) = pre;
    *(bf16x4*)((bf16*)a.C2 + (long)m * a.ldc2 + n) = act;
  } else if constexpr (EPI == EPI_GATE_RES) {
    // y = bf16(acc+bias); x_out = x_in + float(bf16(gate*y))   (sit.py:134-135 under bf16 autocast)
    const bf16* gp = a.gate + (long)(m / a.rows_per_gate) * a.ldgate + n;
    bf16x4 g = *(const bf16x4*)gp;
    f32x4 xin = *(const f32x4*)((const float*)a.R + (long)m * a.ldr + n);
    bf16x4 y;
    f32x4 xo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      y[j] = f2bf(v[j]);
      xo[j] = xin[j] + bfround(bf2f(g[j]) * bf2f(y[j]));
    }
    if (a.C2) *(bf16x4*)((bf16*)a.C2 + (long)m * a.ldc2 + n) = y;
    *(f32x4*)((float*)a.C + (long)m * a.ldc + n) = xo;
  } else if constexpr (EPI == EPI_DGELU || EPI == EPI_DSILU) {
    bf16x4 pre = *(const bf16x4*)((const bf16*)a.R + (long)m * a.ldr + n);
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float du = bfround(v[j]);
      float x = bf2f(pre[j]);
      o[j] = f2bf(du * (EPI == EPI_DGELU ? gelu_tanh_grad_f(x) : silu_grad_f(x)));
    }
    *(bf16x4*)((bf16*)a.C + (long)m * a.ldc + n) = o;
  } else if constexpr (EPI == EPI_F32) {
    float* cp = (float*)a.C + (long)z * a.slab_stride + (long)m * a.ldc + n;
    f32x4 o = {v[0], v[1], v[2], v[3]};
    if (a.accumulate) {
      f32x4 old = *(const f32x4*)cp;
      o += old;
    }
    *(f32x4*)cp = o;
  } else if constexpr (EPI == EPI_ADDF32_RB) {
    float* cp = (float*)a.C + (long)m * a.ldc + n;
    f32x4 old = *(const f32x4*)cp;
#pragma unroll
    for (int j = 0; j < 4; ++j) old[j] += bfround(v[j]);
    *(f32x4*)cp = old;
  } else if constexpr (EPI == EPI_ATOMIC_F32) {
    float* cp = (float*)a.C + (long)m * a.ldc + n;
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(cp + j, v[j]);
  }
}

}  // namespace gemm_detail
