// Shared pieces of the bf16 MFMA GEMM kernels (gemm.hip: 128x128 tile; gemm256.hip: 256x256 tile):
// buffer descriptors, LDS tile formats (swizzles), fragment reads and the fused epilogues.
#pragma once
#include <type_traits>
#include "common.hpp"
#include "gemm.h"

#ifndef REED_WGRAD_ST_NT
#define REED_WGRAD_ST_NT 0
#endif
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
// One wave's NI x 4 grid of 16x16 accumulator tiles covers rows m0+mw .. +16 NI, columns nbase .. +64.  In the MFMA
// layout a lane owns 4 consecutive columns of one row, i.e. a wave-level store would write sixteen 32-byte pieces:
// partial cache lines that HBM takes badly (measured: the output stream cost as much as if nothing overlapped it).
// So every 16-row group goes through a wave-private 4 KiB LDS patch (XOR-swizzled, conflict-free both ways) and
// comes back row-contiguous: lane L owns 8 consecutive columns (L&7) of rows (L>>3) and (L>>3)+8, every global
// access is 16 bytes per lane and a wave-level access covers 8 full 128-byte (bf16) / 256-byte (fp32) row segments.
//
// The bf16-output epilogues (forward / dgrad paths) are also BRANCH-FREE: raw buffer loads/stores through
// descriptors based at the tile's first row; rows >= M fall outside num_records (dropped / read as 0 by the
// hardware range check), a wave whose 64 columns lie beyond N (N % 128 == 0) gets an out-of-range offset, and absent
// optional operands (bias, pre-activation, y) get an empty descriptor.  A wave issues exactly EpiOps<EPI, NI>::value
// vector-memory instructions whatever the tile's position (tools/isa_loops.py checks the count in the ISA).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int EPI, int NI>
struct EpiOps {
  static constexpr int value = EPI == EPI_BF16 ? 1 + NI * 2
                               : (EPI == EPI_GELU || EPI == EPI_SILU || EPI == EPI_DGELU || EPI == EPI_DSILU ||
                                  EPI == EPI_GELU_G || EPI == EPI_SILU_G || EPI == EPI_MUL ||
                                  EPI == EPI_QGELU || EPI == EPI_GELU_ERF || EPI == EPI_RES_BF16) ? 1 + NI * 4
                               : EPI == EPI_BF16_DOT ? 1 + NI * 4
                               : EPI == EPI_GATE_RES ? 1 + NI * 12
                               : EPI == EPI_LS_RES ? 3 + NI * 8
                                                     : -1;  // fp32-accumulate epilogues: pointer path, not counted
};

constexpr int EPI_STAGE_BYTES = 4096;  // per wave: 16 rows x 64 columns x fp32
constexpr int EPI_OOB = 0x7FFFFFF0;    // >= every num_records we build (epi_rsrc clamps to 0x7FFF0000)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const void* p, long bytes) {
  if (!p || bytes < 0) bytes = 0;
  if (bytes > 0x7FFF0000l) bytes = 0x7FFF0000l;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (unsigned)bytes, 0x00020000);
}
template <int AUX = 0>
__device__ __forceinline__ bf16x8 ld_bf16x8(__amdgpu_buffer_rsrc_t rs, int off, int soff = 0) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, AUX));
}
// REED_EPI_SAVED_NT (round 4): the arrays an epilogue writes for the BACKWARD pass only (fc1's pre-activation, the gate + residual
// epilogue's y) and the saved activations a backward epilogue reads once (dGELU's pre-activation, the head-dot epilogue's O) with
// the non-temporal policy: they do not displace what the next kernel reads back (the selective form of the store-policy experiment
// above, where the consumers paid for an all-or-nothing policy)
// 1: the GELU / dGELU epilogues on packed fp32 pairs (common.hpp: gelu_tanh2, gelu_tanh_grad2); 0: the scalar forms (A/B build)
#ifndef REED_EPI_PACKED
#define REED_EPI_PACKED 1
#endif
#ifndef REED_EPI_SAVED_NT
#define REED_EPI_SAVED_NT 0
#endif
constexpr int EPI_SAVED_AUX = REED_EPI_SAVED_NT ? 2 : 0;
// Cache policy of the epilogues' output stores (aux bits of the buffer instructions: 1 = sc0, 2 = nt, 16 = sc1).  Round 4, measured
// (profiles/r4_epilogue_store_policy.txt): with its stores switched off a plain-output GEMM loses 43 us of its 430 at b = 256 while
// its epilogues get only 0.4 us per tile shorter — the output stream costs by what it does to the K loops that follow (it
// allocates in the L2 the operand tiles are served from), not by its own issue.  Non-temporal stores, every GEMM alone at b = 256:
// fc1 + GELU (two 604 MB outputs) 0.682 -> 0.652 ms, dgrad proj + head dots 0.161 -> 0.150, qkv 0.462 -> 0.454, gate + residual
// unchanged or worse — but inside the step the consumers then miss (1266 -> 1270 images/s: + 0.3 %), and at b = 32, where an output
// fits the Infinity Cache, fc1 goes 0.108 -> 0.124 ms.  sc1 stores: + 14-37 %.  So the default policy stays;
// -DREED_EPI_ST_AUX=<bits> / -DREED_EPI_ST_NT_BIG builds repeat the measurement.
template <int EPI>
struct EpiStoreAux {
#if defined(REED_EPI_ST_AUX)
  static constexpr int value = REED_EPI_ST_AUX;
#elif defined(REED_EPI_ST_NT_BIG)
  static constexpr int value = (EPI == EPI_GATE_RES || EPI == EPI_LS_RES) ? 0 : 2;
#else
  static constexpr int value = 0;
#endif
};
#ifndef REED_EPI_LD_AUX
#define REED_EPI_LD_AUX 0
#endif
template <int AUX = 0>
__device__ __forceinline__ void st_bf16x8(bf16x8 v, __amdgpu_buffer_rsrc_t rs, int off) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, AUX);
}
__device__ __forceinline__ f32x4 ld_f32x4(__amdgpu_buffer_rsrc_t rs, int off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, REED_EPI_LD_AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void st_f32x4(f32x4 v, __amdgpu_buffer_rsrc_t rs, int off) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, AUX);
}

template <int EPI, int NI>
__device__ __forceinline__ void tile_epilogue_ptr(const GemmArgs& a, const f32x4 (&acc)[NI][4], int mbase, int nbase,
                                                  int lane, int z);

// stage: this wave's EPI_STAGE_BYTES of LDS (no other wave touches it; the caller made sure the K loop is done with it)
// PF > 0: the operand loads (residual stream / pre-activation) of PF row groups are kept in flight ahead of the one being
// written — for the 4-wave kernel (gemm256w.hip), whose CU has half as many epilogue streams to hide a load round trip
// behind (on the 8-wave kernels it measured a wash: DESIGN.md).  PF = 0: loads of group i right before its transpose.
template <int EPI, int NI, int PF = 0>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& a, const f32x4 (&acc)[NI][4], int m0, int mw, int nbase,
                                              int lane, int z, char* stage) {
  if constexpr (EpiOps<EPI, NI>::value < 0) {
    tile_epilogue_ptr<EPI, NI>(a, acc, m0 + mw, nbase, lane, z);
  } else {
    constexpr int SA = EpiStoreAux<EPI>::value;
    // MFMA-side coordinates (LDS write) and row-contiguous coordinates (LDS read, global access)
    const int wr_row = lane & 15, wr_g = lane >> 4;
    const int rd_row = lane >> 3, rd_c = lane & 7;       // rows rd_row, rd_row + 8; columns 8 rd_c .. +7
    char* wr_base = stage + wr_row * 256;
    const char* rd_base0 = stage + rd_row * 256;
    const char* rd_base1 = stage + (rd_row + 8) * 256;
    const int rd_sw0 = ((2 * rd_c) ^ rd_row) << 4, rd_sw1 = ((2 * rd_c) ^ (rd_row + 8)) << 4;  // chunk c0; c0+1 = ^16

    const bool cv = nbase < a.N;            // wave-uniform: a wave spans 64 columns, N is a multiple of 16
    const long rows = a.M - m0;             // rows of C from the tile's first row on (> 0)
    const int col = nbase + 8 * rd_c;       // lane's first column
    const int rt = mw + rd_row;             // lane's first row inside the tile (i = 0, h = 0)
    float bs[8];
    {
      const __amdgpu_buffer_rsrc_t rsB = epi_rsrc(a.bias, (long)a.N * 2);   // empty if absent -> zeros
      bf16x8 b = ld_bf16x8(rsB, cv ? col * 2 : EPI_OOB);
#pragma unroll
      for (int e = 0; e < 8; ++e) bs[e] = bf2f(b[e]);
    }
    auto tile_rsrc = [&](const void* p, long ld, int es) {
      return epi_rsrc(p ? (const char*)p + (long)m0 * ld * es : nullptr, ((rows - 1) * ld + a.N) * es);
    };
    auto lane_off = [&](long ld, int es) { return cv ? (int)((rt * ld + col) * es) : EPI_OOB; };
    // acc group i -> LDS -> v[h][0..7] = (acc + bias) of row 16 i + rd_row + 8 h, columns col .. col + 7
    auto transpose = [&](int i, float (&v)[2][8]) {
      // one row group at a time: without the scheduling fence the compiler interleaves several groups' LDS round
      // trips, runs out of registers and spills a K-loop value, whose reload costs an s_waitcnt vmcnt(0) per tile
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      f32x4 q[2][2];
#ifdef REED_EPI_DIAG_NOLDS   // diagnosis build: no LDS round trip (values of the wrong elements, the same arithmetic and stores)
      q[0][0] = acc[i][0]; q[0][1] = acc[i][1]; q[1][0] = acc[i][2]; q[1][1] = acc[i][3];
#else
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(wr_base + (((4 * j + wr_g) ^ wr_row) << 4)) = acc[i][j];
      asm volatile("" ::: "memory");   // same wave, LDS executes in order: no barrier, just no compiler reordering
      q[0][0] = *(const f32x4*)(rd_base0 + rd_sw0);
      q[0][1] = *(const f32x4*)(rd_base0 + (rd_sw0 ^ 16));
      q[1][0] = *(const f32x4*)(rd_base1 + rd_sw1);
      q[1][1] = *(const f32x4*)(rd_base1 + (rd_sw1 ^ 16));
      asm volatile("" ::: "memory");
#endif
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[h][e] = q[h][e >> 2][e & 3] + bs[e];
    };

    if constexpr (EPI == EPI_BF16_DOT) {
      // the plain store plus, per row and head, the dot product of the stored values with R (same layout): a lane owns 8
      // columns of one head (hd is a multiple of 8), the 8 lanes of a row cover this wave's 64-column strip, which holds the
      // tail of head hA and possibly the start of head hA + 1; both sums go out from the row's first lane (fixed order: two
      // quad butterflies + the upper quad, as DPP adds).  dpart f32 [N / hd, S, M]: a wave-level store is 8 consecutive rows.
      const __amdgpu_buffer_rsrc_t rsC = tile_rsrc(a.C, a.ldc, 2), rsR = tile_rsrc(a.R, a.ldr, 2);
      int oc = lane_off(a.ldc, 2);
      const int orr = lane_off(a.ldr, 2);
      const int s8 = (int)(8 * a.ldc * 2), r8 = (int)(8 * a.ldr * 2);
      const int hd = a.rows_per_gate, S = hd == 64 ? 1 : 2;
      const int hA = nbase / hd;
      const bool inA = col / hd == hA, two = (nbase + 63) / hd != hA;
      const int slotA = nbase / 64 - (hA * hd) / 64;
      float* dpA = (float*)a.C2 + (long)(hA * S + slotA) * a.M;
      float* dpB = (float*)a.C2 + (long)(hA + 1) * S * a.M;
      bf16x8 rr[NI][2];
      auto fetch = [&](int i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) rr[i][h] = ld_bf16x8<EPI_SAVED_AUX>(rsR, orr + (2 * i + h) * r8);
      };
      auto sum8 = [](float x) {     // lanes 8 k .. 8 k + 7 -> their sum in lane 8 k
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x104, 0xF, 0xF, true));  // row_shl:4
        return x;
      };
#pragma unroll
      for (int i = 0; i < PF && i < NI; ++i) fetch(i);
#pragma unroll
      for (int i = 0; i < NI; ++i, oc += 2 * s8) {
        if constexpr (PF == 0) fetch(i);
        float v[2][8];
        transpose(i, v);
        if (PF > 0 && i + PF < NI) fetch(i + PF);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2bf(v[h][e]);
          st_bf16x8<SA>(o, rsC, oc + h * s8);
          float d = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) d = fmaf(bf2f(o[e]), bf2f(rr[i][h][e]), d);
          const float dA = sum8(inA ? d : 0.f), dB = sum8(inA ? 0.f : d);
          const long row = (long)m0 + rt + 16 * i + 8 * h;
          if (rd_c == 0 && cv && row < a.M) {
            dpA[row] = dA;
            if (two) dpB[row] = dB;
          }
        }
      }
    } else if constexpr (EPI == EPI_BF16) {
      const __amdgpu_buffer_rsrc_t rsC = tile_rsrc(a.C, a.ldc, 2);
      int oc = lane_off(a.ldc, 2);
      const int s8 = (int)(8 * a.ldc * 2);
#pragma unroll
      for (int i = 0; i < NI; ++i, oc += 2 * s8) {
        float v[2][8];
        transpose(i, v);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = f2bf(v[h][e]);
#ifdef REED_EPI_DIAG_NOSTORE
          st_bf16x8<SA>(o, rsC, bf2f(o[0]) == 12345.678f ? 0 : EPI_OOB);
#else
          st_bf16x8<SA>(o, rsC, oc + h * s8);
#endif
        }
      }
    } else if constexpr (EPI == EPI_GELU || EPI == EPI_SILU || EPI == EPI_QGELU || EPI == EPI_GELU_ERF || EPI == EPI_GELU_G ||
                         EPI == EPI_SILU_G) {
      const __amdgpu_buffer_rsrc_t rsC = tile_rsrc(a.C, a.ldc, 2), rsC2 = tile_rsrc(a.C2, a.ldc2, 2);
      int oc = lane_off(a.ldc, 2), oc2 = lane_off(a.ldc2, 2);
      const int s8 = (int)(8 * a.ldc * 2), t8 = (int)(8 * a.ldc2 * 2);
#pragma unroll
      for (int i = 0; i < NI; ++i, oc += 2 * s8, oc2 += 2 * t8) {
        float v[2][8];
        transpose(i, v);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x8 pre, act;
          if constexpr (EPI == EPI_GELU_G) {   // `pre` carries gelu'(pre): what the backward multiplies by (EPI_MUL)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const bf16x2 p2 = __builtin_convertvector(f32x2{v[h][2 * k], v[h][2 * k + 1]}, bf16x2);
              f32x2 av, gv;
              gelu_tanh_both2(__builtin_convertvector(p2, f32x2), av, gv);
              const bf16x2 a2 = __builtin_convertvector(av, bf16x2), g2 = __builtin_convertvector(gv, bf16x2);
              pre[2 * k] = g2[0]; pre[2 * k + 1] = g2[1];
              act[2 * k] = a2[0]; act[2 * k + 1] = a2[1];
            }
          } else if constexpr (EPI == EPI_SILU_G) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float av, gv;
              silu_both(bfround(v[h][e]), av, gv);
              pre[e] = f2bf(gv);
              act[e] = f2bf(av);
            }
          } else if constexpr (EPI == EPI_GELU && REED_EPI_PACKED) {   // two elements per instruction (common.hpp: gelu_tanh2)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const bf16x2 p2 = __builtin_convertvector(f32x2{v[h][2 * k], v[h][2 * k + 1]}, bf16x2);
              const bf16x2 a2 = __builtin_convertvector(gelu_tanh2(__builtin_convertvector(p2, f32x2)), bf16x2);
              pre[2 * k] = p2[0]; pre[2 * k + 1] = p2[1];
              act[2 * k] = a2[0]; act[2 * k + 1] = a2[1];
            }
          } else
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            pre[e] = f2bf(v[h][e]);
            float x = bf2f(pre[e]);
            // QuickGELU = x * sigmoid(1.702 x) with eager-mode bf16 roundings (clip_vit.py:168-170); exact GELU (nn.GELU: the timm /
            // I-JEPA towers) is its OWN instantiation since round 4: as a run-time variant of one epilogue both activations
            // were compiled into every element (11 k instructions, 288 branches: 0.64 ms for ViT-L's fc1 where the tanh-GELU
            // form of the same shape takes a quarter of that)
            if constexpr (EPI == EPI_QGELU) act[e] = f2bf(x * bfround(sigmoid_f(bfround(1.702f * x))));
            else if constexpr (EPI == EPI_GELU_ERF) act[e] = f2bf(gelu_erf_f(x));
            else
              act[e] = f2bf(EPI == EPI_GELU ? gelu_tanh_f(x) : silu_f(x));
          }
          st_bf16x8<SA | EPI_SAVED_AUX>(pre, rsC, oc + h * s8);    // empty descriptor when the pre-activation is not wanted
          st_bf16x8<SA>(act, rsC2, oc2 + h * t8);
        }
      }
    } else if constexpr (EPI == EPI_GATE_RES) {
      // y = bf16(acc+bias); x_out = x_in + float(bf16(gate*y))   (sit.py:134-135 under bf16 autocast)
      const __amdgpu_buffer_rsrc_t rsC = tile_rsrc(a.C, a.ldc, 4), rsR = tile_rsrc(a.R, a.ldr, 4),
                                   rsY = tile_rsrc(a.C2, a.ldc2, 2);
      const long grows = (a.M + a.rows_per_gate - 1) / a.rows_per_gate;
      const __amdgpu_buffer_rsrc_t rsG = epi_rsrc(a.gate, ((grows - 1) * a.ldgate + a.N) * 2);
      const int orr = lane_off(a.ldr, 4);
      const int s8 = (int)(8 * a.ldc * 4), r8 = (int)(8 * a.ldr * 4), y8 = (int)(8 * a.ldc2 * 2);
      // gate row of a 16-row group: scalar walk when groups cannot straddle gate rows, per-lane division otherwise
      const bool gfast = (a.rows_per_gate & 15) == 0;
      const int grow0 = (m0 + mw) / a.rows_per_gate;
      const int grem0 = (m0 + mw) - grow0 * a.rows_per_gate;
      const int og = cv ? col * 2 : EPI_OOB;
      // Round 4: where every row group of the strip lies in ONE gate row (SiT: rows_per_gate = T = 256 = the tile height, always)
      // the gate vector is loaded once instead of once per (group, half) — a third of the epilogue's operand loads — and the 8
      // registers per group that frees carry the residual stream of twice as many groups in flight (PF 4 -> 8: the whole strip).
#ifdef REED_EPI_NOHOIST   // A/B build: round 3's form
      const bool ghoist = false;
#else
      const bool ghoist = gfast && grem0 + 16 * NI <= a.rows_per_gate;
#endif
      auto body = [&](auto hoist_c, auto pf_c) {
        constexpr bool HOIST = decltype(hoist_c)::value;
        constexpr int PFX = decltype(pf_c)::value;
        int grow = grow0, grem = grem0;
        int oc = lane_off(a.ldc, 4), oy = lane_off(a.ldc2, 2);
        bf16x8 gone = bf16x8{};
        if constexpr (HOIST) gone = ld_bf16x8(rsG, og, grow0 * (int)a.ldgate * 2);
        bf16x8 g[HOIST ? 1 : NI][2];
        f32x4 xin[NI][2][2];
        auto fetch = [&](int i) {   // operand loads of row group i (they do not depend on the accumulators)
          int gso = 0;
          if (!HOIST && gfast) {
            gso = grow * (int)a.ldgate * 2;
            grem += 16;
            if (grem >= a.rows_per_gate) { grem -= a.rows_per_gate; ++grow; }
          }
          const int ro = orr + i * 2 * r8;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#ifdef REED_EPI_DIAG_NOLOAD   // diagnosis build (tools/_ab/build_variant.py): the epilogue without its operand loads
            if constexpr (!HOIST) g[i][h] = bf16x8{};
            xin[i][h][0] = xin[i][h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            (void)gso; (void)ro;
#else
            if constexpr (!HOIST) {
              int gvo = og;
              if (!gfast) gvo = cv ? (int)(((long)((m0 + rt + 16 * i + 8 * h) / a.rows_per_gate) * a.ldgate + col) * 2) : EPI_OOB;
              g[i][h] = ld_bf16x8(rsG, gvo, gso);
            }
            xin[i][h][0] = ld_f32x4(rsR, ro + h * r8);
            xin[i][h][1] = ld_f32x4(rsR, ro + h * r8 + 16);
#endif
          }
        };
#pragma unroll
        for (int i = 0; i < PFX && i < NI; ++i) fetch(i);
#pragma unroll
        for (int i = 0; i < NI; ++i, oc += 2 * s8, oy += 2 * y8) {
          if constexpr (PFX == 0) fetch(i);
          float v[2][8];
          transpose(i, v);
          if (PFX > 0 && i + PFX < NI) fetch(i + PFX);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            bf16x8 y;
            f32x4 xo[2];
            const bf16x8 gv = HOIST ? gone : g[HOIST ? 0 : i][h];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              y[e] = f2bf(v[h][e]);
              xo[e >> 2][e & 3] = xin[i][h][e >> 2][e & 3] + bfround(bf2f(gv[e]) * bf2f(y[e]));
            }
#ifdef REED_EPI_DIAG_NOSTORE   // diagnosis build: the stores dropped by the range check (the values stay live through the offset)
            const int dro = (xo[0][0] == 12345.678f && bf2f(y[0]) == 3.f) ? 0 : EPI_OOB;
            st_bf16x8<SA>(y, rsY, dro);
            st_f32x4<SA>(xo[0], rsC, dro);
            st_f32x4<SA>(xo[1], rsC, dro);
#else
            st_bf16x8<SA | EPI_SAVED_AUX>(y, rsY, oy + h * y8);      // empty descriptor when y is not wanted
            st_f32x4<SA>(xo[0], rsC, oc + h * s8);
            st_f32x4<SA>(xo[1], rsC, oc + h * s8 + 16);
#endif
          }
        }
      };
#ifndef REED_EPI_PFH_NUM     // depth of the hoisted form = PF * NUM / 2 (A/B builds: 3 -> 6 groups, default 4 -> 8 = the whole strip)
#define REED_EPI_PFH_NUM 4
#endif
      if (ghoist) body(std::true_type{}, std::integral_constant<int, PF == 0 ? 0 : (PF * REED_EPI_PFH_NUM / 2 < NI ? PF * REED_EPI_PFH_NUM / 2 : NI)>{});
      else body(std::false_type{}, std::integral_constant<int, PF>{});
    } else if constexpr (EPI == EPI_LS_RES) {
      // x_out = x_in + gamma * float(bf16(acc+bias)): LayerScale (an fp32 parameter times the bf16 linear output promotes
      // to fp32 under autocast) and the fp32 residual add of a DINOv2 block
      const __amdgpu_buffer_rsrc_t rsC = tile_rsrc(a.C, a.ldc, 4), rsR = tile_rsrc(a.R, a.ldr, 4);
      const __amdgpu_buffer_rsrc_t rsG = epi_rsrc(a.gate, (long)a.N * 4);
      int oc = lane_off(a.ldc, 4), orr = lane_off(a.ldr, 4);
      const int s8 = (int)(8 * a.ldc * 4), r8 = (int)(8 * a.ldr * 4);
      const f32x4 gm0 = ld_f32x4(rsG, cv ? col * 4 : EPI_OOB), gm1 = ld_f32x4(rsG, cv ? col * 4 + 16 : EPI_OOB);
#pragma unroll
      for (int i = 0; i < NI; ++i, oc += 2 * s8, orr += 2 * r8) {
        f32x4 xin[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          xin[h][0] = ld_f32x4(rsR, orr + h * r8);
          xin[h][1] = ld_f32x4(rsR, orr + h * r8 + 16);
        }
        float v[2][8];
        transpose(i, v);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 xo[2];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            xo[e >> 2][e & 3] = xin[h][e >> 2][e & 3] + (e < 4 ? gm0[e & 3] : gm1[e & 3]) * bfround(v[h][e]);
          st_f32x4<SA>(xo[0], rsC, oc + h * s8);
          st_f32x4<SA>(xo[1], rsC, oc + h * s8 + 16);
        }
      }
    } else {  // EPI_DGELU / EPI_DSILU / EPI_MUL / EPI_RES_BF16
      const __amdgpu_buffer_rsrc_t rsC = tile_rsrc(a.C, a.ldc, 2), rsR = tile_rsrc(a.R, a.ldr, 2);
      int oc = lane_off(a.ldc, 2), orr = lane_off(a.ldr, 2);
      const int s8 = (int)(8 * a.ldc * 2), r8 = (int)(8 * a.ldr * 2);
      bf16x8 pre[NI][2];
      auto fetch = [&](int i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) pre[i][h] = ld_bf16x8<(EPI == EPI_DGELU || EPI == EPI_DSILU || EPI == EPI_MUL) ? EPI_SAVED_AUX : 0>(rsR, orr + i * 2 * r8 + h * r8);
      };
      constexpr int PFD = 2 * PF;   // 8 registers per row group: twice the depth of the fp32 residual's
#pragma unroll
      for (int i = 0; i < PFD && i < NI; ++i) fetch(i);
#pragma unroll
      for (int i = 0; i < NI; ++i, oc += 2 * s8) {
        if constexpr (PF == 0) fetch(i);
        float v[2][8];
        transpose(i, v);
        if (PF > 0 && i + PFD < NI) fetch(i + PFD);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bf16x8 o;
          if constexpr (EPI == EPI_MUL) {   // the saved factor IS the activation's derivative (EPI_GELU_G / EPI_SILU_G)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const f32x2 du = __builtin_convertvector(__builtin_convertvector(f32x2{v[h][2 * k], v[h][2 * k + 1]}, bf16x2), f32x2);
              const f32x2 x = __builtin_convertvector(bf16x2{pre[i][h][2 * k], pre[i][h][2 * k + 1]}, f32x2);
              const bf16x2 o2 = __builtin_convertvector(du * x, bf16x2);
              o[2 * k] = o2[0]; o[2 * k + 1] = o2[1];
            }
          } else if constexpr (EPI == EPI_DGELU && REED_EPI_PACKED) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const f32x2 du = __builtin_convertvector(__builtin_convertvector(f32x2{v[h][2 * k], v[h][2 * k + 1]}, bf16x2), f32x2);
              const f32x2 x = __builtin_convertvector(bf16x2{pre[i][h][2 * k], pre[i][h][2 * k + 1]}, f32x2);
              const bf16x2 o2 = __builtin_convertvector(du * gelu_tanh_grad2(x), bf16x2);
              o[2 * k] = o2[0]; o[2 * k + 1] = o2[1];
            }
          } else
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float du = bfround(v[h][e]);
            float x = bf2f(pre[i][h][e]);
            if constexpr (EPI == EPI_RES_BF16) o[e] = f2bf(du + x);   // bf16 residual stream (frozen encoder)
            else o[e] = f2bf(du * (EPI == EPI_DGELU ? gelu_tanh_grad_f(x) : silu_grad_f(x)));
          }
          st_bf16x8<SA>(o, rsC, oc + h * s8);
        }
      }
    }
  }
}

// fp32-output epilogues (weight gradients, accumulating variants): plain pointers, guarded per row / column group.
template <int EPI, int NI>
__device__ __forceinline__ void tile_epilogue_ptr(const GemmArgs& a, const f32x4 (&acc)[NI][4], int mbase, int nbase,
                                                  int lane, int z) {
  const int lr = lane & 15, lc = 4 * (lane >> 4);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = mbase + 16 * i + lr;
    if (m >= a.M) continue;
    const long rc = (long)m * a.ldc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nbase + 16 * j + lc;
      if (n >= a.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (a.bias) {
        bf16x4 b = *(const bf16x4*)(a.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bf2f(b[e]);
      }
      if constexpr (EPI == EPI_F32) {
        float* cp = (float*)a.C + (long)z * a.slab_stride + rc + n;
        f32x4 o = {v[0], v[1], v[2], v[3]};
        if (a.accumulate) {
          f32x4 old = *(const f32x4*)cp;
          o += old;
        }
#if REED_WGRAD_ST_NT   // weight gradients: 85 MB per block written once, read by the norm / optimiser pass at the end of the step
        __builtin_nontemporal_store(o, (f32x4*)cp);
#else
        *(f32x4*)cp = o;
#endif
      } else if constexpr (EPI == EPI_ADDF32_RB) {
        float* cp = (float*)a.C + rc + n;
        f32x4 old = *(const f32x4*)cp;
#pragma unroll
        for (int e = 0; e < 4; ++e) old[e] += bfround(v[e]);
        *(f32x4*)cp = old;
      } else {  // EPI_ATOMIC_F32
        float* cp = (float*)a.C + rc + n;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(cp + e, v[e]);
      }
    }
  }
}

// The 16-column strip (columns 64..79 of a tile, held by the wave-column-0 waves as one extra MFMA tile per 16 rows)
// straight from the MFMA layout: lane (r = lane & 15, g = lane >> 4) owns row 16 i + r, columns 4 g .. 4 g + 3, i.e.
// 8-byte bf16 / 16-byte fp32 accesses, four lanes per 32 / 64 contiguous bytes of a row.  It is a ninth of the
// output; sending it through tile_epilogue's 64-column LDS round trip (lane-masked) doubled those waves' epilogue.
// Same arithmetic and rounding points as tile_epilogue.
template <int EPI>
__device__ __forceinline__ void strip_epilogue(const GemmArgs& a, const f32x4 (&acc)[4], int mbase, int nbase, int lane) {
  const int n = nbase + 4 * (lane >> 4);
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
    const bf16x4 b = *(const bf16x4*)(a.bias + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) bs[e] = bf2f(b[e]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mbase + 16 * i + (lane & 15);
    if (m >= a.M) continue;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[i][e] + bs[e];
    if constexpr (EPI == EPI_BF16) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
      *(bf16x4*)((bf16*)a.C + (long)m * a.ldc + n) = o;
    } else if constexpr (EPI == EPI_GELU || EPI == EPI_SILU || EPI == EPI_QGELU || EPI == EPI_GELU_ERF || EPI == EPI_GELU_G ||
                         EPI == EPI_SILU_G) {
      bf16x4 pre, act;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pre[e] = f2bf(v[e]);
        const float x = bf2f(pre[e]);
        if constexpr (EPI == EPI_GELU_G || EPI == EPI_SILU_G) {   // the saved array carries the derivative (gemm.h)
          float av, gv;
          if constexpr (EPI == EPI_GELU_G) gelu_tanh_both(x, av, gv);
          else silu_both(x, av, gv);
          pre[e] = f2bf(gv);
          act[e] = f2bf(av);
        } else if constexpr (EPI == EPI_QGELU) act[e] = f2bf(x * bfround(sigmoid_f(bfround(1.702f * x))));
        else if constexpr (EPI == EPI_GELU_ERF) act[e] = f2bf(gelu_erf_f(x));
        else act[e] = f2bf(EPI == EPI_GELU ? gelu_tanh_f(x) : silu_f(x));
      }
      if (a.C) *(bf16x4*)((bf16*)a.C + (long)m * a.ldc + n) = pre;
      *(bf16x4*)((bf16*)a.C2 + (long)m * a.ldc2 + n) = act;
    } else if constexpr (EPI == EPI_GATE_RES) {
      const bf16x4 g = *(const bf16x4*)(a.gate + (long)(m / a.rows_per_gate) * a.ldgate + n);
      const f32x4 xin = *(const f32x4*)((const float*)a.R + (long)m * a.ldr + n);
      bf16x4 y;
      f32x4 xo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = f2bf(v[e]);
        xo[e] = xin[e] + bfround(bf2f(g[e]) * bf2f(y[e]));
      }
      if (a.C2) *(bf16x4*)((bf16*)a.C2 + (long)m * a.ldc2 + n) = y;
      *(f32x4*)((float*)a.C + (long)m * a.ldc + n) = xo;
    } else {  // EPI_DGELU / EPI_DSILU / EPI_RES_BF16
      const bf16x4 pre = *(const bf16x4*)((const bf16*)a.R + (long)m * a.ldr + n);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float du = bfround(v[e]);
        const float x = bf2f(pre[e]);
        if constexpr (EPI == EPI_RES_BF16) o[e] = f2bf(du + x);
        else if constexpr (EPI == EPI_MUL) o[e] = f2bf(du * x);
        else o[e] = f2bf(du * (EPI == EPI_DGELU ? gelu_tanh_grad_f(x) : silu_grad_f(x)));
      }
      *(bf16x4*)((bf16*)a.C + (long)m * a.ldc + n) = o;
    }
  }
}

}  // namespace gemm_detail
