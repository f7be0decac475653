// 256x144x64 bf16 MFMA GEMM for gfx950 — the tile that QUANTISES on SiT-XL/2 (reference: every nn.Linear forward and
// its input gradient, image/models/sit.py:17-24,114-129; timm Attention / Mlp).
//
// Why another tile: every output width of SiT-XL/2 is a multiple of 144 (1152 = 8 x 144, 3456 = 24 x 144, 4608 =
// 32 x 144) and the token count is b x 256, so a 256x144 tile cuts every forward / dgrad GEMM into exactly
// 8 b, 24 b or 32 b equal workgroups: at b = 32 per GPU (the 8-GPU strong-scaling point) that is 1, 3 and 4 full rounds
// of the 256 CUs, where the 256^2 kernel runs the 1152-wide outputs as 160 workgroups on 256 CUs (one round at 62 %
// occupancy, a fifth of the tiles half empty) and the 128^2 kernel as 576 tiles on 512 slots (two rounds, the second
// 12 % full).
//
// 8 waves = 4 (rows) x 2 (columns): wave (wr, wc) owns rows 64 wr .. +63 and columns 0..79 (wc = 0: 4 x 5 MFMA tiles)
// or 80..143 (wc = 1: 4 x 4).  Waves w and w + 4 share a SIMD, and wc = w >> 2, so every SIMD carries one 5-column and
// one 4-column wave: 72 MFMAs per SIMD and K-tile, balanced.  Operand tiles use gemm_common.hpp's LDS formats:
//   A (k-contiguous, NT and NN)  [256][64]  row format, 32 KiB
//   B NT (k-contiguous)          [144][64]  row format, 18 KiB (rows 128..143 = the 2 KiB "piece", staged by waves 0, 1)
//   B NN (k-strided)             [64][128]  transposing-read format, 16 KiB + a [64][16] piece of 2 KiB whose 32-byte
//                                rows sit at slot(r) = r with bits 2 and 3 swapped (the 8 rows one half-wave of a
//                                ds_read_b64_tr_b16 touches are then 256 contiguous bytes: conflict-free)
// Pipeline: a 3-slot ring of 50 KiB stages fed by LDS-DMA two K-tiles ahead, ONE barrier per K-tile placed between the
// two k-halves; every LDS read is inline asm with manual waits, and the fragment reads of the next k-half are issued
// before this half's MFMAs:
//   half 0:  wait reads(t, ks0) | issue reads(t, ks1) | MFMA(t, ks0)
//   half 1:  wait DMA(t+1), reads(t, ks1) | barrier | issue DMA(t+3) into t's slot | issue reads(t+1, ks0) | MFMA(t, ks1)
// NT / NN layouts, bf16-output epilogues (tile_epilogue's LDS-staged row-contiguous stores), no split-K.
#include <math.h>
#include <stdlib.h>

#include "gemm_common.hpp"

namespace {
using namespace gemm_detail;

constexpr int BM4 = 256, BN4 = 144, BK4 = 64;
constexpr int A_BYTES = 32768, B_MAIN = 16384, B_PIECE = 2048;
constexpr int STAGE4 = A_BYTES + B_MAIN + B_PIECE;   // 51200
constexpr int LDS4 = 3 * STAGE4;                     // 150 KiB

__device__ __forceinline__ int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

#define RD128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define RDTR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define TR2(dst, a0, a1, OFF)                                \
  do {                                                       \
    u32x2 lo_, hi_;                                          \
    RDTR(lo_, a0, OFF);                                      \
    RDTR(hi_, a1, OFF);                                      \
    dst = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3);     \
  } while (0)
#define LDS_WAIT0()                                        \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

// (strip_epilogue — the 16-column strip straight from the MFMA layout — lives in gemm_common.hpp: gemm288.hip uses it too)

// LW (the form launched; the 8-wave form without them is the template's other instantiation): 4 extra LOADER waves (8..11, one per SIMD) issue every LDS-DMA of the workgroup;
// the 8 compute waves then carry MFMAs and fragment reads only (no buffer_load ... lds issue cycles, no vmcnt waits).
template <int LAY, int EPI, bool LW>
__global__ __launch_bounds__(LW ? 768 : 512) void gemm144_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 3, wc = wave >> 2;
  const bool extra = wave < 2;   // waves 0 and 1 also stage the 2 KiB B piece (one more DMA per K-tile)

  // ---- block -> tile (XCD-aware, grouped along M; as gemm256.hip) ----
  const int ntm = (a.M + BM4 - 1) / BM4, ntn = a.N / BN4;
  int tm, tn;
  {
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
      int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    constexpr int GM = 4;
    const int per_group = GM * ntn;
    const int group = bid / per_group, first_m = group * GM;
    const int gs = min(ntm - first_m, GM);
    tm = first_m + (bid % per_group) % gs;
    tn = (bid % per_group) / gs;
  }
  const int m0 = tm * BM4, n0 = tn * BN4;
  const int nt = a.K / BK4;

  const __amdgpu_buffer_rsrc_t rsP = make_rsrc(a.P + (long)m0 * a.ldp, ((long)(a.M - m0) * a.ldp) * 2);
  __amdgpu_buffer_rsrc_t rsQ;
  if constexpr (LAY == LAY_NT) rsQ = make_rsrc(a.Q + (long)n0 * a.ldq, ((long)(a.N - n0) * a.ldq) * 2);
  else rsQ = make_rsrc(a.Q + n0, ((long)a.K * a.ldq - n0) * 2);

  // ---- staging: loop-invariant per-thread offsets, everything per K-tile / round goes through the scalar offset ----
  const int r8 = tid >> 3, cp8 = tid & 7;
  const int vA = (int)(((long)r8 * a.ldp + ((cp8 ^ ((r8 >> 1) & 7)) << 3)) * 2);
  const int rndA = (int)(64 * a.ldp * 2);
  int vB, vBp, rndB, kstepB, pieceB;
  if constexpr (LAY == LAY_NT) {
    vB = (int)(((long)r8 * a.ldq + ((cp8 ^ ((r8 >> 1) & 7)) << 3)) * 2);
    vBp = vB;                           // rows 128 .. 143 = "round 2" of the same pattern, threads 0 .. 127
    rndB = (int)(64 * a.ldq * 2);
    kstepB = BK4 * 2;
    pieceB = 2 * rndB;
  } else {
    const int r16 = tid >> 4, chp = tid & 15;
    vB = (int)(((long)r16 * a.ldq + ((chp ^ tr_sw(r16)) << 3)) * 2);
    const int d = tid & 127, rr = swap23(d >> 1);
    vBp = (int)(((long)rr * a.ldq + 128 + (d & 1) * 8) * 2);
    rndB = (int)(32 * a.ldq * 2);
    kstepB = (int)(BK4 * a.ldq * 2);
    pieceB = 0;
  }
  auto issue = [&](int t, int slot) {
    char* sb = smem + slot * STAGE4;
    const int kA = t * (BK4 * 2), kB = t * kstepB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_ptr_t)(sb + (i * 512 + wave * 64) * 16), 16, vA, kA + i * rndA, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(sb + A_BYTES + (i * 512 + wave * 64) * 16), 16, vB,
                                               kB + i * rndB, 0, 0);
    if (extra)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(sb + A_BYTES + B_MAIN + wave * 1024), 16, vBp, kB + pieceB,
                                               0, 0);
  };

  if constexpr (LW) {
    if (wave >= 8) {
      const int lw = wave - 8;
      const bool ex = lw < 2;                    // loaders 0, 1 also take one of the two B-piece KiB: 13 DMAs per K-tile
      const int rin = lane >> 3, cpl = lane & 7;
      const int keyA = (4 * (lw & 1)) | (rin >> 1);   // ((8 q + rin) >> 1) & 7 with q = lw + 4 j
      const int vLA = (int)(((long)rin * a.ldp + ((cpl ^ keyA) << 3)) * 2);
      int vLB, vLP;
      if constexpr (LAY == LAY_NT) {
        vLB = (int)(((long)rin * a.ldq + ((cpl ^ keyA) << 3)) * 2);
        vLP = vLB;
      } else {
        const int r4 = lane >> 4, chp = lane & 15;
        vLB = (int)(((long)r4 * a.ldq + ((chp ^ ((r4 << 2) | lw)) << 3)) * 2);   // tr_sw(4 q + r4), q & 3 == lw
        const int d = (lw & 1) * 64 + lane, rr = swap23(d >> 1);
        vLP = (int)(((long)rr * a.ldq + 128 + (d & 1) * 8) * 2);
      }
      auto lissue = [&](int t, int slot) {
        char* sb = smem + slot * STAGE4;
        const int kA = t * (BK4 * 2), kB = t * kstepB;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int q = lw + 4 * j;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_ptr_t)(sb + q * 1024), 16, vLA, kA + q * 8 * (int)a.ldp * 2, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = lw + 4 * j;
          const int so = (LAY == LAY_NT) ? q * 8 * (int)a.ldq * 2 : q * 4 * (int)a.ldq * 2;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(sb + A_BYTES + q * 1024), 16, vLB, kB + so, 0, 0);
        }
        if (ex) {
          const int so = (LAY == LAY_NT) ? (16 + lw) * 8 * (int)a.ldq * 2 : 0;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(sb + A_BYTES + B_MAIN + lw * 1024), 16, vLP, kB + so, 0, 0);
        }
      };
      lissue(0, 0);
      if (nt > 1) lissue(1, 1);
      if (nt > 2) lissue(2, 2);
      if (nt > 2) {
        if (ex) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      } else if (nt > 1) {
        if (ex) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      int slot = 0;
      for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) {
          if (t + 2 < nt) {
            if (ex) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_s_barrier();
          if (t + 3 < nt) lissue(t + 3, slot);
        }
        slot = slot == 2 ? 0 : slot + 1;
      }
      __syncthreads();
      return;
    }
  }

  // ---- fragment addresses (LDS byte offsets inside a stage) ----
  const int li = lane & 15, lg = lane >> 4, lq = li >> 2, lp = li & 3;
  const int rsw = (li >> 1) & 7;
  const unsigned lds0 = (unsigned)(size_t)(const char __attribute__((address_space(3)))*)smem;
  const unsigned aA = lds0 + (wr * 64 + li) * 128 + ((lg ^ rsw) << 4);              // ks = 1: ^ 64; row tile i: + 2048 i
  const unsigned bR = lds0 + A_BYTES + (wc * 80 + li) * 128 + ((lg ^ rsw) << 4);     // NT: column tile j: + 2048 j
  // NN: row0 = 32 ks + 8 lg + lq, row1 = row0 + 4; 16-column tile jt: ^ (jt << 5); ks = 1: + 8192
  const int sw0 = (lq << 2) | ((lg & 1) << 1);
  const unsigned tb0 = lds0 + A_BYTES + (8 * lg + lq) * 256 + ((((lp >> 1) ^ sw0)) << 4) + ((lp & 1) << 3);
  const unsigned tb1 = lds0 + A_BYTES + (8 * lg + lq + 4) * 256 + ((((lp >> 1) ^ (sw0 | 1))) << 4) + ((lp & 1) << 3);
  // NN piece (column tile 8 = wave column 1's j = 3): slot(row0) = lq | (lg & 1) << 2 | (lg >> 1) << 4 | ks << 5
  const unsigned tp0 = lds0 + A_BYTES + B_MAIN + (lq | ((lg & 1) << 2) | ((lg >> 1) << 4)) * 32 + lp * 8;
  const int jt0 = wc * 5;
  // j = 3 of wave column 1 reads the piece: a different base and a different k-half stride (1024 instead of 8192)
  const unsigned t3a = wc ? tp0 : (tb0 ^ (3u << 5)), t3b = wc ? tp0 + 256 : (tb1 ^ (3u << 5));
  const unsigned ks3 = wc ? 1024u : 8192u;
  const unsigned b4off = wc ? 6144u : 8192u;

  f32x4 accm[4][4], accx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) accm[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // reads of k-half KS of the stage at byte offset `so` into (Af, Bf)
#define LOADF(KS, so, Af, Bf)                                                            \
  do {                                                                                   \
    const unsigned pa_ = ((KS) ? (aA ^ 64u) : aA) + (so);                                \
    RD128(Af[0], pa_, 0); RD128(Af[1], pa_, 2048); RD128(Af[2], pa_, 4096); RD128(Af[3], pa_, 6144); \
    if constexpr (LAY == LAY_NT) {                                                       \
      const unsigned pb_ = ((KS) ? (bR ^ 64u) : bR) + (so);                              \
      RD128(Bf[0], pb_, 0); RD128(Bf[1], pb_, 2048); RD128(Bf[2], pb_, 4096); RD128(Bf[3], pb_, 6144); \
      RD128(Bf[4], pb_ + b4off, 0);   /* wave column 1 has no 5th tile: re-reads its 4th */ \
    } else {                                                                             \
      _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                    \
        const unsigned x_ = (unsigned)(jt0 + j) << 5;                                    \
        const unsigned q0_ = (tb0 ^ x_) + (so), q1_ = (tb1 ^ x_) + (so);                 \
        TR2(Bf[j], q0_, q1_, (KS) * 8192);                                               \
      }                                                                                  \
      {                                                                                  \
        const unsigned q0_ = t3a + (so) + ((KS) ? ks3 : 0u), q1_ = t3b + (so) + ((KS) ? ks3 : 0u); \
        TR2(Bf[3], q0_, q1_, 0);                                                         \
      }                                                                                  \
      {                                                                                  \
        const unsigned q0_ = (tb0 ^ (4u << 5)) + (so), q1_ = (tb1 ^ (4u << 5)) + (so);   \
        TR2(Bf[4], q0_, q1_, (KS) * 8192);   /* wave column 1: tile 4 of the main image, never used */ \
      }                                                                                  \
    }                                                                                    \
  } while (0)

#define MMA(Af, Bf)                                                                                          \
  do {                                                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
      accm[i][j] = REED_MFMA_16x16x32(__builtin_bit_cast(bf16x8, Bf[j]),                \
                                                           __builtin_bit_cast(bf16x8, Af[i]), accm[i][j]); \
    if (wc == 0) {                                                                                           \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        accx[i] = REED_MFMA_16x16x32(__builtin_bit_cast(bf16x8, Bf[4]),                 \
                                                          __builtin_bit_cast(bf16x8, Af[i]), accx[i]); \
    }                                                                                                        \
    __builtin_amdgcn_s_setprio(0);                                                                           \
  } while (0)

  // fragments live in integer vectors: with bf16 vector types the compiler re-packs loop-carried fragments element by
  // element (v_perm_b32) right behind the asm read that defines them, i.e. before the LDS data has arrived
  u32x4 A0f[4], B0f[5], A1f[4], B1f[5];

  // prologue: K-tiles 0, 1, 2 in flight; wait for tile 0
  if constexpr (!LW) {
  issue(0, 0);
  if (nt > 1) issue(1, 1);
  if (nt > 2) issue(2, 2);
  if (nt > 2) {
    if (extra) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  } else if (nt > 1) {
    if (extra) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  LOADF(0, 0u, A0f, B0f);
  LDS_WAIT0();   // landed before the loop header: a register copy the compiler may place on the loop edges is then safe

  // The two waves of a SIMD (w and w + 4 = wave columns 0 and 1) issue their share of a K-tile's DMA half a K-tile
  // apart — column 0 right behind the barrier of half 1, column 1 in the following half 0 — so that one of them is in
  // its MFMAs while the other spends its ~6 x 100 issue cycles on buffer_load ... lds.
  const bool late = wc != 0;
  int pend_t = 0, pend_slot = 0;
  bool pend = false;
  int slot = 0;   // ring slot of K-tile t
  for (int t = 0; t < nt; ++t) {
    const unsigned so = (unsigned)slot * STAGE4;
    const int nslot = slot == 2 ? 0 : slot + 1;
    // ---- half 0 ----  (reads(t, ks0) landed: waited at the end of the previous iteration / the prologue)
    LOADF(1, so, A1f, B1f);
    if constexpr (!LW) { if (pend) { issue(pend_t, pend_slot); pend = false; } }
    __builtin_amdgcn_sched_barrier(0);
    MMA(A0f, B0f);
    __builtin_amdgcn_sched_barrier(0);
    // ---- half 1 ----
    if (t + 1 < nt) {
      if constexpr (!LW) {
      if (t + 2 < nt) {              // K-tile t+1 landed; only t+2's DMAs may still be in flight
        if (extra) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      }
      LDS_WAIT0();                   // reads(t, ks1) landed: this wave is done with slot(t)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (!LW) {
      if (t + 3 < nt) {
        if (late) { pend = true; pend_t = t + 3; pend_slot = slot; }
        else issue(t + 3, slot);
      }
      }
      LOADF(0, (unsigned)nslot * STAGE4, A0f, B0f);
    } else {
      LDS_WAIT0();
    }
    __builtin_amdgcn_sched_barrier(0);
    MMA(A1f, B1f);
    __builtin_amdgcn_sched_barrier(0);
    LDS_WAIT0();                     // reads(t+1, ks0) landed long ago; keeps them ahead of any copy on the back edge
    slot = nslot;
  }
  __syncthreads();   // every wave is done with the ring: the epilogue's staging patches overlay slot 0

  char* patch = smem + wave * EPI_STAGE_BYTES;
  tile_epilogue<EPI, 4>(a, accm, m0, wr * 64, n0 + wc * 80, lane, 0, patch);
  if (wc == 0) strip_epilogue<EPI>(a, accx, m0 + wr * 64, n0 + 64, lane);   // columns 64 .. 79 of the tile
}

// (Round 5 built a persistent form of this kernel — one workgroup per CU walking its tiles as ONE stream of K-tiles through the ring,
// the next tile's first two K-tiles landing under the current tile's last two and its epilogue, bit-identical in 35 test cases —
// for the four-round launches of b = 32 per GPU (the 4608-wide outputs: 1024 tiles).  fc1 forward 0.106-0.107 -> 0.104-0.105 ms: the
// K loop of a 256x144 tile is not waiting for its prologue but for the L2 -> LDS operand stream (50 KiB per K-tile and CU = 12.8 MB
// per step chip-wide, 1.07 us at the ~12 TB/s that path delivers, against 0.6 us of MFMA work), and the NN instantiations need more
// than the 168 registers three waves per SIMD leave (0.100 -> 0.156 ms with the spills).  Removed; profiles/r5_gemm144_persistent_form.txt.)
static bool use_loader_waves() { return true; }   // (the 8-wave form without them: the epilogues that have no loader-wave instantiation)

template <int LAY, int EPI>
int launch144(const GemmArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  constexpr bool HAS_LW = EPI == EPI_BF16 || EPI == EPI_GATE_RES || EPI == EPI_GELU || EPI == EPI_DGELU || EPI == EPI_GELU_G ||
                          EPI == EPI_MUL;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm144_kernel<LAY, EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       LDS4);
    if constexpr (HAS_LW) {
      if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)gemm144_kernel<LAY, EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
    }
    if (e != hipSuccess) { reed_set_error("gemm144: cannot reserve 150 KiB LDS: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, BM4) * (a.N / BN4), 1, 1);
  if constexpr (HAS_LW) {
    if (use_loader_waves()) {
      REED_KLAUNCH((gemm144_kernel<LAY, EPI, true>), grid, dim3(768), LDS4, stream, a);
      REED_LAUNCH_CHECK();
      return REED_OK;
    }
  }
  REED_KLAUNCH((gemm144_kernel<LAY, EPI, false>), grid, dim3(512), LDS4, stream, a);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

template <int LAY>
int dispatch144(int epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_BF16: return launch144<LAY, EPI_BF16>(a, s);
    case EPI_GELU: return launch144<LAY, EPI_GELU>(a, s);
    case EPI_SILU: return launch144<LAY, EPI_SILU>(a, s);
    case EPI_GATE_RES: return launch144<LAY, EPI_GATE_RES>(a, s);
    case EPI_DGELU: return launch144<LAY, EPI_DGELU>(a, s);
    case EPI_DSILU: return launch144<LAY, EPI_DSILU>(a, s);
    case EPI_GELU_G: return launch144<LAY, EPI_GELU_G>(a, s);
    case EPI_SILU_G: return launch144<LAY, EPI_SILU_G>(a, s);
    case EPI_MUL: return launch144<LAY, EPI_MUL>(a, s);
    case EPI_QGELU: return launch144<LAY, EPI_QGELU>(a, s);
    case EPI_GELU_ERF: return launch144<LAY, EPI_GELU_ERF>(a, s);
    case EPI_RES_BF16: return launch144<LAY, EPI_RES_BF16>(a, s);
  }
  reed_set_error("reed_gemm(256x144): epilogue %d has no bf16-output form", epi);
  return REED_ERR_ARG;
}

}  // namespace

// shapes the 256x144 kernel can take: NT / NN, bf16-output epilogue, N a multiple of 144, no split-K
bool reed_gemm144_eligible(int layout, int epi, const GemmArgs& a, int splits) {
  const bool bf16_epi = epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_SILU || epi == EPI_GATE_RES || epi == EPI_DGELU ||
                        epi == EPI_DSILU || epi == EPI_QGELU || epi == EPI_GELU_ERF || epi == EPI_RES_BF16 || epi == EPI_GELU_G ||
                        epi == EPI_SILU_G || epi == EPI_MUL;
  return (layout == LAY_NT || layout == LAY_NN) && bf16_epi && splits <= 1 && a.N % BN4 == 0 && a.K % BK4 == 0 &&
         a.K >= BK4;
}

int reed_num_cus();   // gemm256.hip
double reed_gemm256_rate();

// Kernel selection against the 256^2 / 128^2 kernels, in gemm256.hip's units (one CU x one 128^2 tile; 256^2 tile = 4 /
// 1.18).  A 256x144 tile is 0.5625 of a 256^2 tile; with the loader waves its main loop runs at the chip's dense-MFMA
// ceiling when all 256 CUs are busy, but per round it exposes the same epilogue as the 256^2 kernel on 0.56 of the work,
// so over the block shapes it is worth ~0.92 of the 256^2 kernel per flop (tools/tile_ab.py and, in-step,
// a sweep of the threshold inside the step, b = 32 .. 256, round 2): cost = rounds x 2.25 / 0.92.  The 256^2 side: below two rounds in WHOLE rounds
// (ragged column tiles as workgroups: at 1.25 rounds — b = 64, N = 1152 — the second, quarter-full round costs a full
// tile time), from two rounds on gemm256.hip's own half-round model.  Outcome on SiT-XL/2: the five 1152-wide outputs
// (proj / fc2 forward, dgrads of qkv / proj / fc1) at b <= 64 per GPU and the two 4608-wide ones (fc1 forward, fc2
// dgrad) at b = 32; everything else stays where it was.
bool reed_gemm144_preferred(int layout, int epi, const GemmArgs& a, int splits) {
  if (!reed_gemm144_eligible(layout, epi, a, splits) || a.K < 256) return false;
  const int ncu = reed_num_cus();
  constexpr double eta = 0.92;
  const long tm = cdiv(a.M, 256), tn = cdiv(a.N, 256);
  const long t144 = (long)cdiv(a.M, BM4) * (a.N / BN4);
  const long t128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
  double r256 = (double)((tm * tn + ncu - 1) / ncu);
  if ((a.N % 256) != 0 && (a.N % 256) <= 128) {
    const double w = (double)tm * (tn - 1) + 0.6 * tm;
    if (w >= 2.0 * ncu) r256 = ceil(2.0 * w / ncu) / 2.0;
  }
  const double c144 = (double)((t144 + ncu - 1) / ncu) * 2.25 / eta;
  const double c256 = r256 * 4.0 / reed_gemm256_rate();
  const double c128 = (double)((t128 + 2 * ncu - 1) / (2 * ncu)) * 2.0;
  return c144 < c256 && c144 < c128;
}

int reed_gemm144_launch(int layout, int epi, GemmArgs a, hipStream_t stream) {
  if (layout == LAY_NT) return dispatch144<LAY_NT>(epi, a, stream);
  return dispatch144<LAY_NN>(epi, a, stream);
}
