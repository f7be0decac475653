// 256x256x64 bf16 MFMA GEMM for gfx950 — the large-problem kernel (same layouts, LDS tile formats and fused
// epilogues as gemm.hip; see gemm_common.hpp).  8 waves (2 x 4), each 128x64 of the output = 8x4 MFMA tiles;
// 128 KiB of LDS = 2 K-tile buffers x {A0, A1, B0, B1} half-tiles of 16 KiB (A_h feeds wave row h, B_h feeds wave
// columns 2h, 2h+1).  Halving the global->LDS bytes per flop versus the 128^2 tile is half of the gain; the other
// half is the pipeline: every half-tile is staged by LDS-DMA (buffer_load ... lds) 3-6 phases before its first
// use and the loop never drains the vector-memory queue — one counted s_waitcnt vmcnt(4) per K-tile, raw
// s_barrier — and the LDS fragment reads of the next phase are issued before this phase's 16 MFMAs.
//
// One K-tile = 4 phases; the wave's 128x64 output is cut into quadrants Q(mh, nh) of 64x32:
//   phase   MFMA (regs)        LDS reads issued (for)          LDS-DMA issued (K-tile)
//   p0      Q(0,0)  A0r,B0r    B-nh1(t)            (p1)        A1(t+1)
//   p1      Q(0,1)  A0r,B1r    A-mh1(t)            (p2)        B0(t+2)
//   p2      Q(1,1)  A1r,B1r    -                               B1(t+2)
//   p3      Q(1,0)  A1r,B0r    A-mh0,B-nh0(t+1)    (next p0)   A0(t+2)      [vmcnt(4) + barrier first]
// Each phase starts with s_waitcnt lgkmcnt(0) + s_barrier, which orders (WAR) the reads of a half-tile before the
// DMA that overwrites it two K-tiles later, and (RAW, at p3) the landed K-tile t+1 before its first fragment read.
#include <math.h>
#include <stdlib.h>

#include "gemm_common.hpp"

namespace {
using namespace gemm_detail;

constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int HT = 16384;        // half-tile bytes
constexpr int LDS_BYTES = 8 * HT + 8 * EPI_STAGE_BYTES;  // K-tile buffers + the waves' epilogue staging patches = 160 KiB
// LDS map: slot(operand half, K-tile buffer) = [A0c0 A0c1 A1c0 A1c1 B0c0 B0c1 B1c0 B1c1]: the two K-tile buffers of a
// half-tile are adjacent, so buffer select (cur*HT) and k-step (ks*8192) fit the 16-bit DS immediate offset.
__device__ __forceinline__ constexpr int slotA(int h, int cur) { return (h * 2 + cur) * HT; }
__device__ __forceinline__ constexpr int slotB(int h, int cur) { return (4 + h * 2 + cur) * HT; }

template <int OFF>
__device__ __forceinline__ bf16x8 tr2(const char* a0, const char* a1) {
  bf16x4 lo = ds_read_tr16_off<OFF>(a0);   // inline asm: see gemm_common.hpp (no compiler vmcnt(0) before it)
  bf16x4 hi = ds_read_tr16_off<OFF>(a1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// stage one half-tile (16 KiB) with 512 threads: 2 x 16 B per thread.  The per-thread byte offsets (voff) are
// loop invariant; everything that changes per K-tile / half goes through the scalar soffset.
__device__ __forceinline__ void stage_half(__amdgpu_buffer_rsrc_t rs, char* ht, int voff, int round2, int soff,
                                           int wave) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(ht + (wave * 64) * 16), 16, voff, soff, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(ht + (512 + wave * 64) * 16), 16, voff, soff + round2, 0, 0);
}
// per-thread offset of staging round i for a k-contiguous ("row") / k-strided ("tr") operand half-tile
__device__ __forceinline__ int voff_row(int i, int tid, long ld) {
  int L = i * 512 + tid, r = L >> 3, cp = L & 7;
  return (int)(((long)r * ld + (cp ^ ((r >> 1) & 7)) * 8) * 2);
}
__device__ __forceinline__ int voff_tr(int i, int tid, long ld) {
  int L = i * 512 + tid, r = L >> 4, chp = L & 15;
  return (int)(((long)r * ld + (chp ^ tr_sw(r)) * 8) * 2);
}

#define BARRIER()                                          \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("" ::: "memory");                         \
  } while (0)

// RAGGED = the tile is a last column tile of which only the first 128 columns exist (N = 1152 = 4.5 x 256 for every
// D-wide output of SiT-XL/2: proj, fc2, and the dgrads of qkv, proj, fc1).  With the regular 2 x 4 wave grid half the
// waves of such a tile would multiply zeros for the whole K loop; instead the tile is re-dealt: wave (wr, wc) takes
// the 64-row quadrant (wc >> 1) of its A half-tile and the 64 columns (wc & 1) of B half-tile 0 — a 4 x 4 grid of
// MFMA tiles per wave, two MFMA phases per K-tile instead of four, same staging, waits and barriers.  The two forms
// are separate instantiations of the body (disjoint register live ranges), selected per workgroup.
template <int LAY, int EPI, bool RAGGED>
__device__ __forceinline__ void gemm256_body(const GemmArgs& a, char* smem, const int tm, const int tn) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int z = blockIdx.y;
  const int m0 = tm * BM2, n0 = tn * BN2;
  const int kbeg = z * a.ksplit_len;
  const int kend = min(a.K, kbeg + a.ksplit_len);
  const int nt = (kend - kbeg + BK2 - 1) / BK2;

  __amdgpu_buffer_rsrc_t rsP, rsQ;
  if constexpr (LAY == LAY_TN) rsP = make_rsrc(a.P + m0, ((long)kend * a.ldp - m0) * 2);
  else rsP = make_rsrc(a.P + (long)m0 * a.ldp, ((long)(a.M - m0) * a.ldp) * 2);
  if constexpr (LAY == LAY_NT) rsQ = make_rsrc(a.Q + (long)n0 * a.ldq, ((long)(a.N - n0) * a.ldq) * 2);
  else rsQ = make_rsrc(a.Q + n0, ((long)kend * a.ldq - n0) * 2);

  // loop-invariant per-thread staging offsets
  // (the second staging round of a half-tile is 64 rows (row operand) / 32 rows (tr operand) further down; both
  // swizzle keys are periodic in that, so it is the same per-thread offset plus a scalar)
  const int vA0 = (LAY == LAY_TN) ? voff_tr(0, tid, a.ldp) : voff_row(0, tid, a.ldp);
  const int vB0 = (LAY == LAY_NT) ? voff_row(0, tid, a.ldq) : voff_tr(0, tid, a.ldq);
  const int r2A = (LAY == LAY_TN) ? (int)(32 * a.ldp * 2) : (int)(64 * a.ldp * 2);
  const int r2B = (LAY == LAY_NT) ? (int)(64 * a.ldq * 2) : (int)(32 * a.ldq * 2);
  // scalar byte offsets: per K-tile step and per half (128 rows of a row operand / 128 columns of a tr operand)
  const int kstepA = (LAY == LAY_TN) ? (int)(BK2 * a.ldp * 2) : BK2 * 2;
  const int kstepB = (LAY == LAY_NT) ? BK2 * 2 : (int)(BK2 * a.ldq * 2);
  const int halfA = (LAY == LAY_TN) ? 128 * 2 : (int)(128 * a.ldp * 2);
  const int halfB = (LAY == LAY_NT) ? (int)(128 * a.ldq * 2) : 128 * 2;
  const int kbaseA = (LAY == LAY_TN) ? (int)((long)kbeg * a.ldp * 2) : kbeg * 2;
  const int kbaseB = (LAY == LAY_NT) ? kbeg * 2 : (int)((long)kbeg * a.ldq * 2);
  // half-tile h of operand A (P) / B (Q) of K-tile t into buffer CUR
  auto issueA = [&](int t, int h, int cur) {
    stage_half(rsP, smem + slotA(h, cur), vA0, r2A, kbaseA + t * kstepA + h * halfA, wave);
  };
  auto issueB = [&](int t, int h, int cur) {
    stage_half(rsQ, smem + slotB(h, cur), vB0, r2B, kbaseB + t * kstepB + h * halfB, wave);
  };
  // ---- fragment addressing -------------------------------------------------------------------------------
  // k-contiguous operand (frag_row): byte = row*128 + ((c ^ ((row>>1)&7))<<4), row = tile*16 + (lane&15), c = ks*4 + g:
  //   lane base per ks (XOR of bit 2 of c), tile -> +2048 immediate.
  // k-strided operand (transposing read, see gemm_common.hpp frag_tr): byte = row*256 + ((i' ^ xe)<<5) +
  //   (((p>>1)^hh)<<4) + ((p&1)<<3), row = ks*32 + 8g + q + 4hh, xe = (q<<1)|(g&1), i' = 16-column tile index:
  //   lane bases per hh (+ the tile-index high bit), a 4-entry lane table for the low two tile bits, ks -> +8192.
  const int li = lane & 15, lg = lane >> 4, lq = li >> 2, lp = li & 3;
  const int xe = (lq << 1) | (lg & 1);
  const int xh = xe >> 2;
  const int aoff = slotA(wr, 0), boff = slotB(wc >> 1, 0);
  // Only ONE lane-dependent base per operand lives across the K loop; its siblings are derived inside the loop with
  // XOR/add on bit fields that no other term of the address touches:
  //   tr:  hh=1 base = (hh=0 base + 4 rows*256) ^ 16   (bit 4 = ((p>>1)^hh));  m-half 1 = base ^ 128 (bit 7 = mh^xh);
  //        slot of tile-low-bits k = S0 ^ (k<<5)       (bits 5-6 = k ^ (xe&3))
  //   row: ks=1 base = ks=0 base ^ 64                  (bit 6 = bit 2 of the chunk index ks*4+g, XOR-swizzled)
  int S0 = (xe & 3) << 5;
  int tA = aoff + (8 * lg + lq) * 256 + ((lp >> 1) << 4) + ((lp & 1) << 3) + (xh << 7);
  int tB = boff + (8 * lg + lq) * 256 + ((lp >> 1) << 4) + ((lp & 1) << 3) + (((wc & 1) ^ xh) << 7);
  const int rsw = (li >> 1) & 7;
  int rA = aoff + li * 128 + ((lg ^ rsw) << 4);
  int rB = boff + ((wc & 1) * 64 + li) * 128 + ((lg ^ rsw) << 4);
  if constexpr (RAGGED) {   // A: quadrant (wc >> 1) of half-tile wr; B: columns (wc & 1) * 64 .. of half-tile 0
    rA += (wc >> 1) * 8192;
    tA ^= (wc >> 1) << 7;
    rB -= (wc >> 1) * 2 * HT;
    tB -= (wc >> 1) * 2 * HT;
  }

  auto loadA = [&](int cur, int mh, bf16x8 (&f)[4][2]) {
    if constexpr (LAY == LAY_TN) {
      const int b0 = mh ? (tA ^ 128) : tA;
      const int b1 = (b0 + 1024) ^ 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sl = S0 ^ (i << 5);
        const char* p0 = smem + (b0 + sl);
        const char* p1 = smem + (b1 + sl);
        if (cur == 0) { f[i][0] = tr2<0>(p0, p1); f[i][1] = tr2<8192>(p0, p1); }
        else { f[i][0] = tr2<HT>(p0, p1); f[i][1] = tr2<HT + 8192>(p0, p1); }
      }
    } else {
      const char* q0 = smem + rA;
      const char* q1 = smem + (rA ^ 64);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f[i][0] = *(const bf16x8*)(q0 + cur * HT + (mh * 4 + i) * 2048);
        f[i][1] = *(const bf16x8*)(q1 + cur * HT + (mh * 4 + i) * 2048);
      }
    }
  };
  auto loadB = [&](int cur, int nh, bf16x8 (&f)[2][2]) {
    if constexpr (LAY == LAY_NT) {
      const char* q0 = smem + rB;
      const char* q1 = smem + (rB ^ 64);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f[j][0] = *(const bf16x8*)(q0 + cur * HT + (nh * 2 + j) * 2048);
        f[j][1] = *(const bf16x8*)(q1 + cur * HT + (nh * 2 + j) * 2048);
      }
    } else {
      const int b1 = (tB + 1024) ^ 16;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int sl = S0 ^ ((nh * 2 + j) << 5);
        const char* p0 = smem + (tB + sl);
        const char* p1 = smem + (b1 + sl);
        if (cur == 0) { f[j][0] = tr2<0>(p0, p1); f[j][1] = tr2<8192>(p0, p1); }
        else { f[j][0] = tr2<HT>(p0, p1); f[j][1] = tr2<HT + 8192>(p0, p1); }
      }
    }
  };

  constexpr int NI = RAGGED ? 4 : 8;
  f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define MMA(MH, NH, AF, BF)                                                                       \
  do {                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                 \
      acc[(MH) * 4 + i][(NH) * 2 + j] =                                                           \
          REED_MFMA_16x16x32(BF[j][ks], AF[i][ks], acc[(MH) * 4 + i][(NH) * 2 + j]); \
    __builtin_amdgcn_s_setprio(0);                                                                \
  } while (0)

  // One K-tile; BX holds B-nh0(t) on entry, BY receives B-nh1(t); on exit BY holds B-nh0(t+1) (roles swap).
#define KTILE(T, CUR, BX, BY)                                                   \
  do {                                                                          \
    const int t_ = (T);                                                         \
    asm volatile("" : "+v"(S0), "+v"(tA), "+v"(tB), "+v"(rA), "+v"(rB));        \
    /* p0 */                                                                    \
    BARRIER();                                                                  \
    if (t_ + 1 < nt) issueA(t_ + 1, 1, 1 - (CUR));                              \
    loadB((CUR), 1, BY);                                                        \
    MMA(0, 0, A0r, BX);                                                         \
    /* p1 */                                                                    \
    BARRIER();                                                                  \
    if (t_ + 2 < nt) issueB(t_ + 2, 0, (CUR));                                  \
    loadA((CUR), 1, A1r);                                                       \
    MMA(0, 1, A0r, BY);                                                         \
    /* p2 */                                                                    \
    BARRIER();                                                                  \
    if (t_ + 2 < nt) issueB(t_ + 2, 1, (CUR));                                  \
    MMA(1, 1, A1r, BY);                                                         \
    /* p3: K-tile t+1 must have landed before its first fragment read */        \
    if (t_ + 1 < nt) {                                                          \
      if (t_ + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     \
    }                                                                           \
    BARRIER();                                                                  \
    if (t_ + 2 < nt) issueA(t_ + 2, 0, (CUR));                                  \
    if (t_ + 1 < nt) {                                                          \
      loadA(1 - (CUR), 0, A0r); /* A0r is dead since p1 */                      \
      loadB(1 - (CUR), 0, BY);  /* BY (B-nh1) dead since p2 */                 \
    }                                                                           \
    MMA(1, 0, A1r, BX);                                                         \
  } while (0)

  // ragged tile: the wave's 64 x 64 piece is the quadrants Q(0,0), Q(0,1) of A0r (= its A quadrant); phases p2 and p3
  // keep their staging issues, waits and barriers but have no MFMAs
#define KTILE_R(T, CUR, BX, BY)                                                 \
  do {                                                                          \
    const int t_ = (T);                                                         \
    asm volatile("" : "+v"(S0), "+v"(tA), "+v"(tB), "+v"(rA), "+v"(rB));        \
    BARRIER();                                                                  \
    if (t_ + 1 < nt) issueA(t_ + 1, 1, 1 - (CUR));                              \
    loadB((CUR), 1, BY);                                                        \
    MMA(0, 0, A0r, BX);                                                         \
    BARRIER();                                                                  \
    if (t_ + 2 < nt) issueB(t_ + 2, 0, (CUR));                                  \
    MMA(0, 1, A0r, BY);                                                         \
    BARRIER();                                                                  \
    if (t_ + 2 < nt) issueB(t_ + 2, 1, (CUR));                                  \
    if (t_ + 1 < nt) {                                                          \
      if (t_ + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     \
    }                                                                           \
    BARRIER();                                                                  \
    if (t_ + 2 < nt) issueA(t_ + 2, 0, (CUR));                                  \
    if (t_ + 1 < nt) {                                                          \
      loadA(1 - (CUR), 0, A0r);                                                 \
      loadB(1 - (CUR), 0, BY);                                                  \
    }                                                                           \
  } while (0)

  bf16x8 A0r[4][2], A1r[RAGGED ? 1 : 4][2], Bp[2][2], Bq[2][2];
  if (nt > 0) {
    // prologue: K-tile 0 complete, K-tile 1 minus its A1 half in flight
    issueB(0, 0, 0); issueB(0, 1, 0); issueA(0, 0, 0); issueA(0, 1, 0);
    if (nt > 1) {
      issueB(1, 0, 1); issueB(1, 1, 1); issueA(1, 0, 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    BARRIER();
    loadA(0, 0, A0r);
    loadB(0, 0, Bp);
  }
  int t = 0;
  if constexpr (!RAGGED) {
    for (; t + 1 < nt; t += 2) {
      KTILE(t, 0, Bp, Bq);
      KTILE(t + 1, 1, Bq, Bp);
    }
    if (t < nt) KTILE(t, 0, Bp, Bq);
  } else {
    for (; t + 1 < nt; t += 2) {
      KTILE_R(t, 0, Bp, Bq);
      KTILE_R(t + 1, 1, Bq, Bp);
    }
    if (t < nt) KTILE_R(t, 0, Bp, Bq);
  }

  // One workgroup per output tile.  (A persistent variant — one workgroup per CU walking the tile list, the next
  // tile's prologue DMA issued under this epilogue, counted s_waitcnt across tiles — measured equal or slower at every
  // SiT-XL/2 shape: the hardware's workgroup hand-over already overlaps the store drain with the next launch.)
  // The epilogue stages through its own 32 KiB of LDS, so it needs no barrier against waves still in their last MFMAs.
  if constexpr (!RAGGED)
    tile_epilogue<EPI, 8>(a, acc, m0, wr * 128, n0 + wc * 64, lane, z, smem + 8 * HT + wave * EPI_STAGE_BYTES);
  else
    tile_epilogue<EPI, 4>(a, acc, m0, wr * 128 + (wc >> 1) * 64, n0 + (wc & 1) * 64, lane, z,
                          smem + 8 * HT + wave * EPI_STAGE_BYTES);
}

template <int LAY, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntm = (a.M + BM2 - 1) / BM2, ntn = (a.N + BN2 - 1) / BN2;
  int tm, tn;
  {
    // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs, so blocks L, L+8, ... share an L2; give
    // each XCD a contiguous run of tiles, walked in groups of GM tile rows x all tile columns.
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
      int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int GM = a.tile_gm;
    const int per_group = GM * ntn;
    const int group = bid / per_group, first_m = group * GM;
    const int gs = min(ntm - first_m, GM);
    tm = first_m + (bid % per_group) % gs;
    tn = (bid % per_group) / gs;
  }
  // fp32-output epilogues (pointer path: weight-gradient slabs, accumulating variants) and the TN layout (weight
  // gradients run on the 128^2 kernel) keep the regular form only
  if (LAY != LAY_TN && EpiOps<EPI, 8>::value > 0 && a.N - tn * BN2 <= 128)
    gemm256_body<LAY, EPI, true>(a, smem, tm, tn);
  else
    gemm256_body<LAY, EPI, false>(a, smem, tm, tn);
}

// Tile rows per group of the XCD-local tile walk (kernel: groups of GM tile rows x all tile columns, rows fastest).
int tile_group_rows(const GemmArgs& a) { return 4; }

template <int LAY, int EPI>
int launch256(const GemmArgs& a, int splits, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm256_kernel<LAY, EPI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) { reed_set_error("gemm256: cannot reserve 128 KiB LDS: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, BM2) * cdiv(a.N, BN2), splits, 1);
  GemmArgs b = a;
  b.tile_gm = tile_group_rows(a);
  REED_KLAUNCH((gemm256_kernel<LAY, EPI>), grid, dim3(512), LDS_BYTES, stream, b);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

template <int LAY>
int dispatch256(int epi, const GemmArgs& a, int splits, hipStream_t s) {
  if constexpr (LAY == LAY_TN) {   // weight gradients: fp32 outputs only (the bf16-output epilogues are never launched on TN)
    switch (epi) {
      case EPI_F32: return launch256<LAY, EPI_F32>(a, splits, s);
      case EPI_ADDF32_RB: return launch256<LAY, EPI_ADDF32_RB>(a, splits, s);
      case EPI_ATOMIC_F32: return launch256<LAY, EPI_ATOMIC_F32>(a, splits, s);
    }
  } else {
    switch (epi) {
      case EPI_BF16: return launch256<LAY, EPI_BF16>(a, splits, s);
      case EPI_GELU: return launch256<LAY, EPI_GELU>(a, splits, s);
      case EPI_SILU: return launch256<LAY, EPI_SILU>(a, splits, s);
      case EPI_GATE_RES: return launch256<LAY, EPI_GATE_RES>(a, splits, s);
      case EPI_DGELU: return launch256<LAY, EPI_DGELU>(a, splits, s);
      case EPI_DSILU: return launch256<LAY, EPI_DSILU>(a, splits, s);
      case EPI_GELU_G: return launch256<LAY, EPI_GELU_G>(a, splits, s);
      case EPI_SILU_G: return launch256<LAY, EPI_SILU_G>(a, splits, s);
      case EPI_MUL: return launch256<LAY, EPI_MUL>(a, splits, s);
      case EPI_F32: return launch256<LAY, EPI_F32>(a, splits, s);
      case EPI_ADDF32_RB: return launch256<LAY, EPI_ADDF32_RB>(a, splits, s);
      case EPI_ATOMIC_F32: return launch256<LAY, EPI_ATOMIC_F32>(a, splits, s);
      case EPI_QGELU: return launch256<LAY, EPI_QGELU>(a, splits, s);
      case EPI_GELU_ERF: return launch256<LAY, EPI_GELU_ERF>(a, splits, s);
      case EPI_RES_BF16: return launch256<LAY, EPI_RES_BF16>(a, splits, s);
      case EPI_LS_RES:
        if constexpr (LAY == LAY_NT) return launch256<LAY, EPI_LS_RES>(a, splits, s);
        break;
    }
  }
  reed_set_error("reed_gemm(256^2): epilogue %d is not built for layout %d", epi, LAY);
  return REED_ERR_ARG;
}

}  // namespace

// CUs the tile heuristics plan for = the device's count minus a reserve (reed_set_cu_reserve).  While a
// gradient bucket is in flight RCCL's channels hold CUs, and a grid planned as exactly one round of the 256 CUs — the 256x144
// tile at b = 32 per GPU, the grouped weight gradients' 512 slots — turns into two rounds on what is left.  The data-parallel
// train step measures a few reserves during its first steps and keeps the fastest (reed_amd/trainer.py).
static int g_cu_reserve = 0;
extern "C" int reed_set_cu_reserve(int n) { g_cu_reserve = n > 0 ? n : 0; return 0; }
int reed_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n - g_cu_reserve > 32 ? n - g_cu_reserve : 32;
}
extern "C" int reed_planning_cus(void) { return reed_num_cus(); }
// Collectives run beside the GEMMs (a data-parallel step): kernels that need a whole CU per workgroup for their whole run
// (the persistent form of gemm256w.hip) lose more than they gain when RCCL's channels hold some CUs — the workgroups that
// find no CU start when another finishes its entire list.  The one-shot kernels degrade gracefully; they are used then.
static int g_concurrent_comm = 0;
extern "C" int reed_set_concurrent_comm(int on) { g_concurrent_comm = on ? 1 : 0; return 0; }
int reed_concurrent_comm() { return g_concurrent_comm; }

// Kernel selection (NT forward / NN dgrad), a round-count model fitted to A/B timing at the SiT-XL/2 shapes for b = 64
// and 256 per GPU (tools/stagger_sweep.py): the 256^2 kernel does one tile per CU at a time and is ~1.18x faster per
// flop (half the global->LDS bytes, deeper pipeline); the 128^2 kernel keeps two tiles per CU in flight, which
// quantises better when the 256^2 grid is only one or two rounds.  Time in units of "one CU, one 128^2 tile":
//   t256 = ceil(tiles256 / CUs) * 4 / 1.18        t128 = ceil(tiles128 / (2 CUs)) * 2
// TN (wgrad) stays on the 128^2 kernel with wave-quantised split-K (ops.plan_wgrad); its 256^2 variant is reachable
// through reed_gemm_force_tile only.
// speed of the 256^2 kernel per flop relative to the 128^2 one in the round-count models
double reed_gemm256_rate() { return 1.18; }

bool reed_gemm256_preferred(int layout, int epi, const GemmArgs& a, int splits) {
  if (layout == LAY_TN || splits > 1 || a.K < 256) return false;
  const int ncu = reed_num_cus();
  const long tm = cdiv(a.M, BM2), tn = cdiv(a.N, BN2);
  // a ragged last column tile (<= 128 live columns; bf16-output epilogues) runs the re-dealt two-phase body: ~0.6 of a
  // full tile, and such tiles fill the tail of the last round — count rounds in halves when there are any
  // (A/B at b = 128: fc2 forward 0.407 -> 0.362 ms, fc1 / qkv dgrads 0.385 -> 0.322 / 0.285 -> 0.237 ms on 256^2)
  const bool ragged = (a.N % BN2) != 0 && (a.N % BN2) <= 128 &&
                      (epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_SILU || epi == EPI_GATE_RES || epi == EPI_DGELU ||
                       epi == EPI_DSILU || epi == EPI_QGELU || epi == EPI_GELU_ERF || epi == EPI_RES_BF16 || epi == EPI_LS_RES ||
                       epi == EPI_GELU_G || epi == EPI_SILU_G || epi == EPI_MUL);
  double rounds256;
  if (ragged) {
    const double w = (double)tm * (tn - 1) + 0.6 * tm;
    rounds256 = ceil(2.0 * w / ncu) / 2.0;
  } else {
    rounds256 = (double)((tm * tn + ncu - 1) / ncu);
  }
  const long t128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
  const double c256 = rounds256 * 4.0 / reed_gemm256_rate();
  const double c128 = (double)((t128 + 2 * ncu - 1) / (2 * ncu)) * 2.0;
  return c256 < c128;
}

int reed_gemm256_launch(int layout, int epi, GemmArgs a, int splits, hipStream_t stream) {
  switch (layout) {
    case LAY_NT: return dispatch256<LAY_NT>(epi, a, splits, stream);
    case LAY_NN: return dispatch256<LAY_NN>(epi, a, splits, stream);
    case LAY_TN: return dispatch256<LAY_TN>(epi, a, splits, stream);
  }
  reed_set_error("reed_gemm: unknown layout %d", layout);
  return REED_ERR_ARG;
}
