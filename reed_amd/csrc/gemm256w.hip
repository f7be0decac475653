// 256x256x64 MFMA GEMM with FOUR waves (one per SIMD), each owning a 128x128 piece of the output = 8x8 MFMA tiles
// in 256 accumulator registers (the unified 512-register file of a SIMD that holds a single wave).
//
// Why: in gemm256.hip's 2 x 4 wave grid a wave owns 128x64 and reads 12 operand fragments per 32 MFMAs; with 128x128 per
// wave it is 16 fragments per 64 MFMAs — a third fewer LDS instructions (and DMA pieces per wave stay 16 per K-tile) in the
// issue slots the MFMAs leave.  (The design estimate counted LDS bytes at 128 B/clk: 192 + 64 KiB = 2048 clk per K-tile, level
// with the matrix pipe, against 128 + 64 KiB = 1536 clk; tools/micro/lds_rate.hip later measured 218 B/clk for ds_read_b128 at
// 8 waves per CU and 341 at 16, so the pipe was never byte-bound — the gain is in instructions issued per MFMA.)
// Same LDS tile formats, staging (LDS-DMA half-tiles of 16 KiB, two K-tile buffers), layouts and epilogues as
// gemm256.hip, same accumulation order (bit-identical results).
//
// One K-tile = two phases of 64 MFMAs per wave (k-step 0, k-step 1):
//   phase A  wait own reads;           issue the 16 fragment reads of (t, ks1);   64 MFMAs on (t, ks0)
// (A BK = 32 variant with four 32 KiB stages and the DMA issued four stages = two K-tiles ahead behind counted vmcnt waits
// was built and measured 7 % slower on the NN shapes: the K loop is not waiting for global memory, DESIGN.md.)
//   phase B  K-tile t+1 landed (vmcnt) + barrier: every wave is done reading K-tile t;
//            issue the 16 reads of (t+1, ks0); 64 MFMAs on (t, ks1) with the 16 DMAs of K-tile t+2 (into the buffer
//            K-tile t just left) spread between them, one per 4 MFMAs
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>
#include <type_traits>

#include "gemm_common.hpp"

int reed_num_cus();   // gemm256.hip
int reed_gemm_forced_tile();  // gemm.hip
int reed_concurrent_comm();   // gemm256.hip

#if defined(REED_CLK_PROBE) || defined(REED_CLK_PHASE)
// diagnostic build only (tools/_ab/build_variant.py clk -DREED_CLK_PROBE, read by tools/clk_probe.py): shader-clock and
// 100 MHz stamps around the K loop of every workgroup; the product build has no stamp
__device__ unsigned long long reed_clk_buf[8 * 8192];
extern "C" int reed_clk_probe_read(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(reed_clk_buf), sizeof(unsigned long long) * n);
}
#endif
// row groups of epilogue operand loads (residual stream / pre-activation) kept in flight ahead of the group being written
#ifndef REED_EPI_PF
#define REED_EPI_PF 4
#endif
// (Round 4 measured and round 5 removed two variants: an MFMA-layout epilogue without the LDS patch — bit-identical, plain 4.9 vs
// 4.4 us per tile, gate + residual 27 vs 22 — and counted waits that leave the previous persistent tile's epilogue stores in flight
// under K-tile 0 — equal; profiles/r4_epilogue_forms.txt.  A persistent tile drains (vmcnt(0)) at its start and at its K-tile 0.)
// 1: the four-wave TN form (weight gradients, REED_WGRAD_W4=1) walks its K-tile buffers as a ring of four 32-row slices (below)
#ifndef REED_TN_RING
#define REED_TN_RING 1
#endif
#ifndef REED_TN_Q_NT
#define REED_TN_Q_NT 0
#endif
// 1: a tile row's bias gradient dealt over its first four tiles (two row tiles each) — measured WORSE than 0, the split over the two
// waves of the row's first tile (1.688 vs 1.644 ms per launch in-step: the tiles of an XCD run in lockstep through the L2 they
// share, and four slightly slower tiles per row slow every XCD where one clearly slower tile per row delays only itself)
#ifndef REED_DB_DEAL
#define REED_DB_DEAL 0
#endif
namespace {
using namespace gemm_detail;

constexpr int WBM = 256, WBN = 256, WBK = 64;
constexpr int HTW = 16384;                                  // half-tile bytes
constexpr int LDS_W = 8 * HTW + 4 * EPI_STAGE_BYTES;        // 144 KiB
constexpr int LDS_TN = 8 * HTW;                             // 128 KiB: the grouped weight gradients (four half-tiles x two K-tile buffers)
__device__ __forceinline__ constexpr int wslotA(int h, int cur) { return (h * 2 + cur) * HTW; }
__device__ __forceinline__ constexpr int wslotB(int h, int cur) { return (4 + h * 2 + cur) * HTW; }

__device__ __forceinline__ int wvoff_row(int tid, long ld) {   // round 0 of a k-contiguous half-tile; round i: + 32 rows
  const int r = tid >> 3, cp = tid & 7;
  return (int)(((long)r * ld + (cp ^ ((r >> 1) & 7)) * 8) * 2);
}
__device__ __forceinline__ int wvoff_tr(int tid, long ld) {    // round 0 of a k-strided half-tile; round i: + 16 k-rows
  const int r = tid >> 4, chp = tid & 15;
  return (int)(((long)r * ld + (chp ^ tr_sw(r)) * 8) * 2);
}

// the same on LDS byte addresses (TN: the lane bases already include the LDS base of the dynamic segment; no pointer arithmetic)
template <int OFF>
__device__ __forceinline__ bf16x8 wtr2u(unsigned a0, unsigned a1) {
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "i"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "i"(OFF));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int OFF>
__device__ __forceinline__ bf16x8 wtr2(const char* a0, const char* a1) {
  bf16x4 lo = ds_read_tr16_off<OFF>(a0);
  bf16x4 hi = ds_read_tr16_off<OFF>(a1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// MODE bit 0 = ragged N: a last column tile of which only the first 128 columns exist (N = 1152 = 4.5 x 256: every D-wide
// output of SiT-XL/2); bit 1 = ragged M: a last row tile with 128 rows (TN: weight gradients of the 1152- / 3456-row weights).
// The four waves are re-dealt over what exists instead of multiplying zeros:
//   MODE 0  256 x 256: wave (wr, wc) = A half-tile wr (8 row tiles) x B half-tile wc (8 column tiles)     64 MFMAs / k-step
//   MODE 1  256 x 128: A half-tile wave>>1, 64-row quarter wave&1 (4 tiles) x B half-tile 0 (8 tiles)      32
//   MODE 2  128 x 256: A half-tile 0 (8 tiles) x B half-tile wave>>1, 64-column quarter wave&1 (4 tiles)   32
//   MODE 3  128 x 128: A half-tile 0 quarter wave>>1 (4) x B half-tile 0 quarter wave&1 (4)                16
// and the half-tiles that do not exist are not staged.
// PM = 1: the persistent form (gemm256wp_kernel): the workgroup walks a list of tiles; prev_mode < 0 = this is its first tile (it
// stages its own first two K-tiles), otherwise they were staged by the previous tile (of MODE prev_mode)'s last two K-tiles; nx_mode >= 0: a
// next tile (nx_tm, nx_tn) of MODE nx_mode (0 / 1) follows, and this tile's last two K-tiles stage ITS first two.
// DIRECT = 1 (TN, EPI_F32 without accumulation): the output leaves STRAIGHT from the accumulation registers (buffer stores with AGPR
// data, inline asm).  Through the compiler's own epilogue a kernel whose 256 AGPRs are all live moved part of them to VGPRs via
// scratch ("folded spills": 404-1208 bytes per lane in the grouped weight gradients since round 4) — no time in the K loop, but a
// private segment the runtime has to provide for the dispatch, and a launch time that depended on what the queue had run before
// (profiles/r6_wgrad_second_size_in_process.txt).
typedef int w_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void w_store_acc(const f32x4& q, w_i32x4 rs, int voff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"a"(q), "v"(voff), "s"(rs) : "memory");
}
template <int LAY, int EPI, int MODE, int PM = 0, int DIRECT = 0>
__device__ __forceinline__ void gemm256w_body(const GemmArgs& a, char* smem, const int tm, const int tn,
                                              const int prev_mode = -1, const int nx_mode = -1, const int nx_tm = 0,
                                              const int nx_tn = 0) {
  const bool first = prev_mode < 0;
  // (MODE 4 / 5, round 4: 512 x 128 / 128 x 512 tiles on five half-tiles per K-tile, 1.13 of a full tile's time per K-tile — gone with
  // the K-cut form of the grouped weight gradients, DESIGN_HISTORY.md)
  // MODE 6 / 7 / 8 (TN, one-shot only; round 6: the grouped weight gradients' ragged work as WHOLE-K items that keep a full tile's
  // pace, so that they walk the tokens in step with the full tiles of their rows and take those tiles' operand panels from the
  // XCD's L2 — see TnGroupW): 6 = 384 rows x 128 columns, waves 0..2 on A half-tile w x B half-tile 0 (128 x 128 each, the full
  // tile's instruction stream; four half-tiles per K-tile like a full tile), 7 = 128 x 384, A half-tile 0 x B half-tile w; wave 3
  // owns no output: it stages its share of the K-tiles, keeps the barriers, and forms the BIAS GRADIENT of the item's rows (one
  // MFMA per row tile and k-step against a fragment of ones: 24 / 8 per k-step where a computing wave issues 64).  8 = no output at
  // all: four A half-tiles (512 rows of dY), every wave the bias gradient of one.  tm / tn of these modes count 128-row / -column
  // units; rows / columns past a.M / a.N are computed and not stored.
  static_assert(MODE <= 3 || (MODE >= 6 && MODE <= 8 && LAY == LAY_TN && PM == 0), "three-unit / bias-only items: weight gradients only");
  constexpr bool RN = MODE <= 3 && (MODE & 1) != 0, RM = MODE <= 3 && (MODE & 2) != 0;
  constexpr int NA = RN ? 4 : 8;              // 16-row tiles per wave
  constexpr int NB = RM ? 4 : 8;              // 16-column tiles per wave
  constexpr int NCH = NA * NB / 4;            // chunks of 4 MFMAs per phase
  constexpr int NF = NA + NB;                 // fragment reads per k-step
  constexpr int NAH = MODE == 8 ? 4 : MODE == 6 ? 3 : (MODE == 7 || RM) ? 1 : 2;   // A / B half-tiles staged per K-tile
  constexpr int NBH = MODE == 7 ? 3 : MODE == 8 ? 0 : (MODE == 6 || RN) ? 1 : 2;
  constexpr int ND = 4 * (NAH + NBH);         // DMA instructions per K-tile per wave
  constexpr bool A_TR = LAY == LAY_TN, B_TR = LAY != LAY_NT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the waves that form a bias gradient instead of a piece of the output (MODE 6 / 7: wave 3, MODE 8: all)
  const bool brole = MODE == 8 || ((MODE == 6 || MODE == 7) && wave == 3);
  const int aHalf = (MODE == 6 || MODE == 8) ? wave : (MODE == 0 || MODE == 1) ? (wave >> 1) : 0;
  const int aQuart = MODE == 1 ? (wave & 1) : MODE == 3 ? (wave >> 1) : 0;
  const int bHalf = MODE == 7 ? wave : MODE == 0 ? (wave & 1) : MODE == 2 ? (wave >> 1) : 0;
  const int bQuart = (MODE == 2 || MODE == 3) ? (wave & 1) : 0;
  // LDS slot of half-tile h of buffer cur (modes 0-3: A 0..3, B 4..7; 6: A 0..5, B 6..7; 7: B 0..5, A 6..7; 8: A 0..7)
  auto slotA = [](int h, int cur) constexpr { return MODE == 7 ? (6 + cur) * HTW : wslotA(h, cur); };
  auto slotB = [](int h, int cur) constexpr { return MODE == 6 ? (6 + cur) * HTW : MODE == 7 ? (h * 2 + cur) * HTW : wslotB(h, cur); };
  const int mrow = aHalf * 128 + aQuart * 64, ncol = bHalf * 128 + bQuart * 64;
  const int m0 = MODE >= 6 ? tm * 128 : tm * WBM, n0 = MODE >= 6 ? tn * 128 : tn * WBN;
  const int nt = (a.K + WBK - 1) / WBK;

  __amdgpu_buffer_rsrc_t rsP, rsQ;
  if constexpr (A_TR) rsP = make_rsrc(a.P + m0, ((long)a.K * a.ldp - m0) * 2);
  else rsP = make_rsrc(a.P + (long)m0 * a.ldp, ((long)(a.M - m0) * a.ldp) * 2);
  if constexpr (B_TR) rsQ = make_rsrc(a.Q + n0, ((long)a.K * a.ldq - n0) * 2);
  else rsQ = make_rsrc(a.Q + (long)n0 * a.ldq, ((long)(a.N - n0) * a.ldq) * 2);

  const int vA0 = A_TR ? wvoff_tr(tid, a.ldp) : wvoff_row(tid, a.ldp);
  const int vB0 = B_TR ? wvoff_tr(tid, a.ldq) : wvoff_row(tid, a.ldq);
  const int rsA = A_TR ? (int)(16 * a.ldp * 2) : (int)(32 * a.ldp * 2);        // per staging round
  const int rsB = B_TR ? (int)(16 * a.ldq * 2) : (int)(32 * a.ldq * 2);
  const int kstepA = A_TR ? (int)(WBK * a.ldp * 2) : WBK * 2;
  const int kstepB = B_TR ? (int)(WBK * a.ldq * 2) : WBK * 2;
  const int halfA = A_TR ? 128 * 2 : (int)(128 * a.ldp * 2);
  const int halfB = B_TR ? 128 * 2 : (int)(128 * a.ldq * 2);
  // DMA e of the K-tile's ND: the existing half-tiles in the order A0 [A1] B0 [B1], 4 staging rounds each
  auto dma = [&](int t, int cur, int e) {
    const int hh = e >> 2, i = e & 3;
    if (hh < NAH) {
      char* ht = smem + slotA(hh, cur);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_ptr_t)(ht + (i * 256 + wave * 64) * 16), 16, vA0,
                                               t * kstepA + hh * halfA + i * rsA, 0, 0);
    } else {
      const int hb = hh - NAH;
      char* ht = smem + slotB(hb, cur);
      // REED_TN_Q_NT (build parameter, A/B): the weight gradients' X operand — a saved activation nobody reads after this kernel —
      // with the non-temporal policy
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(ht + (i * 256 + wave * 64) * 16), 16, vB0,
                                               t * kstepB + hb * halfB + i * rsB, 0, (LAY == LAY_TN && REED_TN_Q_NT) ? 2 : 0);
    }
  };

  // the next tile's first two K-tiles (PM): its own descriptors, its own list of half-tiles (MODE NM: A0 A1 B0 [B1])
  __amdgpu_buffer_rsrc_t rsPn = rsP, rsQn = rsQ;
  if constexpr (PM != 0) {
    if (nx_mode >= 0) {
      const int m0n = nx_tm * WBM, n0n = nx_tn * WBN;
      if constexpr (A_TR) rsPn = make_rsrc(a.P + m0n, ((long)a.K * a.ldp - m0n) * 2);
      else rsPn = make_rsrc(a.P + (long)m0n * a.ldp, ((long)(a.M - m0n) * a.ldp) * 2);
      if constexpr (B_TR) rsQn = make_rsrc(a.Q + n0n, ((long)a.K * a.ldq - n0n) * 2);
      else rsQn = make_rsrc(a.Q + (long)n0n * a.ldq, ((long)(a.N - n0n) * a.ldq) * 2);
    }
  }
  // always the full list A0 A1 B0 B1: for a ragged next tile the B1 pieces read beyond its columns (zeros from the range check
  // or the next row's first columns: staged, never read) — one form of the tail instead of one per next-tile MODE keeps the
  // control flow, and with it the allocation of the 256 accumulation registers, as simple as the one-shot kernel's
  auto dmas_next = [&](int kt, int cur, int c) {
    constexpr int NDN = 16;
#pragma unroll
    for (int e = 0; e < NDN; ++e) {
      if (e * NCH / NDN != c) continue;
      const int hh = e >> 2, i = e & 3;
      if (hh < 2) {
        char* ht = smem + wslotA(hh, cur);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsPn, (lds_ptr_t)(ht + (i * 256 + wave * 64) * 16), 16, vA0,
                                                 kt * kstepA + hh * halfA + i * rsA, 0, 0);
      } else {
        const int hb = hh - 2;
        char* ht = smem + wslotB(hb, cur);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQn, (lds_ptr_t)(ht + (i * 256 + wave * 64) * 16), 16, vB0,
                                                 kt * kstepB + hb * halfB + i * rsB, 0, 0);
      }
    }
  };

  // ---- fragment addressing (gemm256.hip's formats) ----------------------------------------------------------------
  const int li = lane & 15, lg = lane >> 4, lq = li >> 2, lp = li & 3;
  const int rsw = (li >> 1) & 7;
  // k-contiguous: row tile i -> + 2048 i, k-step 1 -> ^ 64
  int rA = slotA(aHalf, 0) + aQuart * 8192 + li * 128 + ((lg ^ rsw) << 4);
  int rB = slotB(bHalf, 0) + bQuart * 8192 + li * 128 + ((lg ^ rsw) << 4);
  // k-strided (transposing read): byte = row*256 + ((i' ^ xe)<<5) + (((p>>1)^hh)<<4) + ((p&1)<<3), row = ks*32 + 8g + q + 4hh,
  // i' = 16-column tile index inside the half-tile (a quarter starts at i' = 4)
  const int xe = (lq << 1) | (lg & 1);
  const int trow = (8 * lg + lq) * 256 + ((lp >> 1) << 4) + ((lp & 1) << 3);
  int tA0 = slotA(aHalf, 0) + trow, tB0 = slotB(bHalf, 0) + trow;
  int tSA = (xe << 5) ^ (aQuart << 7), tSB = (xe << 5) ^ (bQuart << 7);
  // TN (both operands transposing): the two read addresses of fragment i are (lane base) XOR (i << 5) — the row part of a base has
  // no bit in 5..7, where the slot swizzle and the tile index live — one vector instruction per read instead of add / xor / add
  const unsigned lds0 = (unsigned)(size_t)(const char __attribute__((address_space(3)))*)smem;   // a multiple of 256 at least
  unsigned uA0 = lds0 + tA0 + tSA, uA1 = lds0 + ((tA0 + 1024) ^ 16) + tSA, uB0 = lds0 + tB0 + tSB, uB1 = lds0 + ((tB0 + 1024) ^ 16) + tSB;
  // the bias role's read bases: the A half-tiles whose column sums it forms (MODE 6: 0..2, MODE 7: 0, MODE 8: the wave's own)
  constexpr int NBR = MODE == 6 ? 3 : 1;
  unsigned ubA0[NBR], ubA1[NBR];
  if constexpr (MODE >= 6) {
#pragma unroll
    for (int h = 0; h < NBR; ++h) {
      const int sb = slotA(MODE == 6 ? h : MODE == 8 ? wave : 0, 0) + trow;
      ubA0[h] = lds0 + sb + (xe << 5);
      ubA1[h] = lds0 + ((sb + 1024) ^ 16) + (xe << 5);
    }
  }

  bf16x8 Af[2][NA], Bf[2][NB];
  auto ldA = [&](int cur, int ks, int i) {
    if constexpr (!A_TR) {
      const char* q = smem + (ks ? (rA ^ 64) : rA) + cur * HTW;
      Af[ks][i] = *(const bf16x8*)(q + i * 2048);
    } else {
      if constexpr (LAY == LAY_TN) {
        const unsigned q0 = uA0 ^ (unsigned)(i << 5), q1 = uA1 ^ (unsigned)(i << 5);
        if (cur == 0) Af[ks][i] = ks ? wtr2u<8192>(q0, q1) : wtr2u<0>(q0, q1);
        else Af[ks][i] = ks ? wtr2u<HTW + 8192>(q0, q1) : wtr2u<HTW>(q0, q1);
        return;
      }
      const int sl = tSA ^ (i << 5);
      const char* p0 = smem + (tA0 + sl);
      const char* p1 = smem + (((tA0 + 1024) ^ 16) + sl);
      if (cur == 0) Af[ks][i] = ks ? wtr2<8192>(p0, p1) : wtr2<0>(p0, p1);
      else Af[ks][i] = ks ? wtr2<HTW + 8192>(p0, p1) : wtr2<HTW>(p0, p1);
    }
  };
  auto ldB = [&](int cur, int ks, int i) {
    if constexpr (!B_TR) {
      const char* q = smem + (ks ? (rB ^ 64) : rB) + cur * HTW;
      Bf[ks][i] = *(const bf16x8*)(q + i * 2048);
    } else {
      if constexpr (LAY == LAY_TN) {
        const unsigned q0 = uB0 ^ (unsigned)(i << 5), q1 = uB1 ^ (unsigned)(i << 5);
        if (cur == 0) Bf[ks][i] = ks ? wtr2u<8192>(q0, q1) : wtr2u<0>(q0, q1);
        else Bf[ks][i] = ks ? wtr2u<HTW + 8192>(q0, q1) : wtr2u<HTW>(q0, q1);
        return;
      }
      const int sl = tSB ^ (i << 5);
      const char* p0 = smem + (tB0 + sl);
      const char* p1 = smem + (((tB0 + 1024) ^ 16) + sl);
      if (cur == 0) Bf[ks][i] = ks ? wtr2<8192>(p0, p1) : wtr2<0>(p0, p1);
      else Bf[ks][i] = ks ? wtr2<HTW + 8192>(p0, p1) : wtr2<HTW>(p0, p1);
    }
  };
  // fragment f of a k-step's NF in the order A0 B0 A1 B1 ... (then what is left of the longer list)
  auto ldf = [&](int cur, int ks, int f) {
    constexpr int NMIN = NA < NB ? NA : NB;
    if (f < 2 * NMIN) {
      if (f & 1) ldB(cur, ks, f >> 1);
      else ldA(cur, ks, f >> 1);
    } else if constexpr (NA > NB) {
      ldA(cur, ks, f - NMIN);
    } else if constexpr (NB > NA) {
      ldB(cur, ks, f - NMIN);
    }
  };
  // the fragment reads / DMAs that ride in chunk c: entry e of N goes to chunk e * NCH / N
  auto ldfrag = [&](int cur, int ks, int c) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
      if (f * NCH / NF == c) ldf(cur, ks, f);
  };
  auto dmas = [&](int t, int cur, int c) {
#pragma unroll
    for (int e = 0; e < ND; ++e)
      if (e * NCH / ND == c) dma(t, cur, e);
  };
  // the ring walk of the TN form (below): the DMA pieces of one 32-row slice that ride in chunk c, and its counted wait
  auto dmas32 = [&](int t64, int cur, int ks, int c) {
#pragma unroll
    for (int e = 0; e < ND; ++e) {
      if (((e & 3) >> 1) != ks) continue;
      const int p = (e >> 2) * 2 + (e & 1);
      if (p * NCH / (ND / 2) == c) dma(t64, cur, e);
    }
  };
  auto ring_wait = [&](bool one) {
    if (one) {
      if constexpr (ND == 20) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if constexpr (ND == 16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (ND == 12) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      if constexpr (ND == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else if constexpr (ND == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if constexpr (ND == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  };

  f32x4 acc[NA][NB];
  if constexpr (PM == 0) {   // (PM: the first k-step's MFMAs take C = 0 — a zeroing pass here is loop-invariant in the
#pragma unroll               //  persistent kernel's tile loop and the compiler hoists it into 256 live VGPRs)
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // TN: bias gradient dbias[m] = sum_k P[k][m] as one more MFMA per row tile against a fragment of ones, in the waves that
  // own the tile's first columns of the first column tile; these accumulators live in VGPRs (the AGPRs are full)
  f32x4 accb[LAY == LAY_TN ? NA : 1];
  bool do_dbias = false;
  // MODE 0 with the ring loop: the two waves that share an A half-tile split its eight bias-gradient MFMAs per k-step by row-tile
  // parity (stamps: a bias-gradient tile ran 160 us longer than its XCD's other full tiles — the 16 extra MFMAs per K-tile of two
  // of its waves, which the barrier charges to all four).  db_par: -1 = this wave takes every row tile it owns, 0 / 1 = the even /
  // odd ones, 2 + q = row tiles 2 q and 2 q + 1
  int db_par = -1;
  bf16x8 ones;
  if constexpr (LAY == LAY_TN) {
    do_dbias = MODE < 6 && a.dbias != nullptr && tn == 0 && ncol == 0;   // (MODE >= 6: the bias role's waves, not the computing ones)
    if (MODE == 0 && PM == 0 && REED_TN_RING && REED_DB_DEAL && a.dbias != nullptr && a.N / WBN >= 4) {
      // a matrix with at least four full tile columns: the bias gradient of a tile row's 16 row tiles is dealt over its first
      // four tiles — tile tn takes row tiles 2 tn, 2 tn + 1 of each A half-tile, in the wave that owns the tile's first columns:
      // two more MFMAs per k-step in two waves instead of eight in two (or four in four)
      do_dbias = tn < 4 && (wave & 1) == 0;
      db_par = 2 + tn;
    } else if (MODE == 0 && PM == 0 && REED_TN_RING && a.dbias != nullptr && tn == 0) {
      do_dbias = true;
      db_par = wave & 1;
    }
    if constexpr (!(PM == 0 && REED_TN_RING)) {   // (the ring walk keeps its bias sums inside its own loop copy: nothing of them is
#pragma unroll                                  //  live across the other copies' loops)
      for (int i = 0; i < NA; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
  }

  // chunk c = 4 MFMAs: row tile CI(c), column tiles CJ(c) .. + 3
#define W_CI(c) (NB == 8 ? ((c) >> 1) : (c))
#define W_CJ(c) (NB == 8 ? (((c) & 1) * 4) : 0)
#define WMMA4(KS, C, Z)                                                                                  \
  _Pragma("unroll") for (int j = W_CJ(C); j < W_CJ(C) + 4; ++j) {                                        \
    if constexpr ((Z) != 0) REED_MFMA_ACC_Z(acc[W_CI(C)][j], Bf[(KS)][j], Af[(KS)][W_CI(C)]);            \
    else REED_MFMA_ACC(acc[W_CI(C)][j], Bf[(KS)][j], Af[(KS)][W_CI(C)]);                                 \
  }                                                                                                      \
  if constexpr (LAY == LAY_TN) {                                                                         \
    if (W_CJ(C) == 0 && do_dbias) REED_MFMA_ACC_V(accb[W_CI(C)], ones, Af[(KS)][W_CI(C)]);               \
  }

  // One phase = NCH chunks of {fragment reads for the NEXT phase, [DMAs of K-tile DMA_T], 4 MFMAs of k-step KS}: a single
  // wave feeds the matrix pipe, so everything else is issued in the shadow of the 4 x 16 clk a chunk's MFMAs take.
// KIND 0: the DMAs are this tile's K-tile DMA_T; 1: the NEXT tile's K-tile DMA_T (PM)
#define WPHASE(KS, LD_CUR, LD_KS, DMA_ON, DMA_T, DMA_CUR, KIND, Z)                   \
  do {                                                                               \
    _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                \
      ldfrag((LD_CUR), (LD_KS), c);                                                  \
      if (DMA_ON) {                                                                  \
        if constexpr ((KIND) == 0) dmas((DMA_T), (DMA_CUR), c);                      \
        else dmas_next((DMA_T), (DMA_CUR), c);                                       \
      }                                                                              \
      WMMA4(KS, c, Z);                                                               \
      __builtin_amdgcn_sched_barrier(0);                                             \
    }                                                                                \
  } while (0)

#define WBARRIER()                                         \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("" ::: "memory");                         \
  } while (0)
// lgkmcnt(0) as the BUILTIN (gfx9 encoding: vmcnt 63, expcnt 7, lgkmcnt 0): the compiler's wait-count pass sees it and
// does not add its own partial waits in front of the MFMAs of the phase
#define WLGKM0()                                           \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_waitcnt(0xC07F);                    \
    asm volatile("" ::: "memory");                         \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

  // PF: K-tile t+2 exists (literal true in the steady-state loop: no branch around the DMAs)
#define WKTILE(T, CUR, PF, KIND, KT) WKTILE_(T, CUR, PF, KIND, KT, 0)
#ifdef REED_CLK_PHASE   /* diagnostic build: cycles per phase (A, the waits + barrier, B), one stamp = s_memtime + lgkmcnt(0) */
#define WSTAMP(K)                                                                          \
  do {                                                                                     \
    unsigned long long now_;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    if ((K) >= 0) wph[(K) < 0 ? 0 : (K)] += now_ - wsp;                                    \
    wsp = now_;                                                                            \
  } while (0)
  unsigned long long wph[3] = {0, 0, 0}, wsp = 0;
#else
#define WSTAMP(K)
#endif
#define WKTILE_(T, CUR, PF, KIND, KT, Z)                                                    \
  do {                                                                                     \
    const int t_ = (T);                                                                    \
    asm volatile("" : "+v"(rA), "+v"(rB), "+v"(tA0), "+v"(tB0), "+v"(tSA), "+v"(tSB));     \
    /* phase A: MFMAs of (t, ks0); reads of (t, ks1) */                                    \
    WLGKM0();                                                                              \
    WSTAMP(-1);                                                                            \
    WPHASE(0, (CUR), 1, false, 0, 0, 0, Z);                                                \
    WSTAMP(0);                                                                             \
    /* phase B: K-tile t+1 landed, every wave done with K-tile t; MFMAs of (t, ks1); reads of (t+1, ks0); DMAs of t+2 */ \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
    WLGKM0();                                                                              \
    WBARRIER();                                                                            \
    WSTAMP(1);                                                                             \
    {                                                                                      \
      const bool pf_ = (PF);                                                               \
      WPHASE(1, 1 - (CUR), 0, pf_, ((KIND) == 0 ? t_ + 2 : (KT)), (CUR), KIND, 0);         \
    }                                                                                      \
    WSTAMP(2);                                                                             \
  } while (0)

  // The bias role (MODE >= 6: wave 3 of a 384-row / 384-column item, every wave of a bias-only item), from the end of the prologue
  // to its store: the same waits, barriers and DMA pieces per phase as the computing waves; per phase the column sums of slice
  // (cur, ks) of its A half-tiles — read, waited for and added within the phase (the computing waves read a slice one phase before
  // its MFMAs; this wave has no MFMA stream to hide a read behind and a tenth of their work).  Its sums and fragments live inside
  // this lambda only: nothing of them is carried across the computing waves' loop.
  auto bias_role = [&]() {
    if constexpr (MODE >= 6) {
      const bool bias_on = a.dbias != nullptr;
      f32x4 accB[NBR * 8];   // row tile i of its half-tile h in accB[8 h + i]
#pragma unroll
      for (int i = 0; i < NBR * 8; ++i) accB[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      // one phase: the reads of half-tile h + 1 are in flight while the MFMAs of half-tile h issue, and the phase's DMA pieces go out
      // behind the first reads (at most 16 LDS instructions are counted at a time: one half-tile's); three fragment buffers, so that
      // no read lands in registers an MFMA in flight still reads
      auto rd_group = [&](int cur, int ks, int h, bf16x8 (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned q0 = ubA0[h] ^ (unsigned)(i << 5), q1 = ubA1[h] ^ (unsigned)(i << 5);
          if (cur == 0) f[i] = ks ? wtr2u<8192>(q0, q1) : wtr2u<0>(q0, q1);
          else f[i] = ks ? wtr2u<HTW + 8192>(q0, q1) : wtr2u<HTW>(q0, q1);
        }
      };
      auto mm_group = [&](int h, const bf16x8 (&f)[8], bool on) {
        if (on) {
#pragma unroll
          for (int i = 0; i < 8; ++i) REED_MFMA_ACC_V(accB[h * 8 + i], ones, f[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto bias_step = [&](int cur, int ks, bool on, auto&& issue_dmas) {
        bf16x8 fa[8], fb[8], fc[8];
        if (bias_on) rd_group(cur, ks, 0, fa);
        issue_dmas();
        if (!bias_on) return;
        WLGKM0();
        if constexpr (NBR > 1) rd_group(cur, ks, 1, fb);
        mm_group(0, fa, on);
        if constexpr (NBR > 1) {
          WLGKM0();
          rd_group(cur, ks, 2, fc);
          mm_group(1, fb, on);
          WLGKM0();
          mm_group(2, fc, on);
        }
      };
      auto no_dmas = []() {};
#ifdef REED_CLK_PROBE
      unsigned long long ck0 = __builtin_amdgcn_s_memtime(), cr0 = __builtin_amdgcn_s_memrealtime();
      __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
      bias_step(0, 0, true, no_dmas);
#define WRINGB(T, CUR)                                                                       \
  do {                                                                                       \
    const int t_ = (T);                                                                      \
    ring_wait(false);                                                                        \
    WBARRIER();                                                                              \
    bias_step((CUR), 1, true, [&]() {                                                        \
      _Pragma("unroll") for (int c = 0; c < NCH; ++c) dmas32(t_ + 2, (CUR), 0, c);           \
    });                                                                                      \
    ring_wait(t_ == 0);                                                                      \
    WBARRIER();                                                                              \
    bias_step(1 - (CUR), 0, t_ + 1 < nt, [&]() {                                             \
      _Pragma("unroll") for (int c = 0; c < NCH; ++c) dmas32(t_ + 2, (CUR), 1, c);           \
    });                                                                                      \
  } while (0)
      int t = 0;
      for (; t + 1 < nt; t += 2) {
        WRINGB(t, 0);
        WRINGB(t + 1, 1);
      }
      if (t < nt) WRINGB(t, 0);
#undef WRINGB
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the MFMAs are inline asm: the sums were just written by the matrix pipe
#ifdef REED_CLK_PROBE
      {
        unsigned long long ck1 = __builtin_amdgcn_s_memtime(), cr1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        const int pidx = a.act_variant;
        if (tid == 0 && pidx >= 1000 && pidx < 8192) {
          reed_clk_buf[8 * pidx + 0] = ck1 - ck0;
          reed_clk_buf[8 * pidx + 1] = cr1 - cr0;
          reed_clk_buf[8 * pidx + 2] = nt;
          reed_clk_buf[8 * pidx + 3] = MODE;
          reed_clk_buf[8 * pidx + 4] = cr0;
          reed_clk_buf[8 * pidx + 5] = cr1;
          reed_clk_buf[8 * pidx + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;
          reed_clk_buf[8 * pidx + 7] = blockIdx.x;
        }
      }
#endif
      if (bias_on && (lane >> 4) == 0) {   // the column sums, one row per lane of the first 16
#pragma unroll
        for (int h = 0; h < NBR; ++h)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int m = m0 + (MODE == 6 ? h : MODE == 8 ? wave : 0) * 128 + i * 16 + (lane & 15);
            if (m < a.M) {
              if (a.accumulate) a.dbias[m] += accB[h * 8 + i][0];
              else a.dbias[m] = accB[h * 8 + i][0];
            }
          }
      }
    }
  };
  // PM: the previous tile staged this tile's K-tiles 0 and 1 beside its last two K-tiles; its epilogue's loads and stores are
  // YOUNGER than those DMAs in the same vmcnt: everything is drained here and at K-tile 0.
  if (PM != 0 && !first) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WBARRIER();
#pragma unroll
    for (int c = 0; c < NCH; ++c) ldfrag(0, 0, c);
  } else if (nt > 0) {
#pragma unroll
    for (int d = 0; d < ND; ++d) dma(0, 0, d);
    if (nt > 1) {
#pragma unroll
      for (int d = 0; d < ND; ++d) dma(1, 1, d);
      if constexpr (ND == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else if constexpr (ND == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if constexpr (ND == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    WBARRIER();
    if constexpr (MODE >= 6) {
      if (brole) {   // (every item has nt >= 2: the launcher's K >= 2 * WBK)
        bias_role();
        return;
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) ldfrag(0, 0, c);
  }
#ifdef REED_CLK_PROBE
  unsigned long long ck0 = __builtin_amdgcn_s_memtime(), cr0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  int t = 0;
#if REED_TN_RING
  // TN (weight gradients), round 4: the two K-tile buffers as a RING of four 32-row slices.  Both operands of a weight gradient
  // stream from HBM (a tile walks all b * 256 tokens of two column blocks), and with the loop above a slice is issued two phases
  // before its wait: the K-tile took 1.5 us whatever its MFMA count (MODE 1-3 tiles included) — a latency, not a rate.  Here slice
  // u + 4 is issued in phase u, right behind the barrier that says every wave has read slice u (same LDS bytes), and waited for
  // at the start of phase u + 3: three phases in flight instead of two, 8 DMAs per phase instead of 16 in every second one, a
  // barrier per phase.  Every slice is issued whatever its index (rows >= K are outside the descriptor: zeros, never read), so the
  // wait is one counted vmcnt: the two youngest slices (ND pieces) may be outstanding — in phase 1, where the prologue's K-tile 1
  // went out with its two slices interleaved, one (ND / 2).
  if constexpr (LAY == LAY_TN && PM == 0) {
    // DB: this workgroup also forms the bias gradient (one more MFMA per row tile against a fragment of ones; the tiles of a problem's
    // first column block only): its own copy of the loop — as a run-time condition it was a branch in every second chunk
#define WMMA4R(KS, C)                                                                                    \
  _Pragma("unroll") for (int j = W_CJ(C); j < W_CJ(C) + 4; ++j)                                          \
    REED_MFMA_ACC(acc[W_CI(C)][j], Bf[(KS)][j], Af[(KS)][W_CI(C)]);                                      \
  if constexpr (DB == 1) {                                                                               \
    if (W_CJ(C) == 0) REED_MFMA_ACC_V(accb[W_CI(C)], ones, Af[(KS)][W_CI(C)]);                           \
  } else if constexpr (DB == 2 || DB == 3) {                                                             \
    if (W_CJ(C) == 0 && (W_CI(C) & 1) == DB - 2) REED_MFMA_ACC_V(accb[W_CI(C)], ones, Af[(KS)][W_CI(C)]); \
  } else if constexpr (DB >= 4) {                                                                        \
    if (W_CJ(C) == 0 && (W_CI(C) >> 1) == DB - 4) REED_MFMA_ACC_V(accb[W_CI(C)], ones, Af[(KS)][W_CI(C)]); \
  }
#define WRING(T, CUR)                                                                        \
  do {                                                                                       \
    const int t_ = (T);                                                                      \
    asm volatile("" : "+v"(uA0), "+v"(uA1), "+v"(uB0), "+v"(uB1));                            \
    /* phase 2 t: MFMAs of (t, ks0); reads of (t, ks1); DMAs of (t + 2, ks0) into the slice (t, ks0) came from */ \
    ring_wait(false);                                                                        \
    WLGKM0();                                                                                \
    WBARRIER();                                                                              \
    _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                        \
      ldfrag((CUR), 1, c);                                                                   \
      dmas32(t_ + 2, (CUR), 0, c);                                                           \
      WMMA4R(0, c);                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                     \
    }                                                                                        \
    /* phase 2 t + 1: MFMAs of (t, ks1); reads of (t + 1, ks0); DMAs of (t + 2, ks1) */       \
    ring_wait(t_ == 0);                                                                      \
    WLGKM0();                                                                                \
    WBARRIER();                                                                              \
    _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                        \
      ldfrag(1 - (CUR), 0, c);                                                               \
      dmas32(t_ + 2, (CUR), 1, c);                                                           \
      WMMA4R(1, c);                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                     \
    }                                                                                        \
  } while (0)
    auto ring = [&](auto db_c) {
      constexpr int DB = decltype(db_c)::value;   // 0: no bias gradient, 1: every row tile, 2 / 3: the even / odd ones, 4 + q: 2 q, 2 q + 1
      f32x4 accb[NA];   // this copy's bias sums (VGPRs; untouched where DB == 0): not live in any other copy of the loop
      if constexpr (DB != 0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      for (; t + 1 < nt; t += 2) {
        WRING(t, 0);
        WRING(t + 1, 1);
      }
      if (t < nt) { WRING(t, 0); ++t; }
      if constexpr (DB != 0) {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        if ((lane >> 4) == 0) {
#pragma unroll
          for (int i = 0; i < NA; ++i) {
            const int m = m0 + mrow + i * 16 + (lane & 15);
            if (m < a.M && (DB == 1 || (DB < 4 ? (i & 1) == DB - 2 : (i >> 1) == DB - 4))) {
              if (a.accumulate) a.dbias[m] += accb[i][0];
              else a.dbias[m] = accb[i][0];
            }
          }
        }
      }
    };
    if (!do_dbias) ring(std::integral_constant<int, 0>{});
    else if (db_par < 0) ring(std::integral_constant<int, 1>{});
    else if constexpr (MODE == 0) {
      if (db_par == 0) ring(std::integral_constant<int, 2>{});
      else if (db_par == 1 || !REED_DB_DEAL) ring(std::integral_constant<int, 3>{});
      else if constexpr (REED_DB_DEAL != 0) {
        if (db_par == 2) ring(std::integral_constant<int, 4>{});
        else if (db_par == 3) ring(std::integral_constant<int, 5>{});
        else if (db_par == 4) ring(std::integral_constant<int, 6>{});
        else ring(std::integral_constant<int, 7>{});
      }
    }
#undef WRING
#undef WMMA4R
  } else
#endif
  {
  if constexpr (PM != 0) {   // nt >= 4 (the host checked): K-tile 0's first k-step starts the accumulation
    WKTILE_(0, 0, true, 0, 0, 1);
    WKTILE(1, 1, true, 0, 0);
    t = 2;
  }
  for (; t + 3 < nt; t += 2) {
    WKTILE(t, 0, true, 0, 0);
    WKTILE(t + 1, 1, true, 0, 0);
  }
  if constexpr (PM != 0) {   // nt is even (the host checked): the last two K-tiles stage the next tile's first two
    const bool has_next = nx_mode >= 0;
    for (; t + 1 < nt; t += 2) {
      WKTILE(t, 0, has_next, 1, 0);
      WKTILE(t + 1, 1, has_next, 1, 1);
    }
  } else {
    for (; t + 1 < nt; t += 2) {
      WKTILE(t, 0, t_ + 2 < nt, 0, 0);
      WKTILE(t + 1, 1, t_ + 2 < nt, 0, 0);
    }
    if (t < nt) WKTILE(t, 0, false, 0, 0);
  }
  }

  // the MFMAs are inline asm: the compiler does not know the accumulators were just written by the matrix pipe
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#ifdef REED_CLK_PHASE
  {
    const int pidx = tm * ((a.N + WBN - 1) / WBN) + tn;
    if (lane == 0 && wave == 0 && pidx < 8192 && MODE == 0) {
      reed_clk_buf[8 * pidx + 0] = wph[0];
      reed_clk_buf[8 * pidx + 1] = wph[1];
      reed_clk_buf[8 * pidx + 2] = wph[2];
      reed_clk_buf[8 * pidx + 3] = nt;
    }
  }
#endif
#ifdef REED_CLK_PROBE
  {
    unsigned long long ck1 = __builtin_amdgcn_s_memtime(), cr1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const int pidx = a.act_variant >= 1000 ? a.act_variant : tm * ((a.N + WBN - 1) / WBN) + tn;   // one record per tile (static items: 1000 + index)
    if (tid == 0 && pidx < 8192) {
      reed_clk_buf[8 * pidx + 0] = ck1 - ck0;
      reed_clk_buf[8 * pidx + 1] = cr1 - cr0;
      reed_clk_buf[8 * pidx + 2] = nt;
      reed_clk_buf[8 * pidx + 3] = MODE;
      reed_clk_buf[8 * pidx + 4] = cr0;   // K loop start / end on the 100 MHz clock (the kernel stamps entry and exit)
      reed_clk_buf[8 * pidx + 5] = cr1;
      if constexpr (LAY == LAY_TN) {      // the grouped weight gradients: where the tile ran (tools/_ab/clk_tn_w4.py)
        reed_clk_buf[8 * pidx + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;   // XCC_ID
        reed_clk_buf[8 * pidx + 7] = blockIdx.x;
      }
    }
  }
#endif
  if constexpr (DIRECT != 0) {
    static_assert(LAY == LAY_TN && EPI == EPI_F32, "direct stores: the weight gradients");
    // lane (lr, lc) owns row 16 i + lr, columns 16 j + lc .. + 3 of the wave's piece: one 16-byte store per accumulator quad; rows /
    // columns past a.M / a.N are whole 16-row / 16-column tiles (M, N multiples of 128): wave-uniform, sent past the descriptor
    const int lr = lane & 15, lc = 4 * (lane >> 4);
    const unsigned long long cb = (unsigned long long)((const char*)a.C + (long)m0 * a.ldc * 4);
    long nb = ((long)(a.M - m0 - 1) * a.ldc + a.N) * 4;
    if (nb < 0) nb = 0;
    if (nb > 0x7FFF0000l) nb = 0x7FFF0000l;
    const w_i32x4 rs = {(int)(unsigned)cb, (int)((cb >> 32) & 0xFFFFu), (int)nb, 0x00020000};
    const int ldc4 = (int)a.ldc * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int ml = mrow + 16 * i;
      const bool rv = m0 + ml < a.M;
      const int ro = (ml + lr) * ldc4 + (n0 + ncol + lc) * 4;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const bool ok = rv && n0 + ncol + 16 * j < a.N;
        w_store_acc(acc[i][j], rs, ok ? ro + 64 * j : EPI_OOB);
      }
    }
    return;   // (the bias gradient left from inside the ring walk's own loop copy)
  }
  // epilogue: the wave's piece in 64-column groups through gemm_common.hpp's tile_epilogue (fp32 outputs: its pointer path)
  char* stage = smem + 8 * HTW + wave * EPI_STAGE_BYTES;
#pragma unroll
  for (int h = 0; h < NB / 4; ++h) {
    f32x4 part[NA][4];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) part[i][j] = acc[i][h * 4 + j];
    tile_epilogue<EPI, NA, REED_EPI_PF>(a, part, m0, mrow, n0 + ncol + h * 64, lane, 0, stage);
  }
  if constexpr (LAY == LAY_TN && !(PM == 0 && REED_TN_RING)) {   // (the ring walk stores its bias gradient itself)
    if (do_dbias && (lane >> 4) == 0) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int m = m0 + mrow + i * 16 + (lane & 15);
        if (m < a.M && (db_par < 0 || (db_par < 2 ? (i & 1) == db_par : (i >> 1) == db_par - 2))) {
          if (a.accumulate) a.dbias[m] += accb[i][0];
          else a.dbias[m] = accb[i][0];
        }
      }
    }
  }
#undef W_CI
#undef W_CJ
#undef WMMA4
#undef WPHASE
#undef WBARRIER
#undef WLGKM0
#undef WKTILE
#undef WKTILE_
#undef WSTAMP
}

template <int LAY, int EPI>
__global__ __launch_bounds__(256, 1) void gemm256w_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntm = (a.M + WBM - 1) / WBM, ntn = (a.N + WBN - 1) / WBN;
  int tm, tn;
  {
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
  // (Tried: all full tiles first, then the half-width tiles of a ragged last column as one short round, to remove the tail a
  // mixed list leaves — simulated 9 % for fc2 forward. Measured WORSE, fc2 forward 0.621 -> 0.700 ms, fc1 dgrad 0.55 -> 0.60:
  // the 256 ragged tiles then stream the whole A operand again with no full tile of their rows beside them in L2.)
#ifdef REED_CLK_PROBE
  const unsigned long long en0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  if (a.N - tn * WBN <= 128) gemm256w_body<LAY, EPI, 1>(a, smem, tm, tn);
  else gemm256w_body<LAY, EPI, 0>(a, smem, tm, tn);
#ifdef REED_CLK_PROBE
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile's stores have left the wave
    const unsigned long long en1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const int pidx = tm * ntn + tn;
    if (threadIdx.x == 0 && pidx < 8192) {
      // physical CU: HW_ID (se_id[15:13] sh_id[12] cu_id[11:8]) and XCC_ID[3:0]
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      reed_clk_buf[8 * pidx + 6] = en0;
      reed_clk_buf[8 * pidx + 7] = (en1 << 16) | ((xcc & 0xF) << 8) | ((hw >> 8) & 0xFF);
    }
  }
#endif
}

// ---- persistent form: one workgroup per CU walks every (workgroups per XCD)-th tile of its XCD's run ----------------------
// Why (tools/clk_probe.py, b = 256, K = 1152): a 256^2 tile spends 1.8 us before its K loop (launch, address set-up, the first
// two K-tiles' DMA round trip), 24.6 us in it, 4.3 us in the epilogue, and the CU then waits 0.6 us for its next workgroup:
// here the next tile's first two K-tiles are staged beside the last two of the current one, so a tile starts with its
// operands in LDS.  And the STATIC deal removes the tail a greedy dispatch leaves with ragged tiles: for the 1152-wide
// outputs (4.5 tile columns) every workgroup gets exactly 4 full tiles + 1 half tile (positions s + 32 k of a run that
// repeats with period 10: two rows x five columns), where the hardware's greedy hand-out took 5.0-5.1 tile times for
// 4.5 tiles of work per CU.  The host uses this form only where its static deal is balanced (launch256wp).
__device__ __forceinline__ void w_tile_of(int b, int ntm, int ntn, int GM, int& tm, int& tn) {
  const int per_group = GM * ntn;
  const int group = b / per_group, first_m = group * GM;
  const int gs = min(ntm - first_m, GM);
  tm = first_m + (b % per_group) % gs;
  tn = (b % per_group) / gs;
}
template <int LAY, int EPI>
__global__ __launch_bounds__(256, 1) void gemm256wp_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntm = (a.M + WBM - 1) / WBM, ntn = (a.N + WBN - 1) / WBN;
  const int nwg = ntm * ntn;
  const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int run0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, runlen = q + (xcd < r ? 1 : 0);
  int p = s;
  if (p >= runlen) return;
  // (Measured and dropped in round 4: starting every other workgroup of an XCD 6 / 12 / 18 us late, so that half the chip is in
  // its K loops while the other half is in its epilogues — every shape of the block slower by about the delay itself, none
  // faster: fwd proj 0.241 -> 0.246, fc1 0.678 -> 0.694, dgrad fc2 0.709 -> 0.727 ms at 12 us; gpurun_out/r4a/stagger.txt.
  // Round 6 tried the stagger for FREE — every XCD owning 32 tile rows of a 4.5-column output, even XCDs running their half tiles
  // first, odd XCDs last, so that half the chip is 0.4 tile times ahead throughout — bit-identical and slower: fc2 forward 608 ->
  // 675 us, proj 250 -> 257, step - 0.9 %: the 32 half tiles of a slot have 32 different activation panels, where this deal runs
  // a half tile beside the full tiles of its row; profiles/r6_staggered_deal.txt, commit ffe488a)
  int tm, tn;
  w_tile_of(run0 + p, ntm, ntn, a.tile_gm, tm, tn);
  int pmode = -1;   // MODE of the previous tile of this workgroup (-1: none)
  for (;;) {
    const int pn = p + wpx;
    int nmode = -1, tmn = 0, tnn = 0;
    if (pn < runlen) {
      w_tile_of(run0 + pn, ntm, ntn, a.tile_gm, tmn, tnn);
      nmode = (a.N - tnn * WBN <= 128) ? 1 : 0;
    }
#ifdef REED_CLK_PROBE
    const unsigned long long en0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    const int cmode = (a.N - tn * WBN <= 128) ? 1 : 0;
    if (cmode) gemm256w_body<LAY, EPI, 1, 1>(a, smem, tm, tn, pmode, nmode, tmn, tnn);
    else gemm256w_body<LAY, EPI, 0, 1>(a, smem, tm, tn, pmode, nmode, tmn, tnn);
#ifdef REED_CLK_PROBE
    {
      const unsigned long long en1 = __builtin_amdgcn_s_memrealtime();   // stores issued (not waited for, as the kernel runs)
      __builtin_amdgcn_s_waitcnt(0xC07F);
      const int pidx = tm * ntn + tn;
      if (threadIdx.x == 0 && pidx < 8192) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        reed_clk_buf[8 * pidx + 6] = en0;
        reed_clk_buf[8 * pidx + 7] = (en1 << 16) | ((xcc & 0xF) << 8) | ((hw >> 8) & 0xFF);
      }
    }
#endif
    if (nmode < 0) break;
    p = pn;
    tm = tmn;
    tn = tnn;
    pmode = cmode;
  }
}

// ---- the weight gradients of one transformer block in one launch: one item per CU, dealt around tile rows per XCD -----------
// SiT-XL/2: fc1 4608x1152, fc2 1152x4608, qkv 3456x1152, proj 1152x1152 — every matrix has a 128-wide ragged edge (1152 = 4.5 x 256,
// 3456 = 13.5 x 256): 212 full 256^2 tiles + 124 units of 128^2 along the edges, 243 tile equivalents for 256 CUs.
// History (DESIGN_HISTORY.md): round 4 ran the edges as 256x128 / 128x256 / 512x128 / 128x512 tiles CUT ALONG K into lists that
// filled the 44 CUs beside the full tiles, partial tiles through a slab + a reduce kernel: 1.58 ms per launch at b = 256, with 7.4 GB
// through the L2s' memory side for 2.48 GB of operands — 3.2 GB of it the K-range pieces, which run at other token offsets than
// anything else on their XCD and share no operand panel — at 1.64-1.77 GHz, and the launch ending with the bias-gradient full
// tiles (one more MFMA per row tile and k-step: 1.0625 of a tile).
// Round 6: NO cut.  Every edge is covered by items of THREE units — 384x128 along a ragged column (MODE 6), 128x384 along a ragged
// row (MODE 7) — whose three computing waves run a full tile's instruction stream on four half-tiles per K-tile, i.e. at a full
// tile's pace over the whole K: 42 items (12 + 12 + 9 + 3 + 3 + 3) + 212 full tiles = 254 CUs.  The items of a matrix are dealt
// in the order of the tile rows they touch (a 384-row item right behind the full-tile rows whose dY panels it shares) and each
// XCD takes a contiguous 1/8 of that sequence: an item walks the tokens in step with the full tiles of its rows and finds their
// panels in the XCD's L2.  The fourth wave of an item forms the BIAS GRADIENT of its rows (the 384-row items of a matrix cover all
// its rows), fc2 — ragged row, no ragged column — gets two bias-only items (MODE 8) for its 1024 full-tile rows: no full tile carries
// a bias gradient any more.  No slab, no reduce kernel, and every tile is ONE K sequence: bit-identical to gemm_tn.hip's grouped
// kernel (the K-cut form differed in fp32 summation order).
struct TnGroupW {
  GemmArgs a[4];
  int n;
  int wpx;                // workgroups (= CUs) per XCD: workgroup b is item[(b & 7) * wpx + (b >> 3)]
  unsigned item[256];     // p | mode << 2 | row unit << 6 | column unit << 14 | (valid units - 1) of the row range << 22, of the column
                          // range << 24 | bias gradient << 26 (units of 128; mode 0 / 6 / 7 / 8); 0xFFFFFFFF: none
};
template <int ACC>   // 0: plain stores (DIRECT, scratch-free: the training step); 1: dw += (gradient accumulation over micro-batches)
__global__ __launch_bounds__(256, 1) void gemm256w_tn_group_kernel(TnGroupW g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned it = g.item[(blockIdx.x & 7) * g.wpx + (blockIdx.x >> 3)];
  if (it == 0xFFFFFFFFu) return;
  const int p = it & 3, mode = (it >> 2) & 15, mu = (it >> 6) & 255, nu = (it >> 14) & 255;
  const int vm = ((it >> 22) & 3) + 1, vn = ((it >> 24) & 3) + 1;
  // by value: through a reference into the kernel-argument array the compiler re-loads M, N, ldc (s_load + lgkmcnt(0)) in the
  // middle of the hand-placed K loop, where the SIMD's only wave then stands still (tools/isa_sload_scan.py)
  GemmArgs a = g.a[p];
  if (((it >> 26) & 1) == 0) a.dbias = nullptr;
#ifdef REED_CLK_PROBE
  a.act_variant = 2000 + (int)blockIdx.x;
#endif
  if (mode == 0) {
    gemm256w_body<LAY_TN, EPI_F32, 0, 0, 1 - ACC>(a, smem, mu >> 1, nu >> 1);
  } else if (mode == 6) {
    a.M = min(a.M, (mu + vm) * 128);
    gemm256w_body<LAY_TN, EPI_F32, 6, 0, 1 - ACC>(a, smem, mu, nu);
  } else if (mode == 7) {
    a.N = min(a.N, (nu + vn) * 128);
    gemm256w_body<LAY_TN, EPI_F32, 7, 0, 1 - ACC>(a, smem, mu, nu);
  } else {
    a.M = min(a.M, (mu + vm) * 128);
    gemm256w_body<LAY_TN, EPI_F32, 8, 0, 1 - ACC>(a, smem, mu, 0);
  }
}

// Tile rows per XCD-local group of the workgroup -> tile map (as gemm256.hip).
// Measured at b = 256 (tools/_ab/gm_w4.sh): the 1152-wide outputs (4.5 column tiles) prefer groups of 2 rows — fc2 forward
// 0.637 -> 0.618 ms, fc1 / qkv dgrads 0.564 -> 0.546 / 0.430 -> 0.417 —, the 3456- / 4608-wide ones groups of 4 (fc1 forward
// 0.684 vs 0.720 with 2).
int w_tile_group_rows(const GemmArgs& a) {
#ifdef REED_TILE_GM_ENV   // diagnostic build only (tools/r6/gm_sweep.sh): the tile rows per XCD-local group from the environment
  if (const char* e = getenv("REED_TILE_GM")) return atoi(e) > 0 ? atoi(e) : 4;
#endif
  if (cdiv(a.M, WBM) < 192) return 4;   // b = 128: 1157 (4 everywhere) vs 1151 images/s with the per-shape choice
  const int ntn = cdiv(a.N, WBN);
  return ntn <= 6 ? 2 : ntn >= 16 ? 5 : 4;   // (4608-wide: fc1 forward 0.688 -> 0.675, fc2 dgrad 0.727 -> 0.721 with 5)
}

// Static deal of the persistent form: the heaviest workgroup's load in full-tile units (a ragged tile counts RAG_COST) when
// wpx workgroups per XCD take positions s, s + wpx, ... of their XCD's run.
constexpr double RAG_COST = 0.58;
double w_static_max_load(int ntm, int ntn, bool rag, int GM, int wpx) {
  const int nwg = ntm * ntn, q = nwg >> 3, r = nwg & 7;
  double worst = 0;
  for (int xcd = 0; xcd < 8; ++xcd) {
    const int run0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, runlen = q + (xcd < r ? 1 : 0);
    for (int s = 0; s < wpx && s < runlen; ++s) {
      double load = 0;
      for (int p = s; p < runlen; p += wpx) {
        const int b = run0 + p, per_group = GM * ntn, group = b / per_group, first_m = group * GM;
        const int gs = std::min(ntm - first_m, GM), tn = (b % per_group) / gs;
        load += (rag && tn == ntn - 1) ? RAG_COST : 1.0;
      }
      worst = std::max(worst, load);
    }
  }
  return worst;
}

// mode 0: never (beside a collective; forced tile 257), 1: where the static deal is balanced (the default), 2: wherever the
// form applies (forced tile 258: tests)
template <int LAY, int EPI>
int launch256wp(const GemmArgs& a, hipStream_t stream, bool* used) {
  const int forced = reed_gemm_forced_tile();   // tests: 257 = the one-shot kernel, 258 = this form wherever it applies
  const int mode = forced == 257 ? 0 : forced == 258 ? 2 : reed_concurrent_comm() ? 0 : 1;
  *used = false;
  const int nt = cdiv(a.K, WBK), ntm = cdiv(a.M, WBM), ntn = cdiv(a.N, WBN);
  const int wpx = reed_num_cus() / 8;
  if (mode == 0 || (nt & 1) || nt < 4 || wpx < 1 || (a.K % WBK) != 0) return REED_OK;
  const bool rag = (a.N % WBN) != 0;
  const int GM = w_tile_group_rows(a);
  const double total = (double)ntm * ((ntn - (rag ? 1 : 0)) + (rag ? RAG_COST : 0.0));
  const double ideal = total / (8.0 * wpx);
  if (mode == 1 && ideal < 3.0) return REED_OK;       // too few tiles per workgroup for the hand-over to matter
  if (mode == 1) {
    // the verdict per (tile grid, group rows, workgroups per XCD) is remembered: the walk below is ~5 k steps on the host
    static thread_local struct { int ntm, ntn, gm, wpx, ok; } memo[8];   // ntn carries the ragged flag in its sign
    static thread_local int memo_n = 0;
    int ok = -1;
    for (int i = 0; i < memo_n; ++i)
      if (memo[i].ntm == ntm && memo[i].ntn == (rag ? -ntn : ntn) && memo[i].gm == GM && memo[i].wpx == wpx) ok = memo[i].ok;
    if (ok < 0) {
      const double worst = w_static_max_load(ntm, ntn, rag, GM, wpx);
      // the greedy hand-out of the one-shot kernel ends about half a tile after the balanced time when ragged tiles are mixed in
      const double greedy = ideal + (rag ? 0.45 : 0.0);
      ok = worst > greedy + 0.05 ? 0 : 1;
      memo[memo_n % 8] = {ntm, rag ? -ntn : ntn, GM, wpx, ok};
      if (memo_n < 8) ++memo_n;
    }
    if (!ok) return REED_OK;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm256wp_kernel<LAY, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       LDS_W);
    if (e != hipSuccess) { reed_set_error("gemm256wp: cannot reserve LDS: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  GemmArgs b = a;
  b.tile_gm = GM;
  REED_KLAUNCH((gemm256wp_kernel<LAY, EPI>), dim3(8 * wpx), dim3(256), LDS_W, stream, b);
  REED_LAUNCH_CHECK();
  *used = true;
  return REED_OK;
}

template <int LAY, int EPI>
int launch256w(const GemmArgs& a, hipStream_t stream) {
  {
    bool used = false;
    const int rc = launch256wp<LAY, EPI>(a, stream, &used);
    if (rc != REED_OK || used) return rc;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm256w_kernel<LAY, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       LDS_W);
    if (e != hipSuccess) { reed_set_error("gemm256w: cannot reserve LDS: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, WBM) * cdiv(a.N, WBN), 1, 1);
  GemmArgs b = a;
  b.tile_gm = w_tile_group_rows(a);
  REED_KLAUNCH((gemm256w_kernel<LAY, EPI>), grid, dim3(256), LDS_W, stream, b);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

template <int LAY>
int dispatch256w(int epi, const GemmArgs& a, hipStream_t s) {
  if constexpr (LAY == LAY_NT) {   // forward GEMMs: SiT blocks, the projector MLP, the frozen towers
    switch (epi) {
      case EPI_BF16: return launch256w<LAY, EPI_BF16>(a, s);
      case EPI_GELU: return launch256w<LAY, EPI_GELU>(a, s);
      case EPI_SILU: return launch256w<LAY, EPI_SILU>(a, s);
      case EPI_GELU_G: return launch256w<LAY, EPI_GELU_G>(a, s);
      case EPI_SILU_G: return launch256w<LAY, EPI_SILU_G>(a, s);
      case EPI_GATE_RES: return launch256w<LAY, EPI_GATE_RES>(a, s);
      case EPI_RES_BF16: return launch256w<LAY, EPI_RES_BF16>(a, s);
      case EPI_LS_RES: return launch256w<LAY, EPI_LS_RES>(a, s);
      case EPI_QGELU: return launch256w<LAY, EPI_QGELU>(a, s);
      case EPI_GELU_ERF: return launch256w<LAY, EPI_GELU_ERF>(a, s);
      // round 6: the input gradients as NT GEMMs on a transposed copy of the weights (engine.py: both operands k-contiguous)
      case EPI_BF16_DOT: return launch256w<LAY, EPI_BF16_DOT>(a, s);
      case EPI_DGELU: return launch256w<LAY, EPI_DGELU>(a, s);
      case EPI_MUL: return launch256w<LAY, EPI_MUL>(a, s);
    }
  } else {                         // input gradients (on the weights as they are: k-strided B operand)
    switch (epi) {
      case EPI_BF16: return launch256w<LAY, EPI_BF16>(a, s);
      case EPI_BF16_DOT: return launch256w<LAY, EPI_BF16_DOT>(a, s);
      case EPI_DGELU: return launch256w<LAY, EPI_DGELU>(a, s);
      case EPI_DSILU: return launch256w<LAY, EPI_DSILU>(a, s);
      case EPI_MUL: return launch256w<LAY, EPI_MUL>(a, s);
    }
  }
  reed_set_error("reed_gemm(256w): epilogue %d not built for this layout", epi);
  return REED_ERR_UNSUPPORTED;
}

}  // namespace

bool reed_gemm256w_eligible(int layout, int epi, const GemmArgs& a, int splits) {
  // (QuickGELU / exact-GELU epilogues stay on the 8-wave kernel: their VALU work — erff, two roundings per element — needs two
  // waves per SIMD to hide its own latency; measured 0.93 vs 0.64 ms on the ViT-L fc1 shape)
  const bool epi_ok = layout == LAY_NT ? (epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_SILU || epi == EPI_GATE_RES ||
                                          epi == EPI_GELU_G || epi == EPI_SILU_G || epi == EPI_RES_BF16 || epi == EPI_LS_RES ||
                                          epi == EPI_BF16_DOT || epi == EPI_DGELU || epi == EPI_MUL)
                                       : (epi == EPI_BF16 || epi == EPI_BF16_DOT || epi == EPI_DGELU || epi == EPI_DSILU || epi == EPI_MUL);
  return (layout == LAY_NT || layout == LAY_NN) && splits <= 1 && a.K % WBK == 0 && a.K >= 2 * WBK && a.N % 128 == 0 && epi_ok;
}

int reed_gemm256w_launch(int layout, int epi, GemmArgs a, hipStream_t stream) {
  if (layout == LAY_NT) return dispatch256w<LAY_NT>(epi, a, stream);
  return dispatch256w<LAY_NN>(epi, a, stream);
}

int reed_gemm_forced_tile();  // gemm.hip

// The deal of the grouped weight gradients (TnGroupW): n problems dw_i [M_i, N_i] with or without a bias gradient on `ncu` CUs.
// Fills item[256] in the kernel's layout (XCD x, workgroup j of it -> item[x * (ncu / 8) + j]; 0xFFFFFFFF = none) and returns the
// number of items, 0 where the form does not apply (the items do not fit one round of CUs, or fill it too thinly).  Host arithmetic
// only: reed_wgrad_group_deal exposes it to the planning side and to the CPU tests.
static int w4_deal(int n, const int* M, const int* N, const int* has_db, int ncu, unsigned* item) {
  if (n < 1 || n > 4 || ncu < 8 || (ncu & 7) != 0 || ncu > 256) return 0;
  const int wpx = ncu / 8;
  auto pack = [](int p, int mode, int mu, int nu, int vm, int vn, int db) {
    return (unsigned)(p | mode << 2 | mu << 6 | nu << 14 | (vm - 1) << 22 | (vn - 1) << 24 | db << 26);
  };
  // pass 1: the census — items without the bias-only ones, and how many of those the matrices without a ragged column would take
  int count = 0, bias_only = 0;
  double equiv = 0.0;
  for (int i = 0; i < n; ++i) {
    if (M[i] <= 0 || N[i] <= 0 || M[i] % 128 || N[i] % 128 || M[i] / 128 > 255 || N[i] / 128 > 255) return 0;
    const int Mu = M[i] / 128, Nu = N[i] / 128, fm = Mu / 2, fn = Nu / 2, rm = Mu & 1, rn = Nu & 1;
    count += fm * fn + (rn ? cdiv(Mu, 3) : 0) + (rm ? cdiv(2 * fn, 3) : 0);
    if (has_db[i] && !rn) bias_only += cdiv(2 * fm, 4);
    equiv += 0.25 * Mu * Nu;
  }
  // one round: every item gets a CU at once, and the CUs should be mostly busy
  constexpr double minfill = 0.7;
  if (count > ncu || equiv < minfill * ncu) return 0;
  const bool use_bias_only = count + bias_only <= ncu;   // otherwise those matrices' first tile column carries the extra MFMA
  // pass 2: every matrix's items in the order of the tile rows (columns) they touch, the matrices one after the other
  std::vector<unsigned> seq;
  for (int i = 0; i < n; ++i) {
    const int Mu = M[i] / 128, Nu = N[i] / 128, fm = Mu / 2, fn = Nu / 2, rm = Mu & 1, rn = Nu & 1;
    const bool colmajor = fm < fn;      // the shorter side of the tile grid runs fastest: an XCD's run is a compact block of panels
    const int db = has_db[i] ? 1 : 0;
    std::vector<std::pair<double, unsigned>> its;
    for (int u = 0; u < (colmajor ? fn : fm); ++u)
      for (int v = 0; v < (colmajor ? fm : fn); ++v) {
        const int tm = colmajor ? v : u, tn = colmajor ? u : v;
        its.push_back({u + 0.5, pack(i, 0, 2 * tm, 2 * tn, 2, 2, (db && !rn && !use_bias_only && tn == 0) ? 1 : 0)});
      }
    if (rn)   // the ragged column, corner included: 384-row items; their fourth waves form the whole matrix's bias gradient
      for (int c = 0; 3 * c < Mu; ++c)
        its.push_back({colmajor ? fn + 0.51 : (3 * c + 1.5) / 2.0, pack(i, 6, 3 * c, 2 * fn, std::min(3, Mu - 3 * c), 1, db)});
    if (rm)   // the ragged row (without the corner if a ragged column took it): 384-column items
      for (int c = 0; 3 * c < 2 * fn; ++c)
        its.push_back({colmajor ? (3 * c + 1.5) / 2.0 : fm + 0.51,
                       pack(i, 7, 2 * fm, 3 * c, 1, std::min(3, 2 * fn - 3 * c), (db && !rn && c == 0) ? 1 : 0)});
    if (db && !rn && use_bias_only)
      for (int c = 0; 4 * c < 2 * fm; ++c)
        its.push_back({(colmajor ? fn : fm) + 1.0, pack(i, 8, 4 * c, 0, std::min(4, 2 * fm - 4 * c), 1, 1)});
    std::stable_sort(its.begin(), its.end(),
                     [](const std::pair<double, unsigned>& x, const std::pair<double, unsigned>& y) { return x.first < y.first; });
    for (auto& e : its) seq.push_back(e.second);
  }
  const int T = (int)seq.size();
  if (T > ncu) return 0;
  for (int q = 0; q < 256; ++q) item[q] = 0xFFFFFFFFu;
  for (int x = 0; x < 8; ++x) {
    const int b = (int)((long)x * T / 8), e = (int)((long)(x + 1) * T / 8);
    for (int q = b; q < e; ++q) item[x * wpx + (q - b)] = seq[q];
  }
  return T;
}

extern "C" int reed_wgrad_group_deal(int n, const int* n_out, const int* k_in, const int* has_bias, int cus, unsigned* items) {
  if (!n_out || !k_in || !has_bias || !items) return 0;
  return w4_deal(n, n_out, k_in, has_bias, cus, items);
}

// 1 = launched; 0 = the problems do not suit this kernel (the caller falls back to gemm_tn.hip's grouped launch)
int reed_gemm256w_tn_group_launch(int n, const GemmArgs* probs, hipStream_t stream, int* launched) {
  *launched = 0;
  // The default where the items fill one round of CUs (see TnGroupW).  REED_WGRAD_W4=0: off (gemm_tn.hip's grouped launch
  // everywhere); =1: also beside a collective — A/B.
  static const int w4mode = getenv("REED_WGRAD_W4") ? atoi(getenv("REED_WGRAD_W4")) : -1;   // -1 = auto
  if (w4mode == 0 || reed_gemm_forced_tile() == 128) return REED_OK;   // force_tile 128: gemm_tn.hip's grouped kernel (tests, A/B)
  // Beside a collective (reed_set_concurrent_comm: the data-parallel backward) the static form is the wrong shape: its workgroups
  // take a whole CU each and every one of them carries 1 / 256 of the launch, so the ones that find their CU held by an RCCL
  // channel start when another workgroup has finished — the launch takes twice as long (measured with a stand-in that holds
  // 16 CUs: profiles/r4_wgrad_under_cu_hog.txt).  gemm_tn.hip's launch has half-size tiles in dynamic order: it loses what the
  // missing CUs carried.  REED_WGRAD_W4=1 forces this kernel there too (A/B on a multi-GPU node).
  if (reed_concurrent_comm() && w4mode != 1) return REED_OK;
  if (n < 1 || n > 4) return REED_OK;
  const int ncu = reed_num_cus();
  TnGroupW g;
  memset(&g, 0, sizeof(g));
  int M[4], N[4], db[4];
  for (int i = 0; i < n; ++i) {
    const GemmArgs& a = probs[i];
    if (a.K < 2 * WBK || a.slab_stride != 0 || a.K != probs[0].K) return REED_OK;
    M[i] = a.M; N[i] = a.N; db[i] = a.dbias != nullptr;
    g.a[i] = a;
  }
  if (w4_deal(n, M, N, db, ncu, g.item) == 0) return REED_OK;
  g.n = n;
  g.wpx = ncu / 8;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm256w_tn_group_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TN);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm256w_tn_group_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TN);
    if (e != hipSuccess) { reed_set_error("gemm256w: cannot reserve LDS: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  bool accum = false;
  for (int i = 0; i < n; ++i) accum = accum || probs[i].accumulate != 0;
  if (accum) REED_KLAUNCH(gemm256w_tn_group_kernel<1>, dim3(ncu), dim3(256), LDS_TN, stream, g);
  else REED_KLAUNCH(gemm256w_tn_group_kernel<0>, dim3(ncu), dim3(256), LDS_TN, stream, g);
  REED_LAUNCH_CHECK();
  *launched = 1;
  return REED_OK;
}
