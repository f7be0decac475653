// 256x256x64 MFMA GEMM with FOUR waves (one per SIMD), each owning a 128x128 piece of the output = 8x8 MFMA tiles
// in 256 accumulator registers (the unified 512-register file of a SIMD that holds a single wave).
//
// Why: in gemm256.hip's 2 x 4 wave grid a wave owns 128x64 and reads 12 operand fragments per 32 MFMAs; the LDS pipe
// (128 B/clk) is then exactly as busy as the MFMA pipe — per K-tile 192 KiB of fragment reads + 64 KiB of DMA writes =
// 2048 clk against 2 waves x 64 MFMAs x 16 clk = 2048 clk per SIMD — and every bank conflict or barrier skew shows.
// With 128x128 per wave it is 16 fragments per 64 MFMAs: 128 + 64 KiB = 1536 clk of LDS under the same 2048 clk of MFMA.
// Same LDS tile formats, staging (LDS-DMA half-tiles of 16 KiB, two K-tile buffers), layouts and epilogues as
// gemm256.hip, same accumulation order (bit-identical results).
//
// One K-tile = two phases of 64 MFMAs per wave (k-step 0, k-step 1):
//   phase A  wait own reads;           issue the 16 fragment reads of (t, ks1);   64 MFMAs on (t, ks0)
//   phase B  K-tile t+1 landed (vmcnt) + barrier: every wave is done reading K-tile t;
//            issue the 16 reads of (t+1, ks0); 64 MFMAs on (t, ks1) with the 16 DMAs of K-tile t+2 (into the buffer
//            K-tile t just left) spread between them, one per 4 MFMAs
#include <math.h>
#include <stdlib.h>

#include "gemm_common.hpp"

namespace {
using namespace gemm_detail;

constexpr int WBM = 256, WBN = 256, WBK = 64;
constexpr int HTW = 16384;                                  // half-tile bytes
constexpr int LDS_W = 8 * HTW + 4 * EPI_STAGE_BYTES;        // 144 KiB
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

template <int OFF>
__device__ __forceinline__ bf16x8 wtr2(const char* a0, const char* a1) {
  bf16x4 lo = ds_read_tr16_off<OFF>(a0);
  bf16x4 hi = ds_read_tr16_off<OFF>(a1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// RAGGED = a last column tile of which only the first 128 columns exist (N = 1152 = 4.5 x 256: every D-wide output of
// SiT-XL/2).  The four waves then take 64 rows x 128 columns each (row quarter `wave`, B half-tile 0): 4 x 8 MFMA tiles, half
// the MFMAs per phase, no DMA for the absent B half-tile — instead of two of the four waves multiplying zeros.
template <int LAY, int EPI, bool RAGGED>
__device__ __forceinline__ void gemm256w_body(const GemmArgs& a, char* smem, const int tm, const int tn) {
  constexpr int NI = RAGGED ? 4 : 8;          // 16-row tiles per wave
  constexpr int NCH = 2 * NI;                 // chunks of 4 MFMAs per phase
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = RAGGED ? 0 : (wave & 1);
  const int m0 = tm * WBM, n0 = tn * WBN;
  const int nt = (a.K + WBK - 1) / WBK;

  __amdgpu_buffer_rsrc_t rsP, rsQ;
  rsP = make_rsrc(a.P + (long)m0 * a.ldp, ((long)(a.M - m0) * a.ldp) * 2);
  if constexpr (LAY == LAY_NT) rsQ = make_rsrc(a.Q + (long)n0 * a.ldq, ((long)(a.N - n0) * a.ldq) * 2);
  else rsQ = make_rsrc(a.Q + n0, ((long)a.K * a.ldq - n0) * 2);

  const int vA0 = wvoff_row(tid, a.ldp);
  const int vB0 = (LAY == LAY_NT) ? wvoff_row(tid, a.ldq) : wvoff_tr(tid, a.ldq);
  const int rsA = (int)(32 * a.ldp * 2);                                        // per staging round
  const int rsB = (LAY == LAY_NT) ? (int)(32 * a.ldq * 2) : (int)(16 * a.ldq * 2);
  const int kstepA = WBK * 2;
  const int kstepB = (LAY == LAY_NT) ? WBK * 2 : (int)(WBK * a.ldq * 2);
  const int halfA = (int)(128 * a.ldp * 2);
  const int halfB = (LAY == LAY_NT) ? (int)(128 * a.ldq * 2) : 128 * 2;
  // one of the 16 DMA instructions of a K-tile: d = 4 * half-tile (A0 A1 B0 B1) + round
  auto dma = [&](int t, int cur, int d) {
    const int h4 = d >> 2, i = d & 3;
    if (h4 < 2) {
      char* ht = smem + wslotA(h4, cur);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_ptr_t)(ht + (i * 256 + wave * 64) * 16), 16, vA0,
                                               t * kstepA + h4 * halfA + i * rsA, 0, 0);
    } else {
      char* ht = smem + wslotB(h4 - 2, cur);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(ht + (i * 256 + wave * 64) * 16), 16, vB0,
                                               t * kstepB + (h4 - 2) * halfB + i * rsB, 0, 0);
    }
  };

  // ---- fragment addressing (gemm256.hip's formats) ----------------------------------------------------------------
  const int li = lane & 15, lg = lane >> 4, lq = li >> 2, lp = li & 3;
  const int rsw = (li >> 1) & 7;
  int rA = wslotA(wr, 0) + (RAGGED ? (wave & 1) * 8192 : 0) + li * 128 + ((lg ^ rsw) << 4);   // row-tile i -> + 2048 i, ks 1 -> ^ 64
  int rB = wslotB(wc, 0) + li * 128 + ((lg ^ rsw) << 4);
  // k-strided (transposing read): byte = row*256 + ((i' ^ xe)<<5) + (((p>>1)^hh)<<4) + ((p&1)<<3), row = ks*32 + 8g + q + 4hh
  const int xe = (lq << 1) | (lg & 1);
  int tB0 = wslotB(wc, 0) + (8 * lg + lq) * 256 + ((lp >> 1) << 4) + ((lp & 1) << 3);
  int tS = xe << 5;

  bf16x8 Af[2][NI], Bf[2][8];
  auto ldA = [&](int cur, int ks, int i) {
    const char* q = smem + (ks ? (rA ^ 64) : rA) + cur * HTW;
    Af[ks][i] = *(const bf16x8*)(q + i * 2048);
  };
  auto ldB = [&](int cur, int ks, int i) {
    if constexpr (LAY == LAY_NT) {
      const char* q = smem + (ks ? (rB ^ 64) : rB) + cur * HTW;
      Bf[ks][i] = *(const bf16x8*)(q + i * 2048);
    } else {
      const int sl = tS ^ (i << 5);
      const char* p0 = smem + (tB0 + sl);
      const char* p1 = smem + (((tB0 + 1024) ^ 16) + sl);
      if (cur == 0) Bf[ks][i] = ks ? wtr2<8192>(p0, p1) : wtr2<0>(p0, p1);
      else Bf[ks][i] = ks ? wtr2<HTW + 8192>(p0, p1) : wtr2<HTW>(p0, p1);
    }
  };
  // the fragment reads of a k-step spread over the phase's chunks: regular 16 chunks x 1 (even: A tile c/2, odd: B tile c/2);
  // ragged 8 chunks: B tile c, and A tile c for c < 4
  auto ldfrag = [&](int cur, int ks, int c) {
    if constexpr (RAGGED) {
      ldB(cur, ks, c);
      if (c < 4) ldA(cur, ks, c);
    } else {
      if (c & 1) ldB(cur, ks, c >> 1);
      else ldA(cur, ks, c >> 1);
    }
  };
  // the 16 (ragged: 12, B half-tile 1 does not exist) DMAs of a K-tile spread the same way
  auto dmas = [&](int t, int cur, int c) {
    if constexpr (RAGGED) {
      if (c < 6) { dma(t, cur, 2 * c); dma(t, cur, 2 * c + 1); }
    } else {
      dma(t, cur, c);
    }
  };

  f32x4 acc[NI][8];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define WMMA4(KS, I, J0)                                                                                 \
  _Pragma("unroll") for (int j = (J0); j < (J0) + 4; ++j)                                                \
      REED_MFMA_ACC(acc[(I)][j], Bf[(KS)][j], Af[(KS)][(I)]);

  // One phase = 16 chunks of {one fragment read for the NEXT phase, [one DMA of K-tile DMA_T], 4 MFMAs of k-step KS}: a
  // single wave feeds the matrix pipe, so everything else is issued in the shadow of the 4 x 16 clk a chunk's MFMAs take.
#define WPHASE(KS, LD_CUR, LD_KS, DMA_ON, DMA_T, DMA_CUR)                            \
  do {                                                                               \
    _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                \
      ldfrag((LD_CUR), (LD_KS), c);                                                  \
      if (DMA_ON) dmas((DMA_T), (DMA_CUR), c);                                       \
      WMMA4(KS, c >> 1, (c & 1) * 4);                                                \
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
#define WKTILE(T, CUR, PF)                                                                 \
  do {                                                                                     \
    const int t_ = (T);                                                                    \
    asm volatile("" : "+v"(rA), "+v"(rB), "+v"(tB0), "+v"(tS));                            \
    /* phase A: MFMAs of (t, ks0); reads of (t, ks1) */                                    \
    WLGKM0();                                                                              \
    WPHASE(0, (CUR), 1, false, 0, 0);                                                      \
    /* phase B: K-tile t+1 landed, every wave done with K-tile t; MFMAs of (t, ks1); reads of (t+1, ks0); DMAs of t+2 */ \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
    WLGKM0();                                                                              \
    WBARRIER();                                                                            \
    {                                                                                      \
      const bool pf_ = (PF);                                                               \
      WPHASE(1, 1 - (CUR), 0, pf_, t_ + 2, (CUR));                                         \
    }                                                                                      \
  } while (0)

  constexpr int ND = RAGGED ? 12 : 16;
  if (nt > 0) {
#pragma unroll
    for (int d = 0; d < ND; ++d) dma(0, 0, d);
    if (nt > 1) {
#pragma unroll
      for (int d = 0; d < ND; ++d) dma(1, 1, d);
      if constexpr (RAGGED) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    WBARRIER();
#pragma unroll
    for (int c = 0; c < NCH; ++c) ldfrag(0, 0, c);
  }
  int t = 0;
  for (; t + 3 < nt; t += 2) {
    WKTILE(t, 0, true);
    WKTILE(t + 1, 1, true);
  }
  for (; t + 1 < nt; t += 2) {
    WKTILE(t, 0, t_ + 2 < nt);
    WKTILE(t + 1, 1, t_ + 2 < nt);
  }
  if (t < nt) WKTILE(t, 0, false);

  // the MFMAs are inline asm: the compiler does not know the accumulators were just written by the matrix pipe
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  // epilogue: the wave's 128 (64) x 128 as two 64-column halves through gemm_common.hpp's tile_epilogue
  char* stage = smem + 8 * HTW + wave * EPI_STAGE_BYTES;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x4 part[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) part[i][j] = acc[i][h * 4 + j];
    tile_epilogue<EPI, NI, 4>(a, part, m0, RAGGED ? wave * 64 : wr * 128, n0 + wc * 128 + h * 64, lane, 0, stage);
  }
#undef WMMA4
#undef WPHASE
#undef WBARRIER
#undef WLGKM0
#undef WKTILE
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
  if (a.N - tn * WBN <= 128) gemm256w_body<LAY, EPI, true>(a, smem, tm, tn);
  else gemm256w_body<LAY, EPI, false>(a, smem, tm, tn);
}

template <int LAY, int EPI>
int launch256w(const GemmArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm256w_kernel<LAY, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       LDS_W);
    if (e != hipSuccess) { reed_set_error("gemm256w: cannot reserve LDS: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, WBM) * cdiv(a.N, WBN), 1, 1);
  GemmArgs b = a;
  b.tile_gm = 4;
  REED_KLAUNCH((gemm256w_kernel<LAY, EPI>), grid, dim3(256), LDS_W, stream, b);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

template <int LAY>
int dispatch256w(int epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_BF16: return launch256w<LAY, EPI_BF16>(a, s);
    case EPI_GELU: return launch256w<LAY, EPI_GELU>(a, s);
    case EPI_GATE_RES: return launch256w<LAY, EPI_GATE_RES>(a, s);
    case EPI_DGELU: return launch256w<LAY, EPI_DGELU>(a, s);
  }
  reed_set_error("reed_gemm(256w): epilogue %d not built", epi);
  return REED_ERR_UNSUPPORTED;
}

}  // namespace

bool reed_gemm256w_eligible(int layout, int epi, const GemmArgs& a, int splits) {
  return (layout == LAY_NT || layout == LAY_NN) && splits <= 1 && a.K % WBK == 0 && a.K >= 2 * WBK && a.N % 128 == 0 &&
         (epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_GATE_RES || epi == EPI_DGELU);
}

int reed_gemm256w_launch(int layout, int epi, GemmArgs a, hipStream_t stream) {
  if (layout == LAY_NT) return dispatch256w<LAY_NT>(epi, a, stream);
  return dispatch256w<LAY_NN>(epi, a, stream);
}
