// Weight-gradient GEMM (TN: C[M,N] f32 (+)= P[K,M]^T Q[K,N], K = tokens) on a 256x128 (or 128x256) output tile with
// 128x64 (64x128) wave tiles — the dominant kernel of the SiT train step (reference: autograd of every nn.Linear,
// image/models/sit.py:17-24,114-129; timm Attention / Mlp).
//
// Why another tile: the 128^2 kernel of gemm.hip gives each of its 4 waves a 64x64 piece, i.e. 8 fragment reads (16
// ds_read_b64_tr_b16, 8 KiB) per 16 MFMAs; with two workgroups per CU the LDS pipe is then as loaded as the MFMA pipe
// (profiles/r1_pmc_gemm.txt: MFMA busy 50 %).  Here a wave owns 128x64: 12 fragments per 32 MFMAs (0.375 instead of
// 0.5 reads per MFMA) and the workgroup stages 24 KiB instead of 32 KiB per 32 MFMAs per wave.  BK = 32 keeps the
// barrier cadence (one per 32 MFMAs per wave), the LDS footprint (48 KiB double-buffered) and the occupancy (4 waves,
// two workgroups per CU) of the 128^2 kernel.  256x128 suits outputs whose row count is a multiple of 256 or large
// (fc1 4608x1152, qkv 3456x1152: the half-empty 14th tile row costs 3.6 %); WIDE = 128x256 is the same kernel turned
// for fc2 (1152x4608).  Ragged M needs no masking: rows >= M read whatever the descriptor returns and are never stored.
// Operand tiles use gemm_common.hpp's k-strided format ([k][128 columns], 256-B rows, chunk swizzle tr_sw) at 32
// k-rows per tile and are read with the transposing LDS read; staging is LDS-DMA as in gemm.hip.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "gemm_common.hpp"

#if defined(REED_CLK_PROBE) || defined(REED_CLK_SEG)
// diagnostic builds only (tools/_ab/build_variant.py clk -DREED_CLK_PROBE, tools/clk_probe_tn.py: stamps around the K loop;
// seg -DREED_CLK_SEG, tools/_ab/seg_tn.py: per-segment stamps inside it)
__device__ unsigned long long reed_clk_buf_tn[4 * 1024];
extern "C" int reed_clk_probe_read_tn(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(reed_clk_buf_tn), sizeof(unsigned long long) * n);
}
#endif
int reed_num_cus();   // gemm256.hip
namespace {
using namespace gemm_detail;

constexpr int TBK = 32;
constexpr int SUB_BYTES = 32 * 256;   // one [32 k][128 col] sub-tile

// stage one [32][128] sub-tile: 512 16-byte chunks, 2 per thread; rs is based at the tile's first column of row 0
__device__ __forceinline__ void stage_sub(__amdgpu_buffer_rsrc_t rs, char* tile, long ld, int k0, int col0, int tid,
                                          int wave) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int L = i * 256 + tid;
    const int r = L >> 4, chp = L & 15;
    const int ch = chp ^ tr_sw(r);
    const int voff = (int)(((long)(k0 + r) * ld + col0 + ch * 8) * 2);
    char* dst = tile + (i * 256 + wave * 64) * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
  }
}

// (tm, tn): the output tile; z / nz: K slice and slice count (1 = write C directly)
template <bool WIDE>
__device__ __forceinline__ void gemm_tn_body(const GemmArgs& a, char* smem, const int tm, const int tn, const int z,
                                             const int nz) {
  constexpr int TM = WIDE ? 4 : 8, TNN = WIDE ? 8 : 4;        // 16x16 MFMA tiles per wave along M / N
  constexpr int PSUB = WIDE ? 1 : 2, QSUB = WIDE ? 2 : 1;     // 128-column sub-tiles of the P / Q operand tile
  constexpr int BMT = PSUB * 128, BNT = QSUB * 128;
  constexpr int STAGE = (PSUB + QSUB) * SUB_BYTES;            // 24 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int m0 = tm * BMT, n0 = tn * BNT;
  const int kbeg = z * a.ksplit_len;
  const int kend = min(a.K, kbeg + a.ksplit_len);
  const int nt = (kend - kbeg + TBK - 1) / TBK;

  const __amdgpu_buffer_rsrc_t rsP = make_rsrc(a.P + m0, ((long)kend * a.ldp - m0) * 2);
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(a.Q + n0, ((long)kend * a.ldq - n0) * 2);
  // the leading dimensions in registers: `a` lives in kernel-argument memory, and every asm statement with a "memory" clobber in
  // the K loop (the LDS waits) made the compiler load them again — an s_load + s_waitcnt lgkmcnt(0) in front of each K-tile's DMAs
  const long ldp = a.ldp, ldq = a.ldq;
  auto stage = [&](int t, int buf) {
    char* tp = smem + buf * STAGE;
    const int k0 = kbeg + t * TBK;
#pragma unroll
    for (int s = 0; s < PSUB; ++s) stage_sub(rsP, tp + s * SUB_BYTES, ldp, k0, s * 128, tid, wave);
#pragma unroll
    for (int s = 0; s < QSUB; ++s) stage_sub(rsQ, tp + (PSUB + s) * SUB_BYTES, ldq, k0, s * 128, tid, wave);
  };

  f32x4 acc[TM][TNN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TNN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_dbias = a.dbias != nullptr && tn == 0 && wn == 0;
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
  // the wave's rows / columns inside the operand tiles: sub-tile + column base
  const int prow = wm * (TM * 16), pcol = wn * (TNN * 16);

  if (nt > 0) stage(0, 0);
  __syncthreads();
#ifdef REED_CLK_PROBE
  const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), cr0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
#ifdef REED_CLK_SEG
  // diagnostic build (tools/_ab/build_variant.py seg -DREED_CLK_SEG, tools/_ab/seg_tn.py): where a wave's time goes inside
  // the K loop (one stamp = s_memtime + lgkmcnt(0), ~40 cycles each; the stamped kernel runs ~10 % slower)
  unsigned long long sg[5] = {0, 0, 0, 0, 0}, sp;
#define SEG_STAMP(K)                                                                        \
  do {                                                                                      \
    unsigned long long now_;                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    if ((K) >= 0) sg[(K) < 0 ? 0 : (K)] += now_ - sp;                                       \
    sp = now_;                                                                              \
  } while (0)
  SEG_STAMP(-1);
#else
#define SEG_STAMP(K)
#endif
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    SEG_STAMP(4);      // 4: the barrier at the end of the previous K-tile (+ loop overhead)
    if (t + 1 < nt) stage(t + 1, buf ^ 1);
    SEG_STAMP(0);      // 0: issue of the 6 LDS-DMAs of K-tile t+1
    const char* tp = smem + buf * STAGE;
    const char* tq = tp + PSUB * SUB_BYTES;
    bf16x8 pf[TM], qf[TNN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int c = prow + i * 16;
      pf[i] = frag_tr(tp + (c >> 7) * SUB_BYTES, c & 127, 0, lane);
    }
#pragma unroll
    for (int j = 0; j < TNN; ++j) {
      const int c = pcol + j * 16;
      qf[j] = frag_tr(tq + (c >> 7) * SUB_BYTES, c & 127, 0, lane);
    }
    SEG_STAMP(1);      // 1: 24 transposing reads issued AND returned (the stamp waits lgkmcnt(0))
    REED_LDS_WAIT();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TNN; ++j)
        acc[i][j] = REED_MFMA_16x16x32(qf[j], pf[i], acc[i][j]);
    if (do_dbias) {
#pragma unroll
      for (int i = 0; i < TM; ++i) accb[i] = REED_MFMA_16x16x32(ones, pf[i], accb[i]);
    }
    SEG_STAMP(2);      // 2: the MFMAs issued
#ifdef REED_CLK_SEG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SEG_STAMP(3);      // 3: K-tile t+1's DMAs landed (what the barrier's vmcnt(0) waits for)
#endif
    __syncthreads();
  }

#ifdef REED_CLK_SEG
  if (lane == 0 && blockIdx.x < 128) {
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) reed_clk_buf_tn[(blockIdx.x * 4 + wave) * 8 + k2] = sg[k2];
    reed_clk_buf_tn[(blockIdx.x * 4 + wave) * 8 + 5] = nt;
  }
#endif
#ifdef REED_CLK_PROBE
  {
    const unsigned long long ck1 = __builtin_amdgcn_s_memtime(), cr1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (tid == 0 && blockIdx.x < 1024) {
      reed_clk_buf_tn[4 * blockIdx.x + 0] = ck1 - ck0;
      reed_clk_buf_tn[4 * blockIdx.x + 1] = cr1 - cr0;
      reed_clk_buf_tn[4 * blockIdx.x + 2] = nt;
      reed_clk_buf_tn[4 * blockIdx.x + 3] = WIDE ? 1 : 0;
    }
  }
#endif
  // fp32 output (slabs / accumulate): pointer-path epilogue, 64 columns at a time
#pragma unroll
  for (int h = 0; h < TNN / 4; ++h) {
    f32x4 part[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) part[i][j] = acc[i][h * 4 + j];
    tile_epilogue_ptr<EPI_F32, TM>(a, part, m0 + prow, n0 + pcol + h * 64, lane, z);
  }
  if (do_dbias && (lane >> 4) == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + prow + i * 16 + (lane & 15);
      if (m < a.M) {
        if (nz > 1) a.dbias[(long)z * a.slab_stride + m] = accb[i][0];
        else if (a.accumulate) a.dbias[m] += accb[i][0];
        else a.dbias[m] = accb[i][0];
      }
    }
  }
}

// one problem's workgroup -> tile map: every XCD a contiguous run of tiles, walked in groups of 4 tile rows x all tile columns
template <bool WIDE>
__device__ __forceinline__ void tn_tile_of(const GemmArgs& a, int bid, int& tm, int& tn) {
  constexpr int BMT = WIDE ? 128 : 256, BNT = WIDE ? 256 : 128;
  const int ntm = (a.M + BMT - 1) / BMT, ntn = a.N / BNT;
  const int nwg = ntm * ntn;
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

template <bool WIDE>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  tn_tile_of<WIDE>(a, blockIdx.x, tm, tn);
  gemm_tn_body<WIDE>(a, smem, tm, tn, blockIdx.y, gridDim.y);
}

// The weight gradients of one transformer block in ONE launch, no split-K: at K = b x 256 tokens a single wgrad has too
// few tiles for the 512 workgroup slots (SiT-XL/2: 162 / 162 / 126 / 45), which the per-GEMM path fills with 3-6 K slices
// written as fp32 slabs and summed by a second kernel (at b = 32 that reduce was 6 % of the step and the slab traffic
// slowed the GEMMs to 490-870 TFLOP/s).  Together the four problems have 495 tiles, 512 with each problem's run padded to
// a multiple of 8 (the XCD interleave): one full round of equal-length K loops writing the gradient arena directly —
// deterministic, and the same at every batch size.
// Workgroup -> tile map (compact = 1): the tiles of all problems form ONE sequence — a 256x128-tile problem row by row
// (column tiles fastest), a 128x256-tile problem column by column — and XCD x (workgroups x, x + 8, ...: the hardware deals
// workgroups round-robin to the 8 XCDs) takes a contiguous 1/8 of it: ~62 tiles = 7 operand blocks of the wide operand x all 9
// blocks of the narrow one, so an operand block is fetched into one or two L2s instead of four or five (PMC).
struct TnGroupArgs {
  GemmArgs a[4];
  int first[5];   // compact = 0: first workgroup of each problem (multiples of 8); compact = 1: first tile in the sequence
  int wide[4];
  int n;
  int compact;
};
__global__ __launch_bounds__(256, 2) void gemm_tn_group_kernel(TnGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int seq = blockIdx.x;
  if (g.compact) {
    seq = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (seq >= g.first[g.n]) return;
  }
  int p = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < g.n && seq >= g.first[i]) p = i;
  const int l = seq - g.first[p];
  const GemmArgs& a = g.a[p];
  const int wide = g.wide[p];
  int tm, tn;
  if (g.compact) {
    if (wide) { const int ntm = (a.M + 127) / 128; tn = l / ntm; tm = l - tn * ntm; }
    else { const int ntn = a.N / 128; tm = l / ntn; tn = l - tm * ntn; }
  } else {
    const int tiles = wide ? ((a.M + 127) / 128) * (a.N / 256) : ((a.M + 255) / 256) * (a.N / 128);
    if (l >= tiles) return;   // padding of the problem's run
    if (wide) tn_tile_of<true>(a, l, tm, tn);
    else tn_tile_of<false>(a, l, tm, tn);
  }
  if (wide) gemm_tn_body<true>(a, smem, tm, tn, 0, 1);
  else gemm_tn_body<false>(a, smem, tm, tn, 0, 1);
}

// (Round 3 also built the grouped weight gradients on a 128x192 tile with THREE workgroups per CU — bit-identical, and 5-8 %
// slower than the kernel above at every batch: 2.00 / 1.07 / 0.266 ms against 1.91 / 1.00 / 0.247 at b = 256 / 128 / 32 — a third wave
// per SIMD does not buy what its smaller tile costs; removed in round 5, DESIGN_HISTORY.md.)

template <bool WIDE>
int launch_tn(const GemmArgs& a, int splits, hipStream_t stream) {
  constexpr int BMT = WIDE ? 128 : 256, BNT = WIDE ? 256 : 128;
  constexpr int LDS = 2 * 3 * SUB_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, BMT) * (a.N / BNT), splits, 1);
  REED_KLAUNCH((gemm_tn_kernel<WIDE>), grid, dim3(256), LDS, stream, a);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

}  // namespace

int reed_num_cus();   // gemm256.hip
int reed_gemm256w_tn_group_launch(int n, const GemmArgs* probs, hipStream_t stream, int* launched);   // gemm256w.hip

// n <= 4 problems dw_i[M_i, N_i] f32 (+)= dy_i[K, M_i]^T x_i[K, N_i] (+ optional dbias_i) sharing the token count K.
// Returns REED_ERR_UNSUPPORTED (nothing launched) when the tiles do not fit one round of workgroup slots.
int reed_gemm_tn_group_launch(int n, const GemmArgs* probs, hipStream_t stream) {
  REED_CHECK_ARG(n >= 1 && n <= 4, "wgrad_group: 1..4 problems, got %d", n);
  {   // 256^2 tiles with four 128x128 waves, one workgroup per CU, when the problems fill one round of those
    int launched = 0;
    const int rc = reed_gemm256w_tn_group_launch(n, probs, stream, &launched);
    if (rc != REED_OK || launched) return rc;
  }
  TnGroupArgs g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  g.compact = 1;   // (0: per-problem runs — the A/B of round 3)
  int at = 0, padded = 0;
  for (int i = 0; i < n; ++i) {
    const GemmArgs& a = probs[i];
    REED_CHECK_ARG(a.N % 128 == 0 && a.M % 16 == 0 && a.K == probs[0].K && a.K > 0,
                   "wgrad_group: problem %d: M=%d N=%d K=%d (N must be a multiple of 128, M of 16, K shared)", i, a.M, a.N, a.K);
    const int tall = cdiv(a.M, 256) * (a.N / 128);
    const int wid = (a.N % 256 == 0) ? cdiv(a.M, 128) * (a.N / 256) : (1 << 30);
    g.a[i] = a;
    g.a[i].ksplit_len = cdiv(a.K, TBK) * TBK;
    g.wide[i] = wid < tall;
    g.first[i] = at;
    at += g.compact ? min(tall, wid) : (min(tall, wid) + 7) & ~7;
    padded += (min(tall, wid) + 7) & ~7;
  }
  g.first[n] = at;
  if (padded > 2 * reed_num_cus()) {   // (the padded count: the planning side, ops.wgrad_group_blocks, counts the same way)
    reed_set_error("wgrad_group: %d workgroups do not fit one round of %d slots", padded, 2 * reed_num_cus());
    return REED_ERR_UNSUPPORTED;
  }
  if (g.compact) at = 8 * cdiv(at, 8);
  constexpr int LDS = 2 * 3 * SUB_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  REED_KLAUNCH(gemm_tn_group_kernel, dim3(at), dim3(256), LDS, stream, g);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

// tile: 1 = 256x128, 2 = 128x256.  EPI_F32 only (weight gradients); the caller has validated the arguments.
int reed_gemm_tn_launch(int tile, GemmArgs a, int splits, hipStream_t stream) {
  if (tile == 2) {
    REED_CHECK_ARG(a.N % 256 == 0, "reed_gemm(TN 128x256): N=%d must be a multiple of 256", a.N);
    return launch_tn<true>(a, splits, stream);
  }
  return launch_tn<false>(a, splits, stream);
}
