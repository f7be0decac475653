// Skinny NT GEMM for gfx950: y = epilogue(x W^T) for a FEW rows of x — the tail launch of the dispatcher's ragged-M split (gemm.hip):
// the 64 rows that 64 x 257 ViT tokens leave beyond the last full 256-row tile (frozen towers).
//
// Why its own kernel (round 4, profiles/r4_tower64_kernel_stats.csv): on the 128^2 kernel those 64 rows made 4-32 workgroups that each
// streamed a 128-column strip of W over the whole K through a two-stage pipeline — 38 us for fc2's tail (K = 4096), 113 us per layer
// over the four GEMMs, 16 % of DINOv2 ViT-L's forward at batch 64 — because a tail is bound by what ONE CU streams from L2 (about
// 70 GB/s: MI355X_MICROARCH.md, indexed rows) times the CUs it runs on, not by arithmetic.  So: one WAVE per workgroup, a tile of
// 16 rows x 64 columns (the unit the fused epilogues of gemm_common.hpp work on), as many workgroups as there are such tiles (64
// for a 64 x 1024 output, 256 for 64 x 4096), and a six-stage LDS ring filled by LDS-DMA five K-tiles ahead (50 KiB in flight per
// wave) behind ONE counted wait — every K-tile, past the end too, issues its ten pieces (out-of-range ones read nothing), so the
// count never changes.  No barrier anywhere (a single wave; LDS executes its instructions in order).  The K order per output
// element is the one of every other kernel (K-tiles of 64 in sequence, k-step 0 then 1): results are bit-identical to theirs.
#include <stdlib.h>

#include "gemm_common.hpp"

namespace {
using namespace gemm_detail;

constexpr int SK = 64;                       // K-tile
constexpr int SA_BYTES = 16 * SK * 2;        // 2 KiB: 16 rows of x
constexpr int SB_BYTES = 64 * SK * 2;        // 8 KiB: 64 rows of W
constexpr int SSTAGE = SA_BYTES + SB_BYTES;  // 10 pieces of 1 KiB
constexpr int NSTAGE = 6;                    // 5 K-tiles = 50 pieces in flight (a counted wait can leave 63)
constexpr int SLDS = NSTAGE * SSTAGE + EPI_STAGE_BYTES;

template <int EPI>
__global__ __launch_bounds__(64) void gemm_skinny_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int ntr = (a.M + 15) / 16, ntn = a.N / 64;
  const int nwg = ntr * ntn;
  // workgroups of one XCD (blockIdx & 7) take a contiguous run of tiles, row tiles of one column strip adjacent: the strip of W is
  // fetched into one L2
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tr = bid % ntr, tn = bid / ntr;
  const int m0 = tr * 16, n0 = tn * 64;
  const int nt = a.K / SK;

  const __amdgpu_buffer_rsrc_t rsP = make_rsrc(a.P + (long)m0 * a.ldp, ((long)(a.M - m0 - 1) * a.ldp + a.K) * 2);
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(a.Q + (long)n0 * a.ldq, ((long)(a.N - n0 - 1) * a.ldq + a.K) * 2);
  // piece i of a [rows][64] k-contiguous tile: rows 8 i .. 8 i + 7, 16-byte chunk (lane & 7) ^ ((row >> 1) & 7) of the row
  const int pr = lane >> 3, pc = lane & 7;
  int voA[2], voB[8];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * i + pr;
    voA[i] = (int)(((long)r * a.ldp + (pc ^ ((r >> 1) & 7)) * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = 8 * i + pr;
    voB[i] = (int)(((long)r * a.ldq + (pc ^ ((r >> 1) & 7)) * 8) * 2);
  }
  // K-tile t into ring slot t % NSTAGE; past the last K-tile the k offset lies beyond every row's K elements only for the LAST row of
  // the descriptor — so those pieces get an explicit out-of-range offset instead (same instruction count, nothing read)
  constexpr int OOB = (int)0xFFFFFFF0u;   // beyond every descriptor (make_rsrc clamps the range to 2^32 - 1 bytes)
  auto issue = [&](int t) {
    char* sl = smem + (t % NSTAGE) * SSTAGE;
    const bool live = t < nt;
    const int k2 = t * SK * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_ptr_t)(sl + i * 1024), 16, live ? voA[i] + k2 : OOB, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(sl + SA_BYTES + i * 1024), 16, live ? voB[i] + k2 : OOB, 0, 0, 0);
  };

  f32x4 acc[1][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[0][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t) issue(t);
  for (int t = 0; t < nt; ++t) {
    issue(t + NSTAGE - 1);
    // K-tile t has landed once at most the (NSTAGE - 1) x 10 pieces issued after it are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 1) * 10) : "memory");
    const char* ta = smem + (t % NSTAGE) * SSTAGE;
    const char* tb = ta + SA_BYTES;
    bf16x8 pf[2], qf[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      pf[ks] = frag_row(ta, 0, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) qf[ks][j] = frag_row(tb, 16 * j, ks, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[0][j] = REED_MFMA_16x16x32(qf[ks][j], pf[ks], acc[0][j]);
    // the slot is refilled by issue(t + NSTAGE) at the top of the next iteration: its fragment reads above have returned by then
    // (the MFMAs consumed them)
    asm volatile("" ::: "memory");
  }
  // the NSTAGE - 1 dummy fills of the last iterations write nothing; the epilogue's patch is its own region
  tile_epilogue<EPI, 1>(a, acc, m0, 0, n0, lane, 0, smem + NSTAGE * SSTAGE);
}

template <int EPI>
int launch_skinny(const GemmArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_skinny_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, SLDS);
    if (e != hipSuccess) { reed_set_error("gemm_skinny: cannot reserve LDS: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  const int nwg = cdiv(a.M, 16) * (a.N / 64);
  REED_KLAUNCH((gemm_skinny_kernel<EPI>), dim3(nwg), dim3(64), SLDS, stream, a);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

}  // namespace

// NT, the row-free bf16 / fp32-residual epilogues of a frozen tower's forward, K a multiple of 64, N of 64, no split-K
bool reed_gemm_skinny_eligible(int layout, int epi, const GemmArgs& a, int splits) {
  const bool epi_ok = epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_SILU || epi == EPI_QGELU || epi == EPI_GELU_ERF ||
                      epi == EPI_GELU_G || epi == EPI_SILU_G ||
                      epi == EPI_RES_BF16 || epi == EPI_LS_RES || epi == EPI_GATE_RES;
  return layout == LAY_NT && epi_ok && splits <= 1 && a.K % SK == 0 && a.K >= SK && a.N % 64 == 0 && a.M >= 1 &&
         (long)cdiv(a.M, 16) * (a.N / 64) < (1l << 30);
}

int reed_gemm_skinny_launch(int epi, GemmArgs a, hipStream_t stream) {
  switch (epi) {
    case EPI_BF16: return launch_skinny<EPI_BF16>(a, stream);
    case EPI_GELU: return launch_skinny<EPI_GELU>(a, stream);
    case EPI_SILU: return launch_skinny<EPI_SILU>(a, stream);
    case EPI_GELU_G: return launch_skinny<EPI_GELU_G>(a, stream);
    case EPI_SILU_G: return launch_skinny<EPI_SILU_G>(a, stream);
    case EPI_QGELU: return launch_skinny<EPI_QGELU>(a, stream);
    case EPI_GELU_ERF: return launch_skinny<EPI_GELU_ERF>(a, stream);
    case EPI_RES_BF16: return launch_skinny<EPI_RES_BF16>(a, stream);
    case EPI_LS_RES: return launch_skinny<EPI_LS_RES>(a, stream);
    case EPI_GATE_RES: return launch_skinny<EPI_GATE_RES>(a, stream);
  }
  reed_set_error("gemm_skinny: epilogue %d not built", epi);
  return REED_ERR_UNSUPPORTED;
}
