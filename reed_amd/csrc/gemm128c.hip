// 128x256x32 bf16 MFMA GEMM built to run TWO workgroups per CU (round 4; NT and NN layouts, the bf16-output fused epilogues).
//
// Why: a 256^2 tile owns its CU (144-160 KiB of LDS, every accumulation register of the four SIMDs), so while its epilogue moves
// 640 KiB (gate + residual: fp32 residual in, fp32 stream + bf16 y out) or 256 KiB (GELU / dGELU: two 16-bit arrays) through the
// CU's vector-memory path — 27 / 11 us at the ~10 B/clk the HBM gives one CU of 256 — the matrix pipe idles, and the K loop of a
// K = 1152 tile is only 24.6 us long (tools/clk_probe.py).  In the driver's table of round 3 the four shapes with those epilogues ran
// at 686-1023 TFLOP/s against 1290 for the plain stores.  Starting every other workgroup late (gemm256w.hip) does not help: what
// is needed is ANOTHER workgroup's K loop on the same SIMDs while this one's epilogue sits in the memory queue.
//
// Form: workgroup = 256 threads = 4 waves (2 x 2), tile 128 (M) x 256 (N), wave 64 x 128 = 4 x 8 MFMA tiles = 128 accumulation
// registers; K step 32 (ONE v_mfma_f32_16x16x32 k-step), three LDS stages of 24 KiB (A 128 x 32, B 256 x 32) = 72 KiB per
// workgroup, two workgroups per CU = 144 KiB; the epilogue's per-wave 4 KiB transposition patches alias stage 0 after the loop.
// Every stage is filled by LDS-DMA (buffer_load ... lds, 16 B per lane, 6 pieces of 1 KiB per wave) two K steps ahead behind ONE
// counted wait per step (s_waitcnt vmcnt(6): the newest stage stays in flight) and one barrier.  Per K step a wave reads 4 + 8
// operand fragments (ds_read_b128 / ds_read_b64_tr_b16) for 32 MFMAs; two co-resident waves per SIMD cover each other's LDS
// round trips, so the loop is left to the compiler's scheduler (the 256^2 four-wave kernel has ONE wave per SIMD and needs every
// read hand-placed).  The price of the smaller tile is 1.5 x the L2 -> LDS bytes per flop of a 256^2 tile.
//
// LDS tile formats (this file only):
//   k-contiguous operand (A always; B for NT): [rows][32 k] = 64-byte rows, 16-byte chunk c of row r at slot c ^ H[(r >> 2) & 3],
//     H = {0, 2, 3, 1}: the 16-lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, ... MI355X_MICROARCH.md) then touch 16 distinct
//     16-byte bank slots (for r & 3 = const the four lanes of a group carry (g ^ H[u]) = {H0, H3, H1 ^ 1, H2 ^ 1} = {0, 1, 3, 2}).
//   k-strided operand (B for NN: W[N(k)][K(n)]): the [32 k][128 n] half-tile format of gemm_common.hpp (256-byte rows, tr_sw
//     swizzle), two half-tiles of 8 KiB for the 256 columns, read with frag_tr.
// Same accumulation order as the other kernels (k ascending in steps of 32 inside one fp32 accumulator): bit-identical results.
#include <stdlib.h>

#include "gemm_common.hpp"

int reed_num_cus();           // gemm256.hip
int reed_gemm_forced_tile();  // gemm.hip

namespace {
using namespace gemm_detail;

constexpr int CBM = 128, CBN = 256, CBK = 32;
constexpr int A_BYTES = CBM * CBK * 2;             // 8 KiB
constexpr int B_BYTES = CBN * CBK * 2;             // 16 KiB
constexpr int STG = A_BYTES + B_BYTES;             // 24 KiB
constexpr int NSTG = 3;
constexpr int LDS_C = NSTG * STG;                  // 72 KiB

__device__ __forceinline__ int hsw(int r) {        // H[(r >> 2) & 3], H = {0, 2, 3, 1}
  const int u = (r >> 2) & 3;
  return (0x78 >> (2 * u)) & 3;                    // 0b01'11'10'00
}
// lane's source offset (bytes from the tile's first row, k0 excluded) for piece `p` of a k-contiguous operand: piece = 16 rows
__device__ __forceinline__ int voff_kc(int p, int lane, long ld) {
  const int r = p * 16 + (lane >> 2), slot = lane & 3;
  return (int)(((long)r * ld + ((slot ^ hsw(r)) * 8)) * 2);
}
// k-strided operand, one 128-column half-tile of 32 k-rows (8 KiB = 8 pieces of 4 k-rows): the format frag_tr reads
__device__ __forceinline__ int voff_ks(int p, int lane, long ld) {
  const int L = p * 64 + lane, r = L >> 4, chp = L & 15;
  return (int)(((long)r * ld + ((chp ^ tr_sw(r)) * 8)) * 2);
}
// lane (i, g): X[rowbase + i][8 g .. 8 g + 7] of a 64-byte-row tile
__device__ __forceinline__ bf16x8 frag_kc(const char* tile, int rowbase, int lane) {
  const int i = lane & 15, g = lane >> 4;
  return *(const bf16x8*)(tile + (rowbase + i) * 64 + ((g ^ hsw(i)) << 4));   // (rowbase is a multiple of 16: hsw(row) = hsw(i))
}

template <int LAY, int EPI>
__global__ __launch_bounds__(256, 2) void gemm128c_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- block -> tile: XCD-contiguous runs, groups of GM tile rows share their B panels in the XCD's L2 ----
  const int ntm = (a.M + CBM - 1) / CBM, ntn = (a.N + CBN - 1) / CBN;
  const int nwg = ntm * ntn;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int GM = (a.tile_gm & 0xFF) > 0 ? (a.tile_gm & 0xFF) : 8;
  const bool dbg_nodma = (a.tile_gm & 0x100) != 0, dbg_nomfma = (a.tile_gm & 0x200) != 0;   // REED_GEMM128C_DBG (diagnosis)
  const int per_group = GM * ntn;
  const int group = bid / per_group, first_m = group * GM;
  const int gs = min(ntm - first_m, GM);
  const int tm = first_m + (bid % per_group) % gs;
  const int tn = (bid % per_group) / gs;
  const int m0 = tm * CBM, n0 = tn * CBN;
  const int nt = a.K / CBK;

  __amdgpu_buffer_rsrc_t rsP = make_rsrc(a.P + (long)m0 * a.ldp, ((long)(a.M - m0) * a.ldp) * 2), rsQ;
  if constexpr (LAY == LAY_NT) rsQ = make_rsrc(a.Q + (long)n0 * a.ldq, ((long)(a.N - n0) * a.ldq) * 2);
  else rsQ = make_rsrc(a.Q + n0, ((long)a.K * a.ldq - n0) * 2);

  // this wave's six pieces of a stage: A pieces 0..7 (waves 0..3 take 2 each), B pieces 0..15 (4 each); per-lane offsets are loop
  // invariant, the K step goes through the scalar offset
  int va[2], vb[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) va[j] = voff_kc(wave * 2 + j, lane, a.ldp);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = wave * 4 + j;
    if constexpr (LAY == LAY_NT) vb[j] = voff_kc(p, lane, a.ldq);
    else {
      // half-tile h = p >> 3 (128 columns each), piece p & 7 inside it; columns past N fall outside the descriptor only through
      // the row bound, so they are pushed out explicitly (ragged last column tile)
      const int h = p >> 3;
      const int col = n0 + h * 128 + ((((p & 7) * 64 + lane) & 15) ^ tr_sw(((p & 7) * 64 + lane) >> 4)) * 8;
      vb[j] = col < a.N ? voff_ks(p & 7, lane, a.ldq) + h * 256 : EPI_OOB;
    }
  }
  const int kstepA = CBK * 2;
  const int kstepB = (LAY == LAY_NT) ? CBK * 2 : (int)(CBK * a.ldq * 2);
  auto stage = [&](int t, int buf) {
    char* sa = smem + buf * STG;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_ptr_t)(sa + (wave * 2 + j) * 1024), 16, va[j], t * kstepA, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(sb + (wave * 4 + j) * 1024), 16, vb[j], t * kstepB, 0, 0);
  };

  f32x4 acc[2][4][4];   // [64-column strip][row tile][column tile]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[s][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // stages t and t + 1 in flight at the top of step t; the wait leaves the newer one (6 pieces of this wave) outstanding
  stage(0, 0);
  if (nt > 1) stage(1, 1);
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt && !dbg_nodma) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();     // stage t landed for every wave; every wave is done reading stage t - 1
    asm volatile("" ::: "memory");
    if (t + 2 < nt && !(dbg_nodma && t >= 2)) stage(t + 2, (t + 2) % NSTG);
    const char* sa = smem + (t % NSTG) * STG;
    const char* sb = sa + A_BYTES;
    bf16x8 pf[4], qf[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = frag_kc(sa, wm * 64 + i * 16, lane);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (LAY == LAY_NT) qf[j] = frag_kc(sb, wn * 128 + j * 16, lane);
      else qf[j] = frag_tr(sb + wn * 8192, j * 16, 0, lane);
    }
    if constexpr (LAY != LAY_NT) REED_LDS_WAIT();   // asm transposing reads: the compiler does not count them
    if (dbg_nomfma) {   // diagnosis: the operand stream alone (the fragments are consumed by one MFMA)
      acc[0][0][0] = REED_MFMA_16x16x32(qf[0], pf[0], acc[0][0][0]);
      continue;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j >> 2][i][j & 3] = REED_MFMA_16x16x32(qf[j], pf[i], acc[j >> 2][i][j & 3]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();       // every wave is done with the stages: the patches alias stage 0
  asm volatile("" ::: "memory");
  char* patch = smem + wave * EPI_STAGE_BYTES;
  tile_epilogue<EPI, 4, 1>(a, acc[0], m0, wm * 64, n0 + wn * 128, lane, 0, patch);
  tile_epilogue<EPI, 4, 1>(a, acc[1], m0, wm * 64, n0 + wn * 128 + 64, lane, 0, patch);
}

template <int LAY, int EPI>
int launch128c(const GemmArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm128c_kernel<LAY, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_C);
    if (e != hipSuccess) { reed_set_error("gemm128c: cannot reserve LDS: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  GemmArgs b = a;
  static int gm = -1;
  if (gm < 0) { const char* e = getenv("REED_GEMM128C_GM"); gm = e ? atoi(e) : 8; }
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("REED_GEMM128C_DBG"); dbg = e ? atoi(e) : 0; }   // 1: no DMA after the first stages, 2: no MFMAs
  b.tile_gm = gm | (dbg << 8);
  const int ntm = cdiv(a.M, CBM), ntn = cdiv(a.N, CBN);
  REED_KLAUNCH((gemm128c_kernel<LAY, EPI>), dim3(ntm * ntn), dim3(256), LDS_C, stream, b);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

template <int LAY>
int dispatch128c(int epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_BF16: return launch128c<LAY, EPI_BF16>(a, s);
    case EPI_GELU:
      if constexpr (LAY == LAY_NT) return launch128c<LAY, EPI_GELU>(a, s);
      break;
    case EPI_GATE_RES:
      if constexpr (LAY == LAY_NT) return launch128c<LAY, EPI_GATE_RES>(a, s);
      break;
    case EPI_DGELU:
      if constexpr (LAY == LAY_NN) return launch128c<LAY, EPI_DGELU>(a, s);
      break;
  }
  reed_set_error("gemm128c: epilogue %d is not built for this layout", epi);
  return REED_ERR_UNSUPPORTED;
}

}  // namespace

// NT: plain / GELU / gate + residual; NN: plain / dGELU.  N a multiple of 128 (a ragged last 128 columns of a 256-column tile are
// multiplied as zeros), K a multiple of 32 with at least two steps, no split-K.
bool reed_gemm128c_eligible(int layout, int epi, const GemmArgs& a, int splits) {
  const bool epi_ok = layout == LAY_NT ? (epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_GATE_RES)
                                       : layout == LAY_NN ? (epi == EPI_BF16 || epi == EPI_DGELU) : false;
  return epi_ok && splits <= 1 && a.K % CBK == 0 && a.K >= 2 * CBK && a.N % 128 == 0;
}

int reed_gemm128c_launch(int layout, int epi, GemmArgs a, hipStream_t stream) {
  if (layout == LAY_NT) return dispatch128c<LAY_NT>(epi, a, stream);
  return dispatch128c<LAY_NN>(epi, a, stream);
}
