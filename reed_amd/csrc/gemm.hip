// bf16 MFMA GEMM for gfx950 with fused epilogues — the dense contraction behind every
// nn.Linear on the SiT hot path (reference: image/models/sit.py:17-24,114-124,126-129,146-150
// and timm Attention/Mlp linears; backward = autograd of the same).
//
//   C[M,N] (+)= sum_k P(m,k) * Q(n,k)
//
// Three operand layouts, all row-major in HBM, no transposed copies anywhere:
//   NT  P = A[M,K]  (k contiguous)   Q = B[N,K]  (k contiguous)   forward:  y = x W^T
//   NN  P = A[M,K]  (k contiguous)   Q = B[K,N]  (k strided)      dgrad:    dx = dy W
//   TN  P = A[K,M]  (k strided)      Q = B[K,N]  (k strided)      wgrad:    dW = dy^T x
// k-contiguous operands are staged as [128][64] LDS tiles (128-B rows, XOR-swizzled 16-B chunks)
// and read with ds_read_b128; k-strided operands are staged as [64][128] tiles (256-B rows,
// swizzled) and read with the gfx950 transposing read ds_read_b64_tr_b16.  Staging is
// buffer_load_dwordx4 ... lds (LDS-DMA, 16 B/lane) with hardware bounds checking, so ragged M
// (and ragged K for TN) need no masking code: out-of-range rows read as zero.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16.  The MFMA is issued "swapped" (Q fragment as A operand, P fragment
// as B operand) so each lane ends up holding 4 consecutive n for one m: 8-B bf16 / 16-B fp32
// epilogue accesses.  LDS is double buffered (64 KiB), one barrier per K step.
#include <stdlib.h>

#include "gemm_common.hpp"

int reed_gemm256_launch(int layout, int epi, GemmArgs a, int splits, hipStream_t stream);
int reed_gemm_tn_launch(int tile, GemmArgs a, int splits, hipStream_t stream);   // gemm_tn.hip
bool reed_gemm256_preferred(int layout, int epi, const GemmArgs& a, int splits);
bool reed_gemm144_eligible(int layout, int epi, const GemmArgs& a, int splits);              // gemm144.hip
bool reed_gemm144_preferred(int layout, int epi, const GemmArgs& a, int splits);
int reed_gemm144_launch(int layout, int epi, GemmArgs a, hipStream_t stream);
bool reed_gemm288_eligible(int layout, int epi, const GemmArgs& a, int splits);              // gemm288.hip: 256x288 tiles (NT)
bool reed_gemm288_preferred(int layout, int epi, const GemmArgs& a, int splits);
int reed_gemm288_launch(int epi, GemmArgs a, hipStream_t stream);
bool reed_gemm256w_eligible(int layout, int epi, const GemmArgs& a, int splits);   // gemm256w.hip: 4 waves x 128x128
int reed_gemm256w_launch(int layout, int epi, GemmArgs a, hipStream_t stream);
int reed_num_cus();   // gemm256.hip
double reed_gemm256_rate();
bool reed_gemm_skinny_eligible(int layout, int epi, const GemmArgs& a, int splits);   // gemm_skinny.hip: 16 x 64 tiles, one wave each
int reed_gemm_skinny_launch(int epi, GemmArgs a, hipStream_t stream);

namespace {
using namespace gemm_detail;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 16384;          // one operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;

// ---- staging ----------------------------------------------------------------
// k-contiguous operand: tile rows [0,128) x k [k0,k0+64); rsrc is based at the tile's first row.
__device__ __forceinline__ void stage_row(__amdgpu_buffer_rsrc_t rs, char* tile, long ld, int k0,
                                          int tid, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int L = i * 256 + tid;
    int r = L >> 3, cp = L & 7;
    int c = cp ^ ((r >> 1) & 7);
    int voff = (int)(((long)r * ld + k0 + c * 8) * 2);
    char* dst = tile + (i * 256 + wave * 64) * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
  }
}
// k-strided operand: k rows [k0,k0+64) x cols [0,128); rsrc is based at column c0 of row 0.
__device__ __forceinline__ void stage_tr(__amdgpu_buffer_rsrc_t rs, char* tile, long ld, int k0,
                                         int tid, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int L = i * 256 + tid;
    int r = L >> 4, chp = L & 15;
    int ch = chp ^ tr_sw(r);
    int voff = (int)(((long)(k0 + r) * ld + ch * 8) * 2);
    char* dst = tile + (i * 256 + wave * 64) * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
  }
}

template <int LAY, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- block -> tile (XCD-aware, grouped along M) ----
  const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
  const int nwg = ntm * ntn;
  int bid = blockIdx.x;
  {
    int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int GM = 8;
  const int per_group = GM * ntn;
  const int group = bid / per_group, first_m = group * GM;
  const int gs = min(ntm - first_m, GM);
  const int tm = first_m + (bid % per_group) % gs;
  const int tn = (bid % per_group) / gs;
  const int z = blockIdx.y;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = z * a.ksplit_len;
  const int kend = min(a.K, kbeg + a.ksplit_len);
  const int nt = (kend - kbeg + BK - 1) / BK;

  // ---- buffer descriptors based at this block's tile origin ----
  __amdgpu_buffer_rsrc_t rsP, rsQ;
  if constexpr (LAY == LAY_TN) {
    // P = A[K, M]: rows are k; records end at row kend
    rsP = make_rsrc(a.P + m0, ((long)kend * a.ldp - m0) * 2);
  } else {
    rsP = make_rsrc(a.P + (long)m0 * a.ldp, ((long)(a.M - m0) * a.ldp) * 2);
  }
  if constexpr (LAY == LAY_NT) {
    rsQ = make_rsrc(a.Q + (long)n0 * a.ldq, ((long)(a.N - n0) * a.ldq) * 2);
  } else {
    rsQ = make_rsrc(a.Q + n0, ((long)kend * a.ldq - n0) * 2);
  }

  auto stage = [&](int t, int buf) {
    char* tp = smem + buf * STAGE_BYTES;
    char* tq = tp + TILE_BYTES;
    int k0 = kbeg + t * BK;
    if constexpr (LAY == LAY_TN) stage_tr(rsP, tp, a.ldp, k0, tid, wave);
    else stage_row(rsP, tp, a.ldp, k0, tid, wave);
    if constexpr (LAY == LAY_NT) stage_row(rsQ, tq, a.ldq, k0, tid, wave);
    else stage_tr(rsQ, tq, a.ldq, k0, tid, wave);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_dbias = (LAY == LAY_TN) && a.dbias != nullptr && tn == 0 && wn == 0;
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;

  if (nt > 0) stage(0, 0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) stage(t + 1, buf ^ 1);
    const char* tp = smem + buf * STAGE_BYTES;
    const char* tq = tp + TILE_BYTES;
    bf16x8 pf[2][4], qf[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (LAY == LAY_TN) pf[ks][i] = frag_tr(tp, wm * 64 + i * 16, ks, lane);
        else pf[ks][i] = frag_row(tp, wm * 64 + i * 16, ks, lane);
        if constexpr (LAY == LAY_NT) qf[ks][i] = frag_row(tq, wn * 64 + i * 16, ks, lane);
        else qf[ks][i] = frag_tr(tq, wn * 64 + i * 16, ks, lane);
      }
    if constexpr (LAY != LAY_NT) REED_LDS_WAIT();  // asm transposing reads: the compiler does not count them
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = REED_MFMA_16x16x32(qf[ks][j], pf[ks][i], acc[i][j]);
      if constexpr (LAY == LAY_TN) {
        if (do_dbias) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            accb[i] = REED_MFMA_16x16x32(ones, pf[ks][i], accb[i]);
        }
      }
    }
    __syncthreads();
  }

  // (the K loop ended on a __syncthreads: the tile buffers are free, each wave stages through its own 4 KiB)
  tile_epilogue<EPI, 4>(a, acc, m0, wm * 64, n0 + wn * 64, lane, z, smem + wave * EPI_STAGE_BYTES);
  if constexpr (LAY == LAY_TN) {
    if (do_dbias && (lane >> 4) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m < a.M) {
          if (gridDim.y > 1) a.dbias[(long)z * a.slab_stride + m] = accb[i][0];  // split-K: per-slice slab (C's stride)
          else if (a.accumulate) a.dbias[m] += accb[i][0];
          else a.dbias[m] = accb[i][0];
        }
      }
    }
  }
}

template <int LAY, int EPI>
int launch(const GemmArgs& a, int splits, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<LAY, EPI>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    attr_set = true;
  }
  const int ntm = cdiv(a.M, BM), ntn = a.N / BN;
  dim3 grid(ntm * ntn, splits, 1);
  REED_KLAUNCH((gemm_kernel<LAY, EPI>), grid, dim3(256), 2 * STAGE_BYTES, stream, a);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

template <int LAY>
int dispatch_epi(int epi, const GemmArgs& a, int splits, hipStream_t s) {
  switch (epi) {
    case EPI_BF16: return launch<LAY, EPI_BF16>(a, splits, s);
    case EPI_GELU: return launch<LAY, EPI_GELU>(a, splits, s);
    case EPI_SILU: return launch<LAY, EPI_SILU>(a, splits, s);
    case EPI_GATE_RES: return launch<LAY, EPI_GATE_RES>(a, splits, s);
    case EPI_DGELU: return launch<LAY, EPI_DGELU>(a, splits, s);
    case EPI_DSILU: return launch<LAY, EPI_DSILU>(a, splits, s);
    case EPI_GELU_G: return launch<LAY, EPI_GELU_G>(a, splits, s);
    case EPI_SILU_G: return launch<LAY, EPI_SILU_G>(a, splits, s);
    case EPI_MUL: return launch<LAY, EPI_MUL>(a, splits, s);
    case EPI_F32: return launch<LAY, EPI_F32>(a, splits, s);
    case EPI_ADDF32_RB: return launch<LAY, EPI_ADDF32_RB>(a, splits, s);
    case EPI_ATOMIC_F32: return launch<LAY, EPI_ATOMIC_F32>(a, splits, s);
    case EPI_QGELU: return launch<LAY, EPI_QGELU>(a, splits, s);
    case EPI_GELU_ERF: return launch<LAY, EPI_GELU_ERF>(a, splits, s);
    case EPI_RES_BF16: return launch<LAY, EPI_RES_BF16>(a, splits, s);
    case EPI_LS_RES:
      if constexpr (LAY == LAY_NT) return launch<LAY, EPI_LS_RES>(a, splits, s);
      break;
  }
  reed_set_error("reed_gemm: unknown epilogue %d", epi);
  return REED_ERR_ARG;
}

}  // namespace

static const bool g_colsplit = getenv("REED_GEMM_COLSPLIT") && atoi(getenv("REED_GEMM_COLSPLIT")) != 0;   // =1: the column split (A/B runs; off by default)
static int g_force_tile = 0;  // 0 = heuristic, 64 / 128 / 256 / 144 / 288 / 257 / 258 = force where the shape allows, 259 = heuristic + column split (tests, A/B timing)
extern "C" int reed_gemm_force_tile(int tile) { g_force_tile = tile; return 0; }
int reed_gemm_forced_tile() { return g_force_tile; }

int reed_gemm_launch(int layout, int epi, GemmArgs a, int splits, hipStream_t stream) {
  REED_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "reed_gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  // epilogue 13 (store + per-head dot products with R) exists in the four-wave 256^2 kernel only: the kernel is selected as for
  // the plain store, and where that selection is another kernel the call returns 1002 without launching (the caller then
  // stores plainly and lets reed_attention_bwd_ws form delta itself)
  const int want = epi;
  if (epi == EPI_BF16_DOT) {
    REED_CHECK_ARG((layout == LAY_NN || layout == LAY_NT) && a.R && a.C2 && (a.rows_per_gate == 64 || a.rows_per_gate == 72) &&
                       a.N % a.rows_per_gate == 0 && a.N % 64 == 0 && splits <= 1,
                   "reed_gemm(epilogue 13): NN or NT, R and C2 given, rows_per_gate = head_dim 64 or 72 dividing N");
    epi = EPI_BF16;
  }
#define REED_ONLY_PLAIN()                                                                             \
  do {                                                                                                \
    if (want == EPI_BF16_DOT) {                                                                       \
      reed_set_error("reed_gemm(epilogue 13): this shape runs on a kernel without it (use epilogue 0)"); \
      return REED_ERR_UNSUPPORTED;                                                                    \
    }                                                                                                 \
  } while (0)
  const bool can144 = reed_gemm144_eligible(layout, epi, a, splits);
  REED_CHECK_ARG(a.N % BN == 0 || can144, "reed_gemm: N=%d must be a multiple of %d (or, NT / NN with a bf16-output epilogue, of 144)", a.N, BN);
  REED_CHECK_ARG(a.ldp % 8 == 0 && a.ldq % 8 == 0, "reed_gemm: leading dims must be multiples of 8 elements");
  REED_CHECK_ARG(a.ldc >= 0 && a.ldc < (1 << 20) && a.ldc2 >= 0 && a.ldc2 < (1 << 20) && a.ldr >= 0 && a.ldr < (1 << 20) &&
                     a.ldc % 8 == 0 && a.ldc2 % 8 == 0 && a.ldr % 8 == 0,
                 "reed_gemm: output/residual leading dims must be multiples of 8 below 2^20 (32-bit tile offsets)");
  REED_CHECK_ARG(((uintptr_t)a.P % 16) == 0 && ((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.C % 16) == 0,
                 "reed_gemm: operands must be 16-byte aligned");
  const int tn_tile = layout == LAY_TN_TALL ? 1 : layout == LAY_TN_WIDE ? 2 : 0;   // gemm_tn.hip's 256x128 / 128x256 tiles
  if (tn_tile) {
    REED_CHECK_ARG(epi == EPI_F32, "reed_gemm(TN 256x128 / 128x256): fp32 (weight-gradient) epilogue only");
    layout = LAY_TN;
  }
  if (layout == LAY_TN) {
    REED_CHECK_ARG(a.M % BM == 0, "reed_gemm(TN): M=%d must be a multiple of %d", a.M, BM);
  } else {
    REED_CHECK_ARG(a.K % BK == 0, "reed_gemm(NT/NN): K=%d must be a multiple of %d", a.K, BK);
  }
  if (splits < 1) splits = 1;
  // K per split: multiple of BK
  int ksteps = cdiv(a.K, BK);
  int per = cdiv(ksteps, splits);
  splits = cdiv(ksteps, per);
  a.ksplit_len = per * BK;
  if (splits > 1) {
    REED_CHECK_ARG(epi == EPI_ATOMIC_F32 || (epi == EPI_F32 && a.slab_stride > 0),
                   "reed_gemm: split-K needs the atomic or slab fp32 epilogue");
  }
  if (tn_tile) return reed_gemm_tn_launch(tn_tile, a, splits, stream);
  // Ragged-M split (round 4).  M = B * 257 tokens of a ViT tower (256 patches + CLS) is 64.25 tile rows at B = 64: the 65th row
  // tile (64 live rows) costs every GEMM of the tower a whole extra round of the chip — 65 x 16 = 1040 tiles of the fc1 GEMM are
  // 5 rounds of 256 CUs where 64 x 16 = 1024 are exactly 4 (qkv 4 -> 3, proj and fc2 2 -> 1).  Where dropping the ragged row
  // tile saves a round, the full rows go out as one launch and the <= 128 tail rows as a second, small one (the same kernels
  // on offset pointers: bit-identical results; epilogues whose row index carries meaning — the per-sample gate, the head-dot
  // slots — are left alone).
  {
    const bool epi_rows_free = epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_SILU || epi == EPI_QGELU || epi == EPI_GELU_ERF ||
                               epi == EPI_RES_BF16 || epi == EPI_LS_RES || epi == EPI_DGELU || epi == EPI_DSILU ||
                               epi == EPI_GELU_G || epi == EPI_SILU_G || epi == EPI_MUL;
    const int r = a.M % 256, mfull = a.M - r;
    if (g_force_tile == 0 && want != EPI_BF16_DOT && epi_rows_free && splits <= 1 && (layout == LAY_NT || layout == LAY_NN) &&
        r > 0 && r <= 128 && mfull >= 2048) {
      const int ncu = reed_num_cus(), ntn = cdiv(a.N, 256), rows = mfull / 256;
      if (cdiv((long)rows * ntn, ncu) < cdiv((long)(rows + 1) * ntn, ncu)) {
        const int cb = epi == EPI_LS_RES ? 4 : 2, rb = epi == EPI_LS_RES ? 4 : 2;   // bytes per element of C and R
        GemmArgs m = a, t = a;
        m.M = mfull;
        t.M = r;
        t.P = a.P + (long)mfull * a.ldp;
        if (a.C) t.C = (char*)a.C + (long)mfull * a.ldc * cb;
        if (a.C2) t.C2 = (char*)a.C2 + (long)mfull * a.ldc2 * 2;
        if (a.R) t.R = (const char*)a.R + (long)mfull * a.ldr * rb;
        const int rc = reed_gemm_launch(layout, epi, m, splits, stream);
        if (rc != REED_OK) return rc;
        // the tail: a few rows against the whole weight matrix — bound by how many CUs stream it (gemm_skinny.hip: 16 x 64 tiles,
        // one wave each)
        if (reed_gemm_skinny_eligible(layout, epi, t, splits)) return reed_gemm_skinny_launch(epi, t, stream);
        return reed_gemm_launch(layout, epi, t, splits, stream);
      }
    }
  }
  // Column split (round 6; OFF by default: REED_GEMM_COLSPLIT=1 or force_tile 259).  M = 8192 tokens (b = 32 per GPU) x N = 4608 (fc1
  // forward, the fc2 input gradient) is 32 x 18 = 576 tiles of 256^2 = 2.25 rounds of 256 CUs: three rounds on the 256^2 kernels,
  // four on 256x144 tiles (what the selection below takes).  Where a leading block of tile COLUMNS fills whole rounds exactly, that
  // block goes out on the four-wave 256^2 kernel and the remaining columns as a second launch through the ordinary selection (here
  // 512 columns = 256 tiles of 128^2, one per CU): the same kernels on offset pointers, every element formed by the same products
  // in the same order (bit-identical: tests/test_gemm_gpu.py).  Measured (profiles/r6_column_split.txt): fc1 forward alone 98 -> 94
  // us, the fc2 input gradient 98 -> 96, and the b = 32 step EQUAL (957.4 / 959.0 / 954.5 against 959.0 / 953.2 / 958.1 images/s,
  // alternating on one box) — the second launch's prologue and epilogue eat what the saved round gives.  Kept as a switch, not
  // as the default: it is the cheap stand-in for the 256x288 tile (two rounds of one kernel) that VERDICT round 5 asks for, and
  // it bounds what that tile could give from below.
  {
    const bool epi_cols_free = epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_SILU || epi == EPI_QGELU || epi == EPI_GELU_ERF ||
                               epi == EPI_RES_BF16 || epi == EPI_GATE_RES || epi == EPI_DGELU || epi == EPI_DSILU ||
                               epi == EPI_GELU_G || epi == EPI_SILU_G || epi == EPI_MUL;
    if (((g_force_tile == 0 && g_colsplit) || g_force_tile == 259) && want != EPI_BF16_DOT && epi_cols_free && splits <= 1 &&
        (layout == LAY_NT || layout == LAY_NN) && a.N % 256 == 0 && a.K >= 256) {
      const int ncu = reed_num_cus();
      const long tm = cdiv(a.M, 256), tn = a.N / 256;
      const long full = tm * tn / ncu, rem = tm * tn % ncu;
      if (full >= 1 && rem > 0 && (full * ncu) % tm == 0) {
        const long tn1 = full * ncu / tm;
        GemmArgs h = a, t = a;
        h.N = (int)(tn1 * 256);
        t.N = a.N - h.N;
        if (reed_gemm256w_eligible(layout, epi, h, splits)) {
          const double r256 = 4.0 / reed_gemm256_rate();
          // the tail through the selection's own models: 256x144 tiles, 256^2 tiles in whole rounds, 128^2 tiles two per CU
          const long t128 = cdiv(a.M, 128) * (long)(t.N / 128);
          double ctail = (double)cdiv(t128, 2L * ncu) * 2.0;
          ctail = fmin(ctail, (double)cdiv(tm * (tn - tn1), (long)ncu) * r256);
          if (t.N % 144 == 0) ctail = fmin(ctail, (double)cdiv(tm * (t.N / 144), (long)ncu) * 2.25 / 0.92);
          const double csplit = (double)full * r256 + ctail;
          double cone = (double)(full + 1) * r256;                                                       // 256^2, whole rounds
          cone = fmin(cone, (double)cdiv(cdiv(a.M, 128) * (long)(a.N / 128), 2L * ncu) * 2.0);          // 128^2
          if (a.N % 144 == 0) cone = fmin(cone, (double)cdiv(tm * (a.N / 144), (long)ncu) * 2.25 / 0.92); // 256x144
          if (csplit < 0.97 * cone) {
            const int cb = epi == EPI_GATE_RES ? 4 : 2, rb = epi == EPI_GATE_RES ? 4 : 2;   // bytes per element of C and R
            t.Q = layout == LAY_NT ? a.Q + (long)h.N * a.ldq : a.Q + h.N;
            if (a.C) t.C = (char*)a.C + (long)h.N * cb;
            if (a.C2) t.C2 = (char*)a.C2 + (long)h.N * 2;
            if (a.R) t.R = (const char*)a.R + (long)h.N * rb;
            if (a.bias) t.bias = a.bias + h.N;
            if (a.gate) t.gate = a.gate + h.N;
            const int rc = reed_gemm256w_launch(layout, epi, h, stream);
            if (rc != REED_OK) return rc;
            const int ft = g_force_tile;
            g_force_tile = 0;                     // the tail through the ordinary selection
            const int rt = reed_gemm_launch(layout, epi, t, splits, stream);
            g_force_tile = ft;
            return rt;
          }
        }
      }
    }
  }
  if (g_force_tile == 259) {   // no split for this shape: the ordinary selection
    g_force_tile = 0;
    const int rc = reed_gemm_launch(layout, want, a, splits, stream);
    g_force_tile = 259;
    return rc;
  }
  // (a 128x256 tile with two workgroups per CU — an epilogue overlapping the other workgroup's K loop — was built in round 4,
  // bit-identical and slower: its operand stream is 1.5x per flop; profiles/r4_gemm128c_*.txt, DESIGN_HISTORY.md; removed in round 5)
  if (g_force_tile == 64 && want != EPI_BF16_DOT && reed_gemm_skinny_eligible(layout, epi, a, splits))
    return reed_gemm_skinny_launch(epi, a, stream);   // tests: the skinny kernel on any shape it accepts
  if ((g_force_tile == 257 || g_force_tile == 258) && reed_gemm256w_eligible(layout, epi, a, splits))
    return reed_gemm256w_launch(layout, want, a, stream);   // 257: one-shot form, 258: persistent form wherever it applies
  if ((g_force_tile == 288 && reed_gemm288_eligible(layout, epi, a, splits)) ||
      (g_force_tile == 0 && reed_gemm288_preferred(layout, epi, a, splits))) {
    REED_ONLY_PLAIN();
    return reed_gemm288_launch(epi, a, stream);
  }
  if (can144 && (g_force_tile == 144 || a.N % BN != 0 || (g_force_tile == 0 && reed_gemm144_preferred(layout, epi, a, splits)))) {
    REED_ONLY_PLAIN();
    return reed_gemm144_launch(layout, epi, a, stream);
  }
  if (g_force_tile != 128 && (g_force_tile == 256 || reed_gemm256_preferred(layout, epi, a, splits)) &&
      !(layout == LAY_TN && a.dbias)) {
    // the 256^2 tile: four waves of 128x128 (gemm256w.hip) where that kernel is built, else eight of 128x64 (gemm256.hip);
    // force_tile 256 keeps the 8-wave kernel (tests, A/B timing)
    if (g_force_tile != 256 && reed_gemm256w_eligible(layout, epi, a, splits))
      return reed_gemm256w_launch(layout, want, a, stream);
    REED_ONLY_PLAIN();
    return reed_gemm256_launch(layout, epi, a, splits, stream);
  }
  REED_ONLY_PLAIN();
#undef REED_ONLY_PLAIN
  switch (layout) {
    case LAY_NT: return dispatch_epi<LAY_NT>(epi, a, splits, stream);
    case LAY_NN: return dispatch_epi<LAY_NN>(epi, a, splits, stream);
    case LAY_TN: return dispatch_epi<LAY_TN>(epi, a, splits, stream);
  }
  reed_set_error("reed_gemm: unknown layout %d", layout);
  return REED_ERR_ARG;
}
