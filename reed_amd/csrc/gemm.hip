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
#include "common.hpp"
#include "gemm.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 16384;          // one operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long bytes) {
  if (bytes < 0) bytes = 0;
  if (bytes > 0xFFFFFFFFl) bytes = 0xFFFFFFFFl;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (unsigned)bytes, 0x00020000);
}

typedef void __attribute__((address_space(3))) * lds_ptr_t;

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
__device__ __forceinline__ int tr_sw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
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

// ---- fragment reads -----------------------------------------------------------
// lane (i = lane&15, g = lane>>4) gets X[rowbase+i][ks*32 + 8g .. +7]
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int rowbase, int ks, int lane) {
  int row = rowbase + (lane & 15);
  int c = ks * 4 + (lane >> 4);
  return *(const bf16x8*)(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
}
// lane (i, g) gets X[k = ks*32 + 8g + j][colbase + i], j = 0..7 (two transposing reads)
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int colbase, int ks, int lane) {
  int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
  int row0 = ks * 32 + 8 * g + q, row1 = row0 + 4;
  int ch = (colbase >> 3) + (p >> 1);
  const char* a0 = tile + row0 * 256 + ((ch ^ tr_sw(row0)) << 4) + ((p & 1) << 3);
  const char* a1 = tile + row1 * 256 + ((ch ^ tr_sw(row1)) << 4) + ((p & 1) << 3);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)a0);
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)a1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- epilogue -------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ void epilogue(const GemmArgs& a, f32x4 acc, int m, int n, int z) {
  if (m >= a.M) return;
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
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 pf[4], qf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (LAY == LAY_TN) pf[i] = frag_tr(tp, wm * 64 + i * 16, ks, lane);
        else pf[i] = frag_row(tp, wm * 64 + i * 16, ks, lane);
        if constexpr (LAY == LAY_NT) qf[i] = frag_row(tq, wn * 64 + i * 16, ks, lane);
        else qf[i] = frag_tr(tq, wn * 64 + i * 16, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], pf[i], acc[i][j], 0, 0, 0);
      if constexpr (LAY == LAY_TN) {
        if (do_dbias) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[i], accb[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = ..+(lane&15)][n = ..+4*(lane>>4) .. +3] ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      epilogue<EPI>(a, acc[i][j], m, n, z);
    }
  }
  if constexpr (LAY == LAY_TN) {
    if (do_dbias && (lane >> 4) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m < a.M) {
          if (gridDim.y > 1) a.dbias[(long)z * a.M + m] = accb[i][0];  // split-K: per-slice slab, reduced by the caller
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
    case EPI_F32: return launch<LAY, EPI_F32>(a, splits, s);
    case EPI_ADDF32_RB: return launch<LAY, EPI_ADDF32_RB>(a, splits, s);
    case EPI_ATOMIC_F32: return launch<LAY, EPI_ATOMIC_F32>(a, splits, s);
  }
  reed_set_error("reed_gemm: unknown epilogue %d", epi);
  return REED_ERR_ARG;
}

}  // namespace

int reed_gemm_launch(int layout, int epi, GemmArgs a, int splits, hipStream_t stream) {
  REED_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "reed_gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  REED_CHECK_ARG(a.N % BN == 0, "reed_gemm: N=%d must be a multiple of %d", a.N, BN);
  REED_CHECK_ARG(a.ldp % 8 == 0 && a.ldq % 8 == 0, "reed_gemm: leading dims must be multiples of 8 elements");
  REED_CHECK_ARG(((uintptr_t)a.P % 16) == 0 && ((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.C % 16) == 0,
                 "reed_gemm: operands must be 16-byte aligned");
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
  switch (layout) {
    case LAY_NT: return dispatch_epi<LAY_NT>(epi, a, splits, stream);
    case LAY_NN: return dispatch_epi<LAY_NN>(epi, a, splits, stream);
    case LAY_TN: return dispatch_epi<LAY_TN>(epi, a, splits, stream);
  }
  reed_set_error("reed_gemm: unknown layout %d", layout);
  return REED_ERR_ARG;
}
