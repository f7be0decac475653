// fp32-operand GEMM for gfx950 — the dense contraction of the fp32 build of the library (libreed_hip_f32.so, -DREED_FP32):
// the reference's `--mixed-precision no` training (image/train.py:505) and `generate.py --no-tf32` sampling
// (image/generate.py:41,183), where every nn.Linear multiplies fp32 operands and accumulates in fp32.
//
//   C[M,N] (+)= sum_k P(m,k) * Q(n,k)        same three operand layouts and the same epilogue ids as csrc/gemm.hip
//
// Matrix instruction: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate; exact fp32 products, 64 cycles per SIMD = 1/16 of
// the bf16 rate, MI355X_MICROARCH.md "Matrix cores").  The matrix pipe is slow enough that the loop can stay the plain form:
// tile 128x128x16, four waves of 64x64 (2x2 MFMA tiles, 64 accumulator registers), two workgroups per CU, operands staged
// global -> registers -> LDS k-major ([16][128+32] floats per operand: a fragment read is 2 rows x 32 consecutive floats, the
// 32-float pad puts the two rows on disjoint bank halves), double buffered, one barrier per K step, XCD-contiguous tile map.
// Measured (tools/_ab/time_gemm_f32.py): 81-112 TFLOP/s at M = 8192, 112-127 at M = 65536 (71-81 % of the 157 TFLOP/s fp32
// MFMA peak).  (The geometry is a template on the wave grid: a 256x256 tile of SIXTEEN waves, one workgroup of 1024 threads
// per CU, was built for its 64 flop per operand byte and measured 59-120 TFLOP/s — level at M = 65536, 20-40 % slower at
// M = 8192 — so only the 2 x 2 grid is instantiated.)  Every bound
// (ragged M, N, K; split-K ranges) is a guard on the staging loads and on the stores, so there are no shape restrictions
// beyond 4-element alignment.  In this build `bf16` IS float (common.hpp): C, C2, R, bias and the gate are fp32 arrays and
// every rounding point of the mixed-precision epilogues is the identity — one kernel per layout, the epilogue a run-time
// switch.  This file is not part of the 16-bit builds (reed_amd/build.py).
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>

#include "../../include/reed_hip.h"
#include "gemm.h"

#ifndef REED_FP32
#error "gemm_f32.hip belongs to the fp32-operand build (-DREED_FP32) only"
#endif

int reed_num_cus();

namespace {

constexpr int BK = 16;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Tile geometry for a W x W grid of 64x64 waves: BT rows / columns, NT threads, LDT floats per k-row of an LDS tile (BT + 32
// pad), PP 16-byte pieces per thread, operand and K step.
template <int W>
struct Geo {
  static constexpr int BT = 64 * W, NT = 64 * W * W, LDT = BT + 32, TILE_FLOATS = BK * LDT, PP = BT * 4 / NT;
};

template <int N, class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int PP>
struct Stage {       // one K step of one operand in registers: PP float4 per thread
  f32x4 v[PP];
};

// k-contiguous operand X[rows][K] (ld floats per row): tile rows r0.. x k [k0, k0+16).  Piece p = t + i NT: row p>>2, k (p&3)*4.
template <int W>
__device__ __forceinline__ void load_row(Stage<Geo<W>::PP>& s, const float* X, long ld, int r0, int rows, int k0, int kend, int t) {
#pragma unroll
  for (int i = 0; i < Geo<W>::PP; ++i) {
    const int p = t + i * Geo<W>::NT;
    const int r = r0 + (p >> 2), k = k0 + (p & 3) * 4;
    s.v[i] = (r < rows && k < kend) ? *(const f32x4*)(X + (long)r * ld + k) : f32x4{0.f, 0.f, 0.f, 0.f};   // K % 4 == 0
  }
}
template <int W>
__device__ __forceinline__ void store_row(const Stage<Geo<W>::PP>& s, float* tile, int t) {
#pragma unroll
  for (int i = 0; i < Geo<W>::PP; ++i) {
    const int p = t + i * Geo<W>::NT;
    const int r = p >> 2, kq = (p & 3) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(kq + e) * Geo<W>::LDT + r] = s.v[i][e];
  }
}
// k-strided operand X[K][cols]: tile k [k0, k0+16) x cols c0..  Piece p = t + i NT: k-row p / (BT/4), columns (p % (BT/4))*4.
template <int W>
__device__ __forceinline__ void load_tr(Stage<Geo<W>::PP>& s, const float* X, long ld, int c0, int cols, int k0, int kend, int t) {
  constexpr int Q = Geo<W>::BT / 4;
#pragma unroll
  for (int i = 0; i < Geo<W>::PP; ++i) {
    const int p = t + i * Geo<W>::NT;
    const int k = k0 + p / Q, c = c0 + (p % Q) * 4;
    s.v[i] = (k < kend && c < cols) ? *(const f32x4*)(X + (long)k * ld + c) : f32x4{0.f, 0.f, 0.f, 0.f};     // cols % 4 == 0
  }
}
template <int W>
__device__ __forceinline__ void store_tr(const Stage<Geo<W>::PP>& s, float* tile, int t) {
  constexpr int Q = Geo<W>::BT / 4;
#pragma unroll
  for (int i = 0; i < Geo<W>::PP; ++i) {
    const int p = t + i * Geo<W>::NT;
    *(f32x4*)(tile + (p / Q) * Geo<W>::LDT + (p % Q) * 4) = s.v[i];
  }
}

__device__ __forceinline__ float act_fwd(int epi, int variant, float x) {
  if (epi == EPI_GELU) return gelu_tanh_f(x);
  if (epi == EPI_SILU) return silu_f(x);
  return variant ? gelu_erf_f(x) : x * sigmoid_f(1.702f * x);   // EPI_QGELU
}

// C > 0 (NT only): P is not a matrix but the 3x3 window gather of an NHWC activation a.P f32 [B, Hi, Wi, C] — row m = output pixel
// (b, y, x) of the [B, Hi << up, Wi << up] grid, column k = tap * C + c = a(b, (y + tap / 3 - 1) >> up, (x + tap % 3 - 1) >> up, c),
// zero outside the grid: the convolutions of the SD-VAE decoder (reed_conv3x3) without an im2col matrix.  C == 0: a plain GEMM.
struct ConvGeo {
  int C, Hi, Wi, up;
};

template <int LAY, int W>
__global__ __launch_bounds__(64 * W * W, W == 2 ? 2 : 1) void gemm_f32_kernel(GemmArgs a, int epi, ConvGeo cg) {
  constexpr int BM = Geo<W>::BT, BN = Geo<W>::BT, LDT = Geo<W>::LDT, TILE_FLOATS = Geo<W>::TILE_FLOATS;
  extern __shared__ __attribute__((aligned(16))) float smem_f32[];        // [buffer][P | Q][TILE_FLOATS]
  float (*smem)[2][TILE_FLOATS] = (float (*)[2][TILE_FLOATS])smem_f32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / W, wn = wave % W;
  // block -> tile: XCD-contiguous runs (consecutive workgroup ids alternate over the 8 XCDs), grouped along M: the tiles an
  // XCD works on at a time share operand panels in its L2
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  const int nwg = ntm * ntn;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
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

  typedef Stage<Geo<W>::PP> stage_t;
  // window gather: the output pixel of each of this thread's row pieces, decomposed once
  int cy[Geo<W>::PP], cx[Geo<W>::PP], cb[Geo<W>::PP];
  if (LAY == LAY_NT && cg.C > 0) {
    const int Ho = cg.Hi << cg.up, Wo = cg.Wi << cg.up;
#pragma unroll
    for (int i = 0; i < Geo<W>::PP; ++i) {
      const int m = m0 + ((tid + i * Geo<W>::NT) >> 2);
      const int b = m / (Ho * Wo), rem = m - b * (Ho * Wo);
      cy[i] = m < a.M ? rem / Wo : -4;          // a row beyond M: every tap lands outside
      cx[i] = rem - (rem / Wo) * Wo;
      cb[i] = b * cg.Hi * cg.Wi;
    }
  }
  auto load = [&](int t, stage_t& sp, stage_t& sq) {
    const int k0 = kbeg + t * BK;
    if constexpr (LAY == LAY_TN) load_tr<W>(sp, a.P, a.ldp, m0, a.M, k0, kend, tid);
    else if (LAY == LAY_NT && cg.C > 0) {
      const int Ho = cg.Hi << cg.up, Wo = cg.Wi << cg.up;
#pragma unroll
      for (int i = 0; i < Geo<W>::PP; ++i) {
        const int k = k0 + ((tid + i * Geo<W>::NT) & 3) * 4;
        const int tap = k / cg.C, c = k - tap * cg.C;
        const int yy = cy[i] + tap / 3 - 1, xx = cx[i] + (tap - (tap / 3) * 3) - 1;
        const bool in = k < kend && (unsigned)yy < (unsigned)Ho && (unsigned)xx < (unsigned)Wo;
        sp.v[i] = in ? *(const f32x4*)(a.P + ((long)(cb[i] + (yy >> cg.up) * cg.Wi + (xx >> cg.up)) * cg.C + c))
                     : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else load_row<W>(sp, a.P, a.ldp, m0, a.M, k0, kend, tid);
    if constexpr (LAY == LAY_NT) load_row<W>(sq, a.Q, a.ldq, n0, a.N, k0, kend, tid);
    else load_tr<W>(sq, a.Q, a.ldq, n0, a.N, k0, kend, tid);
  };
  auto store = [&](int buf, const stage_t& sp, const stage_t& sq) {
    if constexpr (LAY == LAY_TN) store_tr<W>(sp, smem[buf][0], tid);
    else store_row<W>(sp, smem[buf][0], tid);
    if constexpr (LAY == LAY_NT) store_row<W>(sq, smem[buf][1], tid);
    else store_tr<W>(sq, smem[buf][1], tid);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // TN: the bias gradient dbias[m] = sum_k P[k][m] (column sums of dY) rides along in the blocks of the first column tile:
  // thread t < 128 owns column m0 + t of the staged P tile
  const bool do_dbias = (LAY == LAY_TN) && a.dbias != nullptr && tn == 0;
  float bsum = 0.f;

  stage_t sp, sq;
  if (nt > 0) {
    load(0, sp, sq);
    store(0, sp, sq);
  }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) load(t + 1, sp, sq);
    const float* tp = smem[buf][0];
    const float* tq = smem[buf][1];
    const int kr = lane >> 5, c = lane & 31;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float pf[2], qf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        pf[i] = tp[(2 * ks + kr) * LDT + wm * 64 + i * 32 + c];
        qf[i] = tq[(2 * ks + kr) * LDT + wn * 64 + i * 32 + c];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[i], qf[j], acc[i][j], 0, 0, 0);
    }
    if (do_dbias && tid < BM) {
#pragma unroll
      for (int k = 0; k < BK; ++k) bsum += tp[k * LDT + tid];
    }
    if (t + 1 < nt) store(buf ^ 1, sp, sq);
    __syncthreads();
  }

  // ---- epilogue: lane owns column n (32 consecutive lanes = 32 consecutive columns) of 16 rows per MFMA tile ----
  const float* bias = a.bias;
  const float* Rf = (const float*)a.R;
  float* C = (float*)a.C;
  float* C2 = (float*)a.C2;
  // (compile-time indices into the accumulators: with run-time ones — a loop the compiler does not fully unroll around the
  //  epilogue switch — the 64 accumulator registers become a scratch array that is written back after EVERY K step: 16 KiB per
  //  wave and step, four times the operand traffic; that was the first version of this kernel at 35-50 TFLOP/s)
  auto elem = [&](auto I, auto J, auto Rr) {
    constexpr int i = decltype(I)::value, j = decltype(J)::value, r = decltype(Rr)::value;
    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
    const int m = m0 + wm * 64 + i * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
    if (n >= a.N || m >= a.M) return;
    const float v = acc[i][j][r] + (bias ? bias[n] : 0.f);
    switch (epi) {
      case EPI_BF16: C[(long)m * a.ldc + n] = v; break;
      case EPI_GELU: case EPI_SILU: case EPI_QGELU:
        if (C) C[(long)m * a.ldc + n] = v;
        C2[(long)m * a.ldc2 + n] = act_fwd(epi, a.act_variant, v);
        break;
      case EPI_GATE_RES: {
        if (C2) C2[(long)m * a.ldc2 + n] = v;
        const float g = a.gate[(long)(m / a.rows_per_gate) * a.ldgate + n];
        C[(long)m * a.ldc + n] = Rf[(long)m * a.ldr + n] + g * v;
      } break;
      case EPI_LS_RES: C[(long)m * a.ldc + n] = Rf[(long)m * a.ldr + n] + a.gate[n] * v; break;
      case EPI_GELU_G: case EPI_SILU_G: {   // C carries the activation's derivative (fp32 here: no rounding), gemm.h
        float av, gv;
        if (epi == EPI_GELU_G) gelu_tanh_both(v, av, gv);
        else silu_both(v, av, gv);
        if (C) C[(long)m * a.ldc + n] = gv;
        C2[(long)m * a.ldc2 + n] = av;
      } break;
      case EPI_MUL: C[(long)m * a.ldc + n] = v * Rf[(long)m * a.ldr + n]; break;
      case EPI_DGELU: C[(long)m * a.ldc + n] = v * gelu_tanh_grad_f(Rf[(long)m * a.ldr + n]); break;
      case EPI_DSILU: C[(long)m * a.ldc + n] = v * silu_grad_f(Rf[(long)m * a.ldr + n]); break;
      case EPI_RES_BF16: C[(long)m * a.ldc + n] = v + Rf[(long)m * a.ldr + n]; break;
      case EPI_F32: {
        float* cp = C + (long)z * a.slab_stride + (long)m * a.ldc + n;
        *cp = a.accumulate ? *cp + v : v;
      } break;
      case EPI_ADDF32_RB: C[(long)m * a.ldc + n] += v; break;
      case EPI_ATOMIC_F32: atomicAdd(C + (long)m * a.ldc + n, v); break;
    }
  };
  static_for<2>([&](auto I) { static_for<2>([&](auto J) { static_for<16>([&](auto Rr) { elem(I, J, Rr); }); }); });
  if (do_dbias && tid < BM) {
    const int m = m0 + tid;
    if (m < a.M) {
      if (gridDim.y > 1) a.dbias[(long)z * a.slab_stride + m] = bsum;   // split-K: per-slice slab (C's stride)
      else if (a.accumulate) a.dbias[m] += bsum;
      else a.dbias[m] = bsum;
    }
  }
}

template <int LAY, int W>
int launch_w(const GemmArgs& a, int epi, int splits, hipStream_t stream, ConvGeo cg = ConvGeo{0, 0, 0, 0}) {
  constexpr int lds = 2 * 2 * Geo<W>::TILE_FLOATS * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_f32_kernel<LAY, W>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { reed_set_error("gemm_f32: cannot reserve LDS: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  dim3 grid(cdiv(a.M, Geo<W>::BT) * cdiv(a.N, Geo<W>::BT), splits, 1);
  REED_KLAUNCH((gemm_f32_kernel<LAY, W>), grid, dim3(Geo<W>::NT), lds, stream, a, epi, cg);
  REED_LAUNCH_CHECK();
  return REED_OK;
}

template <int LAY>
int launch(const GemmArgs& a, int epi, int splits, hipStream_t stream) {
  return launch_w<LAY, 2>(a, epi, splits, stream);
}

}  // namespace

// The tile-selection knobs of the 16-bit kernels: accepted and without effect here (one kernel).
static int g_force_tile = 0, g_cu_reserve = 0, g_concurrent_comm = 0;
extern "C" int reed_gemm_force_tile(int tile) { g_force_tile = tile; return 0; }
extern "C" int reed_set_cu_reserve(int n) { g_cu_reserve = n > 0 ? n : 0; return 0; }
extern "C" int reed_set_concurrent_comm(int on) { g_concurrent_comm = on ? 1 : 0; return 0; }
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

// The grouped weight-gradient launch is a 16-bit MFMA kernel (gemm_tn.hip): not part of this build; the caller falls back to
// one reed_gemm(TN) per weight (ops.wgrad_group returns False on this code).
int reed_gemm_tn_group_launch(int, const GemmArgs*, hipStream_t) {
  reed_set_error("reed_wgrad_group: not built for fp32 operands (one reed_gemm(TN) per weight instead)");
  return REED_ERR_UNSUPPORTED;
}

extern "C" int reed_wgrad_group_deal(int, const int*, const int*, const int*, int, unsigned*) { return 0; }   // (16-bit builds only)

// 3x3 convolution, padding 1, optional nearest x2 upsampling of the input, as an implicit GEMM on the fp32 MFMA kernel above (the
// 16-bit builds have their own: conv.hip): out f32 [B*Ho*Wo, ldc] (+)= conv(a f32 NHWC [B, Hi, Wi, C]; w f32 [N, 9 C]) + bias.
extern "C" int reed_conv3x3(const void* act, const void* w, const float* bias, float* out, int64_t ldc, int B, int Hi, int Wi, int C,
                            int N, int upsample, int accumulate, void* stream) {
  REED_CHECK_ARG(act && w && out && B > 0 && Hi > 0 && Wi > 0 && C > 0 && N > 0, "reed_conv3x3: empty problem");
  REED_CHECK_ARG(upsample == 0 || upsample == 1, "reed_conv3x3: upsample must be 0 or 1 (nearest x2)");
  REED_CHECK_ARG(C % 4 == 0 && N % 4 == 0 && ldc >= N, "reed_conv3x3(fp32): C=%d and N=%d must be multiples of 4, ldc >= N", C, N);
  const long M = (long)B * (Hi << upsample) * (Wi << upsample);
  REED_CHECK_ARG(M < (1l << 31) - 256 && (long)B * Hi * Wi < (1l << 31), "reed_conv3x3: too many positions for one call");
  REED_CHECK_ARG(((uintptr_t)act % 16) == 0 && ((uintptr_t)w % 16) == 0 && (!bias || (uintptr_t)bias % 16 == 0),
                 "reed_conv3x3: operands must be 16-byte aligned");
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.P = (const float*)act;
  a.Q = (const float*)w;
  a.ldp = 9 * C;
  a.ldq = 9 * C;
  a.M = (int)M;
  a.N = N;
  a.K = 9 * C;
  a.C = out;
  a.ldc = ldc;
  a.bias = bias;
  a.accumulate = accumulate;
  a.rows_per_gate = 1;
  a.ksplit_len = a.K;
  return launch_w<LAY_NT, 2>(a, EPI_F32, 1, (hipStream_t)stream, ConvGeo{C, Hi, Wi, upsample});
}

int reed_gemm_launch(int layout, int epi, GemmArgs a, int splits, hipStream_t stream) {
  REED_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "reed_gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  REED_CHECK_ARG(a.ldp % 4 == 0 && a.ldq % 4 == 0 && a.N % 4 == 0, "reed_gemm(fp32): leading dims and N must be multiples of 4 elements");
  REED_CHECK_ARG(((uintptr_t)a.P % 16) == 0 && ((uintptr_t)a.Q % 16) == 0, "reed_gemm: operands must be 16-byte aligned");
  if (layout == LAY_TN_TALL || layout == LAY_TN_WIDE) layout = LAY_TN;   // tile hints of the 16-bit weight-gradient kernel
  if (layout == LAY_TN) REED_CHECK_ARG(a.M % 4 == 0, "reed_gemm(TN, fp32): M=%d must be a multiple of 4", a.M);
  else REED_CHECK_ARG(a.K % 4 == 0, "reed_gemm(NT/NN, fp32): K=%d must be a multiple of 4", a.K);
  switch (epi) {
    case EPI_BF16: case EPI_GELU: case EPI_SILU: case EPI_GATE_RES: case EPI_DGELU: case EPI_DSILU: case EPI_F32:
    case EPI_ADDF32_RB: case EPI_ATOMIC_F32: case EPI_QGELU: case EPI_RES_BF16: case EPI_LS_RES:
    case EPI_GELU_G: case EPI_SILU_G: case EPI_MUL: break;
    default: reed_set_error("reed_gemm: unknown epilogue %d", epi); return REED_ERR_ARG;
  }
  if (splits < 1) splits = 1;
  // K per split: the same arithmetic as the 16-bit kernels (units of 64), so callers that size slab workspaces by it
  // (ops.linear_wgrad, engine.py) see the same slab count from either build
  const int ksteps = cdiv(a.K, 64);
  const int per = cdiv(ksteps, splits);
  splits = cdiv(ksteps, per);
  a.ksplit_len = per * 64;
  if (splits > 1)
    REED_CHECK_ARG(epi == EPI_ATOMIC_F32 || (epi == EPI_F32 && a.slab_stride > 0),
                   "reed_gemm: split-K needs the atomic or slab fp32 epilogue");
  switch (layout) {
    case LAY_NT: return launch<LAY_NT>(a, epi, splits, stream);
    case LAY_NN: return launch<LAY_NN>(a, epi, splits, stream);
    case LAY_TN: return launch<LAY_TN>(a, epi, splits, stream);
  }
  reed_set_error("reed_gemm: unknown layout %d", layout);
  return REED_ERR_ARG;
}
