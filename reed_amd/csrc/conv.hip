// 3x3 convolution (padding 1, optional nearest x2 upsampling of the input) as an IMPLICIT GEMM on the 16-bit matrix cores — the
// convolutions of the SD-VAE decoder (image/generate.py:87,156, image/train.py:446-447 through diffusers'
// AutoencoderKL.decode).
//
//   out f32 [B*Ho*Wo, N] (+)= sum_{tap, c} a[b, (y + tap/3 - 1) >> up, (x + tap%3 - 1) >> up, c] * w[n, tap * C + c]  + bias[n]
//
// a = the activation in the operand type, NHWC (the output of reed_conv_rows with taps = 1: GroupNorm apply + SiLU + rounding, one
// 6-byte-per-element pass), w = the weight as [N, 9 C] in (ky, kx, ci) order.  No im2col matrix exists anywhere: the GEMM's row
// operand is gathered by the LDS-DMA itself — a K-tile of 64 channels lies inside one tap (C % 64 == 0), so for K-tile t every
// lane adds the tap's (dy, dx) to the output pixel of each of its four tile rows, shifts by the upsampling and points its
// 16-byte `buffer_load ... lds` at that pixel's channels; a window position outside the image gets an offset beyond the
// descriptor's range and the hardware writes zeros (the padding).  The nine taps re-read the same activation rows out of L2.
// Tile 128 x 128 x 64, four waves of 64 x 64 (v_mfma_f32_16x16x32), two workgroups per CU, LDS double buffered — the structure
// of gemm.hip's kernel; the epilogue goes through the wave's LDS patch so that every global access is row-contiguous, adds the
// fp32 bias and, with `accumulate`, the fp32 residual already in `out` (the ResNet block's skip connection, in place).
#include "gemm_common.hpp"

namespace {
using namespace gemm_detail;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 16384;
constexpr int STAGE_BYTES = 2 * TILE_BYTES;

struct ConvArgs {
  const bf16* a;      // [B, Hi, Wi, C]
  const bf16* w;      // [N, 9 C]
  const float* bias;  // [N] or null
  float* out;         // [M, ldc]
  long ldc;
  int M, N, C, Hi, Wi, up, accumulate;
};

__global__ __launch_bounds__(256, 2) void conv3x3_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // block -> tile: XCD-contiguous runs, grouped along M (the tiles of a group share their activation rows in the XCD's L2)
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
  const int m0 = tm * BM, n0 = tn * BN;
  const int K = 9 * a.C, nt = K / BK;
  const int Ho = a.Hi << a.up, Wo = a.Wi << a.up;

  const long abytes = (long)(a.M >> (2 * a.up)) * a.C * 2;        // B * Hi * Wi * C operand elements
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.a, abytes);
  const __amdgpu_buffer_rsrc_t rsW = make_rsrc(a.w + (long)n0 * K, (long)(a.N - n0) * K * 2);

  // this lane's four tile rows (i * 32 + tid / 8): output pixel and image base, fixed for the whole K loop
  int py[4], px[4], pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + i * 32 + (tid >> 3);
    const int b = m / (Ho * Wo), rem = m - b * (Ho * Wo);
    py[i] = m < a.M ? rem / Wo : -4;                              // a row beyond M: every tap lands outside the image
    px[i] = rem - (rem / Wo) * Wo;
    pb[i] = b * a.Hi * a.Wi;
  }
  const int cchunk = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;          // stage_row's swizzle: chunk (tid & 7) of row r holds k-chunk c

  auto stage = [&](int t, int buf) {
    char* tp = smem + buf * STAGE_BYTES;
    char* tq = tp + TILE_BYTES;
    const int k0 = t * BK;
    const int tap = k0 / a.C, cb = k0 - tap * a.C;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int yy = py[i] + dy, xx = px[i] + dx;
      const bool in = (unsigned)yy < (unsigned)Ho && (unsigned)xx < (unsigned)Wo;
      const int pix = pb[i] + (yy >> a.up) * a.Wi + (xx >> a.up);
      const int voff = in ? (pix * a.C + cb + cchunk) * 2 : 0x7FFFFFF0;
      char* dst = tp + (i * 256 + wave * 64) * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int L = i * 256 + tid;
      const int r = L >> 3, c = (L & 7) ^ ((r >> 1) & 7);
      const int voff = (int)(((long)r * K + k0 + c * 8) * 2);
      char* dst = tq + (i * 256 + wave * 64) * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
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
        pf[ks][i] = frag_row(tp, wm * 64 + i * 16, ks, lane);
        qf[ks][i] = frag_row(tq, wn * 64 + i * 16, ks, lane);
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = REED_MFMA_16x16x32(qf[ks][j], pf[ks][i], acc[i][j]);
    __syncthreads();
  }

  // epilogue: 16-row groups through this wave's 4 KiB LDS patch -> lane owns 8 consecutive columns of two rows
  char* stg = smem + wave * EPI_STAGE_BYTES;
  const int wr_row = lane & 15, wr_g = lane >> 4;
  const int rd_row = lane >> 3, rd_c = lane & 7;
  char* wr_base = stg + wr_row * 256;
  const char* rd_base0 = stg + rd_row * 256;
  const char* rd_base1 = stg + (rd_row + 8) * 256;
  const int rd_sw0 = ((2 * rd_c) ^ rd_row) << 4, rd_sw1 = ((2 * rd_c) ^ (rd_row + 8)) << 4;
  const int col = n0 + wn * 64 + 8 * rd_c;
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (a.bias) {
    b0 = *(const f32x4*)(a.bias + col);
    b1 = *(const f32x4*)(a.bias + col + 4);
  }
  const long rows = a.M - m0;
  const __amdgpu_buffer_rsrc_t rsC = epi_rsrc(a.out + (long)m0 * a.ldc, ((rows - 1) * a.ldc + a.N) * 4);
  int oc = (int)(((wm * 64 + rd_row) * a.ldc + col) * 4);
  const int s8 = (int)(8 * a.ldc * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i, oc += 2 * s8) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(wr_base + (((4 * j + wr_g) ^ wr_row) << 4)) = acc[i][j];
    asm volatile("" ::: "memory");
    f32x4 q00 = *(const f32x4*)(rd_base0 + rd_sw0), q01 = *(const f32x4*)(rd_base0 + (rd_sw0 ^ 16));
    f32x4 q10 = *(const f32x4*)(rd_base1 + rd_sw1), q11 = *(const f32x4*)(rd_base1 + (rd_sw1 ^ 16));
    asm volatile("" ::: "memory");
    q00 += b0; q01 += b1; q10 += b0; q11 += b1;
    if (a.accumulate) {
      q00 += ld_f32x4(rsC, oc);
      q01 += ld_f32x4(rsC, oc + 16);
      q10 += ld_f32x4(rsC, oc + s8);
      q11 += ld_f32x4(rsC, oc + s8 + 16);
    }
    st_f32x4(q00, rsC, oc);
    st_f32x4(q01, rsC, oc + 16);
    st_f32x4(q10, rsC, oc + s8);
    st_f32x4(q11, rsC, oc + s8 + 16);
  }
}

}  // namespace

extern "C" int reed_conv3x3(const void* a, const void* w, const float* bias, float* out, int64_t ldc, int B, int Hi, int Wi, int C,
                            int N, int upsample, int accumulate, void* stream) {
  REED_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && C > 0 && N > 0, "reed_conv3x3: empty problem");
  REED_CHECK_ARG(upsample == 0 || upsample == 1, "reed_conv3x3: upsample must be 0 or 1 (nearest x2)");
  REED_CHECK_ARG(C % 64 == 0 && N % 128 == 0, "reed_conv3x3: C=%d must be a multiple of 64 and N=%d of 128 (use reed_conv_rows + reed_gemm otherwise)", C, N);
  REED_CHECK_ARG(ldc >= N && ldc % 4 == 0 && ldc < (1 << 20), "reed_conv3x3: ldc=%ld", (long)ldc);
  const long M = (long)B * (Hi << upsample) * (Wi << upsample);
  REED_CHECK_ARG((long)B * Hi * Wi * C * 2 <= 0x7FFF0000l && M < (1l << 31) - 128,
                 "reed_conv3x3: the activation must stay below 2 GiB (32-bit DMA offsets): decode fewer images per call");
  REED_CHECK_ARG(((uintptr_t)a % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 16) == 0 && (!bias || (uintptr_t)bias % 16 == 0),
                 "reed_conv3x3: operands must be 16-byte aligned");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv3x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    attr_set = true;
  }
  ConvArgs ca{(const bf16*)a, (const bf16*)w, bias, out, (long)ldc, (int)M, N, C, Hi, Wi, upsample, accumulate};
  const int ntm = cdiv(M, BM), ntn = N / BN;
  REED_KLAUNCH(conv3x3_kernel, dim3(ntm * ntn), dim3(256), 2 * STAGE_BYTES, (hipStream_t)stream, ca);
  REED_LAUNCH_CHECK();
  return REED_OK;
}
