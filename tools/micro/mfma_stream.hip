// Micro-benchmark: the four-wave GEMM's K loop as a bare stream — per K-tile of 64 and wave 32 fragment reads (ds_read_b128), 16 LDS-DMA
// pieces of the next K-tile (from an L2-resident buffer of random bf16), one barrier, and the 128 x 128 x 64 product of the wave either
// as 128 v_mfma_f32_16x16x32_bf16 (what csrc/gemm256w.hip issues: a chunk = {read, [DMA], 4 MFMAs}) or as 64 v_mfma_f32_32x32x16_bf16
// (a chunk = {read, [DMA every second chunk], 2 MFMAs}) on the same 256 accumulator registers.  One workgroup of four waves per CU,
// 256 CUs, long enough to reach the clock the chip holds.  Question: does the 32x32x16 form leave the issue port (8 of 16 cycles per
// 16x16x32 MFMA, 8 of 32 per 32x32x16) enough room, and what clock does each hold?  Not a GEMM: the fragments are whatever the reads
// return.  Build: hipcc -O3 --offload-arch=gfx950 mfma_stream.hip -o mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef void __attribute__((address_space(3))) * lds_ptr_t;

constexpr int HT = 16384;   // half-tile: 128 rows x 64 k bf16
constexpr int LDS_BYTES = 8 * HT;

// TR: every fragment through two ds_read_b64_tr_b16 (both operands k-strided: the weight gradients' TN product; 64 transposing
// reads per K-tile and wave) instead of one ds_read_b128
template <int M32, int DMA_ON, int TR = 0>
__global__ __launch_bounds__(256, 1) void k(const __bf16* __restrict__ src, float* out, int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // fill LDS with this workgroup's random slice once
  const __bf16* mine = src + (long)blockIdx.x * (LDS_BYTES / 2);
  for (int i = tid; i < LDS_BYTES / 16; i += 256) ((uint4*)smem)[i] = ((const uint4*)mine)[i];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(mine), 0, LDS_BYTES, 0x00020000);
  // fragment addresses: k-contiguous [128][64] half-tiles, 128-byte rows, chunk swizzle (row >> 1) & 7
  const int li = M32 ? (lane & 31) : (lane & 15), lg = M32 ? (lane >> 5) : (lane >> 4);
  const int rsw = (li >> 1) & 7;
  const int rA = (wave >> 1) * 2 * HT + li * 128 + ((lg ^ rsw) << 4);          // A half (wave >> 1), buffer 0
  const int rB = (4 + (wave & 1) * 2) * HT + li * 128 + ((lg ^ rsw) << 4);      // B half (wave & 1), buffer 0
  const int vo = (tid >> 3) * 128 + ((tid & 7) ^ ((tid >> 4) & 7)) * 16;        // DMA source offset of piece round 0

  bf16x8 fr[2][16];
  f32x4 a16[M32 ? 1 : 64];
  f32x16 a32[M32 ? 16 : 1];
  if constexpr (M32) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) a32[i][e] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 64; ++i) a16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // fragment f of (buffer cur, k-step ks): 16x16x32: f < 8 row tile f of A, else column tile f - 8 of B; 32x32x16: f = 8 s + (0..3 A | 4..7 B)
  auto ldfrag = [&](int cur, int ks, int f) -> bf16x8 {
    int a;
    if constexpr (M32) {
      const int s = f >> 3, t = f & 7;
      a = ((t < 4 ? rA : rB) ^ (ks ? 64 : 0) ^ (s ? 32 : 0)) + cur * HT + (t & 3) * 4096;
    } else {
      a = ((f < 8 ? rA : rB) ^ (ks ? 64 : 0)) + cur * HT + (f & 7) * 2048;
    }
    if constexpr (TR) {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 lo, hi;
      const unsigned ad = (unsigned)(size_t)(const char __attribute__((address_space(3)))*)(smem + a);
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(ad));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8" : "=v"(hi) : "v"(ad));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    return *(const bf16x8*)(smem + a);
  };
  auto dma = [&](int cur, int d) {   // piece d (0..15) of the next K-tile into buffer cur: 4 half-tiles x 4 rounds of 4 KiB per workgroup
    const int hh = d >> 2, i = d & 3;
    char* ht = smem + ((hh < 2 ? hh * 2 : 4 + (hh - 2) * 2) + cur) * HT;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(ht + (i * 256 + wave * 64) * 16), 16, vo + i * 4096, hh * HT, 0, 0);
  };
#pragma unroll
  for (int f = 0; f < 16; ++f) fr[0][f] = ldfrag(0, 0, f);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {   // phase ph: MFMAs of k-step ph from fr[ph], reads of the next k-step into fr[ph ^ 1]
      if (ph == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      } else {
        __builtin_amdgcn_s_waitcnt(0xC07F);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        fr[ph ^ 1][c] = ldfrag(ph == 0 ? cur : cur ^ 1, ph ^ 1, c);
        if (DMA_ON && ph == 1) dma(cur, c);
        if constexpr (M32) {
          // chunk c: sub-step s = c >> 3, row tile (c >> 1) & 3, column tiles 2 (c & 1), + 1
          const int s = c >> 3, i = (c >> 1) & 3, j0 = 2 * (c & 1);
#pragma unroll
          for (int j = j0; j < j0 + 2; ++j)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a32[i * 4 + j]) : "v"(fr[ph][8 * s + 4 + j]), "v"(fr[ph][8 * s + i]));
        } else {
          const int i = c >> 1, j0 = 4 * (c & 1);
#pragma unroll
          for (int j = j0; j < j0 + 4; ++j)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(a16[i * 8 + j]) : "v"(fr[ph][8 + j]), "v"(fr[ph][i]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  if constexpr (M32) {
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += a32[i][0] + a32[i][7];
  } else {
#pragma unroll
    for (int i = 0; i < 64; ++i) sum += a16[i][0] + a16[i][3];
  }
  out[blockIdx.x * 256 + tid] = sum;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int M32, int DMA_ON, int TR = 0>
static void run(const char* name, const __bf16* src, float* out, long long* cyc) {
  hipFuncSetAttribute((const void*)k<M32, DMA_ON, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  const int iters = 40000;   // ~ 50 ms
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<M32, DMA_ON, TR>), dim3(256), dim3(256), LDS_BYTES, 0, src, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double flop = 2.0 * 256 * 256 * 64 * (double)iters * 256;   // a 256 x 256 x 64 tile step per workgroup and iteration
  printf("%-44s %7.1f TFLOP/s | %7.1f cycles per K-tile (floor 2048) | clock %.3f GHz\n", name, flop / (ms * 1e-3) / 1e12,
         (double)h / iters, (double)h / (ms * 1e-3) / 1e9);
}

int main() {
  __bf16* src; float* out; long long* cyc;
  const size_t n = (size_t)256 * LDS_BYTES / 2;
  hipMalloc(&src, n * 2); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  unsigned short* h = (unsigned short*)malloc(n * 2);
  srand(1);
  for (size_t i = 0; i < n; ++i) {   // random bf16 in (-2, 2): sign, exponent 125..127, random mantissa
    const unsigned r = (unsigned)rand();
    h[i] = (unsigned short)(((r & 1) << 15) | ((125 + (r >> 1) % 3) << 7) | ((r >> 8) & 0x7f));
  }
  hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
  run<0, 1>("16x16x32, reads + DMA", src, out, cyc);
  run<1, 1>("32x32x16, reads + DMA", src, out, cyc);
  run<0, 0>("16x16x32, reads only", src, out, cyc);
  run<1, 0>("32x32x16, reads only", src, out, cyc);
  run<0, 1, 1>("16x16x32, 64 transposing reads + DMA", src, out, cyc);
  run<0, 0, 1>("16x16x32, 64 transposing reads only", src, out, cyc);
  run<0, 1>("16x16x32, reads + DMA (again)", src, out, cyc);
  run<1, 1>("32x32x16, reads + DMA (again)", src, out, cyc);
  return 0;
}
