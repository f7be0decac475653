// Micro-benchmark: what ONE CU streams from / to HBM as a function of how many CUs stream at the same time -- the question behind
// the GEMM epilogues: are 640 KiB per tile at 12-16 B/clk the CU's limit or the chip's?
// n workgroups of 4 or 8 waves (128 KiB of LDS each: one per CU), each sweeping its own 8 MiB of a 2 GiB buffer (beyond the
// 256 MiB Infinity Cache) with 16-byte-per-lane instructions of the epilogue's shape (a wave instruction = 1 KiB contiguous),
// loads kept DEPTH deep per wave, stores fire-and-forget, and the 1 : 1.5 load : store mix of the gate + residual epilogue.
// Build: hipcc -O3 --offload-arch=gfx950 cu_rate.hip -o cu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr long WG_BYTES = 8l << 20;

// MODE 0: loads, 1: stores, 2: one load + one and a half stores per KiB loaded
template <int MODE, int DEPTH>
__global__ void k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, unsigned* sink) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const long per_wave = WG_BYTES / 16 / nw;                       // 16-byte elements per wave
  const u32x4* s = src + (long)blockIdx.x * (WG_BYTES / 16) + wave * per_wave + lane;
  u32x4* d = dst + (long)blockIdx.x * (WG_BYTES / 16) + wave * per_wave + lane;
  u32x4* d2 = dst + 256 * (WG_BYTES / 16) + (long)blockIdx.x * (WG_BYTES / 32) + wave * (per_wave / 2) + lane;   // the bf16 copy
  const int steps = (int)(per_wave / 64);
  u32x4 acc = {0, 0, 0, 0};
  if (MODE == 1) {
    const u32x4 v = {(unsigned)lane, 1, 2, 3};
    for (int i = 0; i < steps; ++i) __builtin_nontemporal_store(v, d + (long)i * 64);
  } else {
    u32x4 r[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) r[j] = __builtin_nontemporal_load(s + (long)j * 64);
    for (int i = 0; i < steps; i += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        const u32x4 v = r[j];
        const int nx = i + DEPTH + j;
        r[j] = __builtin_nontemporal_load(s + (long)(nx < steps ? nx : j) * 64);
        if (MODE == 2) {
          __builtin_nontemporal_store(v, d + (long)(i + j) * 64);
          if (j & 1) __builtin_nontemporal_store(v, d2 + (long)((i + j) >> 1) * 64);   // half the bytes again
        } else {
          acc ^= v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) acc ^= r[j];
  }
  if (acc[0] == 0x12345 && smem[lane] == 77) sink[0] = acc[1] ^ acc[2] ^ acc[3];
}

// The epilogue's real shape: a wave instruction covers R rows x (1024 / R) contiguous bytes of a row-major matrix (pitch 7168 B),
// the wave walks down its column strip.  R = 1: contiguous; 8: the LDS-patch epilogues (8 rows x 128 B); 16: the MFMA layout.
template <int R, bool LOAD>
__global__ void kr(const u32x4* __restrict__ src, u32x4* __restrict__ dst, unsigned* sink) {
  extern __shared__ char smem[];
  constexpr long P = 7168, S = 1024 / R, NS = P / S, RPW = (2l << 20) / S;   // pitch, segment, strips per row, rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long w = (long)blockIdx.x * 4 + wave, c = w % NS, rb = w / NS;
  const long lr = lane / (64 / R), lc = lane % (64 / R);
  const long base = ((rb * RPW + lr) * P + c * S + lc * 16) / 16;
  u32x4 acc = {0, 0, 0, 0};
  const u32x4 v = {(unsigned)lane, 1, 2, 3};
  if (LOAD) {
    u32x4 r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __builtin_nontemporal_load(src + base + (long)j * R * P / 16);
    for (long i = 0; i < RPW / R; i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc ^= r[j];
        const long nx = i + 8 + j;
        r[j] = __builtin_nontemporal_load(src + base + (nx < RPW / R ? nx : j) * R * P / 16);
      }
    }
  } else {
    for (long i = 0; i < RPW / R; ++i) __builtin_nontemporal_store(v, dst + base + i * R * P / 16);
  }
  if (acc[0] == 0x12345 && smem[lane] == 77) sink[0] = acc[1] ^ acc[2] ^ acc[3];
}
template <int R, bool LOAD>
static void run_rows(u32x4* src, u32x4* dst, unsigned* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)kr<R, LOAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
  const int ns[] = {8, 32, 64, 128, 256};
  printf("%s, %2d rows x %4d B per wave instruction |", LOAD ? "loads 8 deep" : "stores      ", R, 1024 / R);
  for (int n : ns) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((kr<R, LOAD>), dim3(n), dim3(256), 128 << 10, 0, src, dst, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double gbs = (double)WG_BYTES / (best * 1e-3) / 1e9;
    printf(" n=%3d: %5.1f GB/s/CU %5.2f TB/s |", n, gbs, gbs * n / 1e3);
  }
  printf("\n");
}

template <int MODE, int DEPTH>
static void run(const char* name, int waves, u32x4* src, u32x4* dst, unsigned* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
  const int ns[] = {8, 32, 64, 96, 128, 192, 256};
  printf("%-34s %d waves per CU |", name, waves);
  for (int n : ns) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(n), dim3(64 * waves), 128 << 10, 0, src, dst, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double bytes = (MODE == 2 ? 2.5 : 1.0) * WG_BYTES;     // per CU
    const double gbs = bytes / (best * 1e-3) / 1e9;
    printf(" n=%3d: %5.1f GB/s/CU %5.2f TB/s |", n, gbs, gbs * n / 1e3);
  }
  printf("\n");
}

int main() {
  u32x4 *src, *dst; unsigned* sink;
  if (hipMalloc(&src, 288 * WG_BYTES) != hipSuccess || hipMalloc(&dst, 384 * WG_BYTES) != hipSuccess) return 1;
  hipMalloc(&sink, 64);
  hipMemset(src, 1, 288 * WG_BYTES); hipMemset(dst, 0, 384 * WG_BYTES);
  run_rows<1, false>(src, dst, sink);
  run_rows<2, false>(src, dst, sink);
  run_rows<4, false>(src, dst, sink);
  run_rows<8, false>(src, dst, sink);
  run_rows<16, false>(src, dst, sink);
  run_rows<1, true>(src, dst, sink);
  run_rows<4, true>(src, dst, sink);
  run_rows<8, true>(src, dst, sink);
  run_rows<16, true>(src, dst, sink);
  run<0, 4>("loads, 4 deep per wave", 4, src, dst, sink);
  run<0, 8>("loads, 8 deep per wave", 4, src, dst, sink);
  run<0, 16>("loads, 16 deep per wave", 4, src, dst, sink);
  run<0, 8>("loads, 8 deep per wave", 8, src, dst, sink);
  run<0, 16>("loads, 16 deep per wave", 8, src, dst, sink);
  run<1, 1>("stores", 4, src, dst, sink);
  run<1, 1>("stores", 8, src, dst, sink);
  run<2, 4>("load + 1.5 stores, loads 4 deep", 4, src, dst, sink);
  run<2, 8>("load + 1.5 stores, loads 8 deep", 4, src, dst, sink);
  run<2, 16>("load + 1.5 stores, loads 16 deep", 4, src, dst, sink);
  run<2, 8>("load + 1.5 stores, loads 8 deep", 8, src, dst, sink);
  return 0;
}
