// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 vs the legacy v_mfma_f32_16x16x16_bf16 on gfx950
// (one wave per SIMD, 4 independent accumulators).  Build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  f32x4 acc[4] = {};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * (threadIdx.x ^ j)); }
  s16x4 a4 = __builtin_bit_cast(s16x4, __builtin_shufflevector(a, a, 0, 1, 2, 3));
  s16x4 b4 = __builtin_bit_cast(s16x4, __builtin_shufflevector(b, b, 0, 1, 2, 3));
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
      else acc[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[u], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc; long long h;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 100000;
  for (int kind = 0; kind < 2; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      hipDeviceSynchronize();
    }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s: %.2f cycles per MFMA per wave (memtime ticks / %d)\n", kind == 0 ? "16x16x32_bf16" : "16x16x16_bf16", (double)h / (iters * 4.0), iters * 4);
  }
  return 0;
}
