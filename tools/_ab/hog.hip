// A stand-in for RCCL's channels on a one-GPU box: n workgroups of 256 threads with some LDS that hold their CUs for a given number
// of s_memrealtime ticks (100 MHz) and do nothing.  Every wave leaves when the ticks are over: the grid always drains.
// Build: hipcc -O2 --offload-arch=gfx950 -shared -fPIC hog.hip -o libhog.so   (tools/_ab/wgrad_under_hog.py loads it)
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void hog_kernel(long long ticks, int* sink) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (lds[(threadIdx.x + 1) & 255] == -1) *sink = 1;
}
extern "C" int hog_launch(int n, long long ticks, int lds_bytes, int* sink, void* stream) {
  if (n < 1 || n > 256 || ticks < 0 || ticks > 20000000 || lds_bytes < 1024 || lds_bytes > 65536) return -1;   // <= 0.2 s
  hipLaunchKernelGGL(hog_kernel, dim3(n), dim3(256), lds_bytes, (hipStream_t)stream, ticks, sink);
  return (int)hipGetLastError();
}
