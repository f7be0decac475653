// A separate PROCESS that holds n CUs for a number of seconds: back-to-back launches of a kernel whose n workgroups (256 threads,
// some LDS) sit on their CUs for 100 ms each (s_memrealtime, 100 MHz) and do nothing — a stand-in for RCCL's channels beside a whole
// training step on a one-GPU box (tools/r4/fortysixth.sh).  Every wave leaves when its ticks are over; the program ends by itself.
// Build: hipcc -O2 --offload-arch=gfx950 hog_main.hip -o hog_main     usage: hog_main <n CUs <= 128> <seconds <= 150> [LDS bytes]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void hog_kernel(long long ticks, int* sink) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (lds[(threadIdx.x + 1) & 255] == -1) *sink = 1;
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 16;
  const double secs = argc > 2 ? atof(argv[2]) : 10.0;
  const int lds = argc > 3 ? atoi(argv[3]) : 16384;
  if (n < 1 || n > 128 || secs <= 0 || secs > 150 || lds < 1024 || lds > 65536) { fprintf(stderr, "usage: hog_main <n> <seconds> [lds]\n"); return 2; }
  int* sink;
  if (hipMalloc(&sink, 4) != hipSuccess) return 1;
  hipStream_t st[2];
  hipStreamCreate(&st[0]); hipStreamCreate(&st[1]);
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  // two streams, each with one kernel in flight or queued: the next one starts when the previous one of its stream ends, and the two
  // streams' kernels overlap by half a period, so that n to 2 n CUs are held at every moment
  hipLaunchKernelGGL(hog_kernel, dim3(n), dim3(256), lds, st[0], 5000000LL, sink);
  for (int i = 1;; ++i) {
    hipLaunchKernelGGL(hog_kernel, dim3(n), dim3(256), lds, st[i & 1], 10000000LL, sink);
    ++launches;
    hipStreamSynchronize(st[(i & 1) ^ 1]);
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > secs) break;
  }
  hipDeviceSynchronize();
  fprintf(stderr, "hog_main: %d CUs x 2 streams held for %.1f s (%ld launches)\n", n, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), launches);
  return 0;
}
