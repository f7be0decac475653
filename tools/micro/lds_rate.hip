// Micro-benchmark: LDS read throughput per CU on gfx950 for ds_read_b128, ds_read_b64 and the transposing ds_read_b64_tr_b16,
// with 4 / 8 / 16 waves per CU (one workgroup of 256 / 512 / 1024 threads per CU, 256 workgroups).  Addresses follow the GEMMs'
// conflict-free patterns: b128 = 16 rows x 128 B with the chunk XOR swizzle; tr = the [k][128 col] format's 8-byte pieces.
// Build: hipcc -O3 --offload-arch=gfx950 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(unsigned* out, int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 4; i += blockDim.x) ((unsigned*)smem)[i] = i * 2654435761u;
  __syncthreads();
  unsigned base;
  if (KIND == 0) {          // b128: lane (i = lane & 15, g = lane >> 4): row i, chunk (g ^ ((i >> 1) & 7)) of a 128-byte row
    const int i = lane & 15, g = lane >> 4;
    base = (wave & 7) * 4096 + i * 128 + ((g ^ ((i >> 1) & 7)) << 4);
  } else if (KIND == 1) {   // b64 plain: 8 bytes per lane, consecutive
    base = (wave & 7) * 4096 + lane * 8;
  } else {                  // tr: lane (i, g), q = i >> 2, p = i & 3: row 8 g + q of 256-byte rows, piece p of a 32-byte tile slot
    const int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
    const int xe = (q << 1) | (g & 1);   // the GEMM format's slot swizzle: the four rows of a 16-lane group use different slots
    base = (8 * g + q) * 256 + (xe << 5) + ((p >> 1) << 4) + ((p & 1) << 3);
  }
  const unsigned a = (unsigned)(size_t)(char __attribute__((address_space(3)))*)smem + base;
  u32x4 acc4 = {0, 0, 0, 0};
  u32x2 acc2 = {0, 0};
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) {
        u32x4 r;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(u * 2048 % 4096));
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        acc4 ^= r;
      } else if (KIND == 1) {
        u32x2 r;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(u * 512));
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        acc2 ^= r;
      } else {
        u32x2 r;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(u * 8192));
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        acc2 ^= r;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + tid] = acc4[0] ^ acc4[1] ^ acc4[2] ^ acc4[3] ^ acc2[0] ^ acc2[1];
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  unsigned* out; long long* cyc; long long h;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  const char* names[3] = {"ds_read_b128", "ds_read_b64", "ds_read_b64_tr_b16"};
  const int bytes[3] = {1024, 512, 512};   // per wave instruction
  for (int kind = 0; kind < 3; ++kind)
    for (int waves = 4; waves <= 16; waves *= 2) {
      for (int rep = 0; rep < 2; ++rep) {
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 65536, 0, out, iters, cyc);
        else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 65536, 0, out, iters, cyc);
        else hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 65536, 0, out, iters, cyc);
        hipDeviceSynchronize();
      }
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const double per_cu = (double)waves * iters * 8 * bytes[kind] / (double)h;
      printf("%-20s %2d waves per CU: %.1f bytes per clock per CU (%.1f cycles per wave instruction)\n", names[kind], waves, per_cu,
             (double)h / (iters * 8.0));
    }
  return 0;
}
