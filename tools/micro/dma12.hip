// Where does `buffer_load_dwordx3 ... lds` put each lane's 12 bytes?  (gfx950: LDS-DMA in 12-byte lanes)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef void __attribute__((address_space(3))) * lds_ptr_t;
__global__ void k(const int* p, int* out) {
  __shared__ int smem[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) smem[i] = -1;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(p), 0, 1 << 20, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)smem, 12, threadIdx.x * 12, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = smem[i];
}
int main() {
  int *p, *o, h[1024];
  hipMalloc(&p, 4096); hipMalloc(&o, 4096);
  for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMemcpy(p, h, 4096, hipMemcpyHostToDevice);
  k<<<1, 64>>>(p, o);
  hipMemcpy(h, o, 4096, hipMemcpyDeviceToHost);
  for (int i = 0; i < 260; ++i) printf("%d%c", h[i], (i % 16 == 15) ? '\n' : ' ');
  printf("\n");
  return 0;
}
