// Micro-benchmark: issue cost of vector instructions for ONE wave per SIMD (the four-wave GEMM's epilogue situation): cycles per
// instruction of a stream of independent instructions (8 rotating destination registers), 256 workgroups of 256 threads.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f;
  float d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a, b}, p1 = {b, c}, q0 = {0, 0}, q1 = {0, 0}, q2 = {0, 0}, q3 = {0, 0}, q4 = {0, 0}, q5 = {0, 0}, q6 = {0, 0}, q7 = {0, 0};
  unsigned u0 = threadIdx.x, u1 = threadIdx.x * 3;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d##i) : "v"(a), "v"(b), "v"(c));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %2" : "=v"(q##i) : "v"(p0), "v"(p1));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 2) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(q##i) : "v"(p0), "v"(p1));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 3) {
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(q##i) : "v"(p0), "v"(p1));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 4) {
#define X(i) asm volatile("v_exp_f32 %0, %1" : "=v"(d##i) : "v"(a));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 5) {
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d##i) : "v"(a), "v"(b));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 6) {
#define X(i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d##i) : "v"(a), "v"(b));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 7) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d##i) : "v"(a), "v"(b));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 8) {
#define X(i) asm volatile("v_rcp_f32 %0, %1" : "=v"(d##i) : "v"(a));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 9) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(d##i) : "v"(u0));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 10) {
#define X(i) asm volatile("v_accvgpr_write_b32 a" #i ", %0" ::"v"(a));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 11) {
#define X(i) asm volatile("v_accvgpr_read_b32 %0, a" #i : "=v"(d##i));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 12) {
      asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\t"
                   "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\t"
                   "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\t"
                   "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %0, %1"
                   : "+v"(u0), "+v"(u1));
    } else if (KIND == 13) {   // the GELU sequence as the compiler packs it: 1 pk_mul + 1 pk_fma + 1 pk_mul per 2 elements
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %2, %2\n\tv_pk_mul_f32 %0, %0, %1" : "=&v"(q##i) : "v"(p0), "v"(p1));
      REP8(X)
#undef X
    } else if (KIND == 14) {   // the same work unpacked: 6 plain instructions per 2 elements
#define X(i) asm volatile("v_mul_f32 %0, %2, %2\n\tv_mul_f32 %1, %3, %3\n\tv_fma_f32 %0, %0, %3, %3\n\tv_fma_f32 %1, %1, %2, %2\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3" : "=&v"(d##i), "=&v"(q##i[0]) : "v"(a), "v"(b));
      REP8(X)
#undef X
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + q0[0] + q1[1] + q2[0] + q3[1] + q4[0] + q5[1] + q6[0] + q7[1] + (float)(u0 ^ u1);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_iter, float* out, long long* cyc) {
  const int iters = 20000;
  long long h;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
  }
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %.2f cycles per instruction (one wave per SIMD)\n", name, (double)h / ((double)iters * per_iter));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<0>("v_fma_f32", 16, out, cyc);
  run<6>("v_add_f32", 16, out, cyc);
  run<7>("v_mul_f32", 16, out, cyc);
  run<1>("v_pk_fma_f32", 16, out, cyc);
  run<2>("v_pk_mul_f32", 16, out, cyc);
  run<3>("v_pk_add_f32", 16, out, cyc);
  run<4>("v_exp_f32", 16, out, cyc);
  run<8>("v_rcp_f32", 16, out, cyc);
  run<5>("v_cvt_pk_bf16_f32", 16, out, cyc);
  run<9>("v_lshlrev_b32", 16, out, cyc);
  run<10>("v_accvgpr_write_b32", 16, out, cyc);
  run<11>("v_accvgpr_read_b32", 16, out, cyc);
  run<12>("v_permlane16_swap_b32 (dependent)", 16, out, cyc);
  run<13>("3 packed ops per 2 elements (per instruction)", 24, out, cyc);
  run<14>("6 plain ops per 2 elements (per instruction)", 48, out, cyc);
  return 0;
}
