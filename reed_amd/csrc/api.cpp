// C-ABI glue: error reporting, version, and the GEMM entry point (see include/reed_hip.h).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/reed_hip.h"
#include "common.hpp"
#include "gemm.h"

static thread_local char g_err[512] = "";

extern "C" void reed_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* reed_last_error(void) { return g_err; }
extern "C" int reed_version(void) { return 100; }
extern "C" int reed_half_kind(void) { return REED_HALF_KIND; }   // 0: bfloat16 operands (libreed_hip.so), 1: IEEE half (libreed_hip_f16.so)

extern "C" int reed_gemm(int layout, int epilogue, const void* P, int64_t ldp, const void* Q,
                         int64_t ldq, int M, int N, int K, void* C, int64_t ldc, void* C2,
                         int64_t ldc2, const void* R, int64_t ldr, const void* bias,
                         const void* gate, int64_t ldgate, int rows_per_gate, float* dbias,
                         int accumulate, int split_k, int64_t slab_stride, void* stream) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.P = (const bf16*)P;
  a.Q = (const bf16*)Q;
  a.ldp = ldp;
  a.ldq = ldq;
  a.M = M;
  a.N = N;
  a.K = K;
  a.C = C;
  a.ldc = ldc;
  a.C2 = C2;
  a.ldc2 = ldc2;
  a.R = R;
  a.ldr = ldr;
  a.bias = (const bf16*)bias;
  a.gate = (const bf16*)gate;
  a.ldgate = ldgate;
  a.rows_per_gate = rows_per_gate > 0 ? rows_per_gate : 1;
  a.dbias = dbias;
  a.accumulate = accumulate;
  a.slab_stride = slab_stride;
  if (epilogue == EPI_GELU_ERF) {
    a.act_variant = 1;
#if defined(REED_FP32)
    epilogue = EPI_QGELU;   // the fp32-operand GEMM switches on the variant at run time (gemm_f32.hip)
#endif
  }
  const bool act_epi = epilogue == EPI_GELU || epilogue == EPI_SILU || epilogue == EPI_QGELU || epilogue == EPI_GELU_ERF ||
                       epilogue == EPI_GELU_G || epilogue == EPI_SILU_G;
  REED_CHECK_ARG(P && Q && (C || act_epi), "reed_gemm: null operand");
  if (act_epi) REED_CHECK_ARG(C2, "reed_gemm: activation epilogue needs C2");
  if (epilogue == EPI_RES_BF16) REED_CHECK_ARG(R, "reed_gemm: residual epilogue needs R");
  if (epilogue == EPI_GATE_RES) REED_CHECK_ARG(R && gate, "reed_gemm: gate-residual epilogue needs R and gate");
  if (epilogue == EPI_LS_RES)
    REED_CHECK_ARG(R && gate && layout == LAY_NT && split_k <= 1, "reed_gemm: LayerScale-residual epilogue: NT, R and gamma (gate)");
  if (epilogue == EPI_DGELU || epilogue == EPI_DSILU || epilogue == EPI_MUL) REED_CHECK_ARG(R, "reed_gemm: activation-grad epilogue needs R");
  return reed_gemm_launch(layout, epilogue, a, split_k, (hipStream_t)stream);
}

int reed_gemm_tn_group_launch(int n, const GemmArgs* probs, hipStream_t stream);   // gemm_tn.hip

extern "C" int reed_wgrad_group(int n, const void* const* dy, const void* const* x, float* const* dw, float* const* dbias,
                                const int* n_out, const int* k_in, int tokens, int accumulate, void* stream) {
  REED_CHECK_ARG(n >= 1 && n <= 4 && dy && x && dw && n_out && k_in && tokens > 0, "reed_wgrad_group: bad arguments");
  GemmArgs a[4];
  memset(a, 0, sizeof(a));
  for (int i = 0; i < n; ++i) {
    REED_CHECK_ARG(dy[i] && x[i] && dw[i], "reed_wgrad_group: null operand in problem %d", i);
    REED_CHECK_ARG(((uintptr_t)dy[i] % 16) == 0 && ((uintptr_t)x[i] % 16) == 0 && ((uintptr_t)dw[i] % 16) == 0 &&
                       n_out[i] % 8 == 0, "reed_wgrad_group: problem %d: operands must be 16-byte aligned", i);
    a[i].P = (const bf16*)dy[i];
    a[i].Q = (const bf16*)x[i];
    a[i].ldp = n_out[i];
    a[i].ldq = k_in[i];
    a[i].M = n_out[i];
    a[i].N = k_in[i];
    a[i].K = tokens;
    a[i].C = dw[i];
    a[i].ldc = k_in[i];
    a[i].dbias = dbias ? dbias[i] : nullptr;
    a[i].accumulate = accumulate;
    a[i].rows_per_gate = 1;
  }
  return reed_gemm_tn_group_launch(n, a, (hipStream_t)stream);
}
