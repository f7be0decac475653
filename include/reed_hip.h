/*
 * reed_hip.h — C ABI of libreed_hip.so, the MI355X (gfx950) kernel library behind the
 * reed_amd drop-in for REED's image/ SiT training + sampling path.
 *
 * The reference (ChenyuWang-Monica/REED, image/) has no FFI of its own: it is pure PyTorch.
 * Each entry point below replaces the arithmetic of the reference lines cited next to it; the
 * Python host (reed_amd/) binds them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates; the library never
 *    allocates, frees or retains memory past the call); tensors are contiguous row-major unless
 *    a leading dimension argument says otherwise;
 *  - `stream` is a hipStream_t passed as void* (PyTorch's current stream); all work is enqueued
 *    there and the call returns immediately;
 *  - return value 0 = OK, otherwise an argument-check code (1001/1002) or a hipError_t;
 *    reed_last_error() returns a thread-local message;
 *  - dtypes are fixed per entry point: "bf16" = __bf16 / torch.bfloat16, f32, f64, i64.
 */
#ifndef REED_HIP_H
#define REED_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* reed_last_error(void);
int reed_version(void);
/* 0: this library was built with bfloat16 operands (libreed_hip.so: the training path); 1: IEEE half operands
 * (libreed_hip_f16.so, the same sources with -DREED_FP16: the sampling path at the mantissa of the reference's TF32) */
int reed_half_kind(void);

/* ---------------------------------------------------------------------------------------------
 * Dense contractions (every nn.Linear on the path: sit.py:17-24 projector, :38-42 t-MLP,
 * :114-124 attention qkv/proj + Mlp fc1/fc2 (timm), :126-129 & :146-150 adaLN; and their autograd).
 *   layout 0 NT: C[M,N] = P[M,K] Q[N,K]^T     (forward  y = x W^T)
 *   layout 1 NN: C[M,N] = P[M,K] Q[K,N]       (dgrad    dx = dy W)
 *   layout 2 TN: C[M,N] = P[K,M]^T Q[K,N]     (wgrad    dW = dy^T x), optional dbias[M] = colsum(P)
 *   layout 3 / 4: TN on a 256x128 / 128x256 output tile (128x64 / 64x128 per wave; epilogue 6 only; 4 needs N%256==0)
 * epilogue codes: see reed_amd/csrc/gemm.h (11 = exact GELU(erf) with the layout of 9; 0 bf16, 1 gelu, 2 silu, 3 gate+residual, 4 dgelu,
 *   5 dsilu, 6 f32 (+=), 7 f32 += bf16-rounded, 8 f32 atomic, 9 QuickGELU, 10 + bf16 residual, 12 LayerScale + fp32 residual:
 *   C f32 = R f32 + gamma[n] * float(bf16(acc + bias)) with `gate` = the fp32 gamma vector, NT only; 13 = 0 plus the per-row,
 *   per-head partial dot products with R for reed_attention_bwd_dp (NN; returns 1002 without launching where the shape's
 *   kernel has no such epilogue); 14 / 15 = 1 / 2 with C = the activation's DERIVATIVE at the pre-activation instead of the
 *   pre-activation itself (what the backward needs: nothing else read the saved pre-activation), 16 = C bf16 =
 *   bf16(bf16(acc) * R): the activation backward as one multiply by that saved derivative.  N%128==0 (NT/NN with a
 *   bf16-output epilogue also N%144==0: the 256x144 tile of csrc/gemm144.hip); K%64==0 (NT/NN);
 *   M%128==0 (TN).  split_k>1 only with epilogue 8, or 6 with slab_stride>0 (C then holds split_k slabs;
 *   dbias likewise holds split_k slabs of M floats AT THE SAME slab_stride — put slab 0 of dbias right behind slab 0
 *   of C and one reed_reduce_slabs call over M*N + M floats finishes both; deterministic).
 * ------------------------------------------------------------------------------------------- */
int reed_gemm(int layout, int epilogue, const void* P, int64_t ldp, const void* Q, int64_t ldq,
              int M, int N, int K, void* C, int64_t ldc, void* C2, int64_t ldc2, const void* R,
              int64_t ldr, const void* bias, const void* gate, int64_t ldgate, int rows_per_gate,
              float* dbias, int accumulate, int split_k, int64_t slab_stride, void* stream);

/* The weight gradients of one transformer block (autograd of the four nn.Linear of image/models/sit.py:114-129's Attention /
 * Mlp) in ONE launch without split-K: n <= 4 problems dw_i f32 [n_out_i, k_in_i] (+)= dy_i[tokens, n_out_i]^T x_i[tokens, k_in_i]
 * (16-bit operands, row-major, leading dims = widths), optional dbias_i f32 [n_out_i] (+)= colsum(dy_i).  Pointer / int
 * arrays live on the host.  k_in % 128 == 0, n_out % 16 == 0.  Returns 1002 without launching when the problems' 256x128 /
 * 128x256 tiles do not fit one round of 2 x CUs workgroups (use reed_gemm layout 3 / 4 with split_k then). */
int reed_wgrad_group(int n, const void* const* dy, const void* const* x, float* const* dw, float* const* dbias,
                     const int* n_out, const int* k_in, int tokens, int accumulate, void* stream);
/* Planning side of reed_wgrad_group (host arithmetic, no device call): how the 16-bit builds' one-workgroup-per-CU form of the
 * grouped launch (csrc/gemm256w.hip: TnGroupW) deals the problems over `cus` CUs.  items[256]: entry x * (cus / 8) + j = the item of
 * workgroup j of XCD x, 0xFFFFFFFF = none; an item = problem | mode << 2 | row unit << 6 | column unit << 14 | (valid row units - 1)
 * << 22 | (valid column units - 1) << 24 | bias gradient << 26 in units of 128 rows / columns of dw: mode 0 = a 256 x 256 tile,
 * 6 = 384 x 128 (+ the bias gradient of its rows), 7 = 128 x 384, 8 = the bias gradient of up to 512 rows, no output.  Returns the
 * number of items; 0 = the form does not apply to these problems (or this build: fp32 operands) and reed_wgrad_group uses the
 * two-workgroups-per-CU kernel (csrc/gemm_tn.hip). */
int reed_wgrad_group_deal(int n, const int* n_out, const int* k_in, const int* has_bias, int cus, unsigned* items);

/* CUs the GEMM tile heuristics plan for: the device's count minus a reserve for kernels that hold CUs
 * beside the GEMMs (RCCL channels during a gradient bucket).  reed_set_cu_reserve(n): n >= 0; reed_planning_cus(): the result. */
int reed_set_cu_reserve(int n);
int reed_planning_cus(void);
/* reed_set_concurrent_comm(1): collectives will run beside the following GEMMs (a data-parallel backward).  Kernels that hold
 * a whole CU per workgroup for their entire run (the persistent form of the 256x256 kernel) are then not selected. */
int reed_set_concurrent_comm(int on);

/* tile selection override for tests / A-B timing: 0 = heuristic (default), 128 or 256 = force that kernel, 144 = force
 * the 256x144 kernel wherever it applies (NT / NN, bf16-output epilogue, N % 144 == 0, no split-K) */
int reed_gemm_force_tile(int tile);

/* bias gradient: out[n] (+)= sum_m x[m,n], x bf16 [M,N] (row stride ld); ws: caller scratch of
 * reed_colsum_ws_floats(M, N) floats. Deterministic (fixed reduction order). */
int64_t reed_colsum_ws_floats(int M, int N);
int reed_colsum_bf16(const void* x, int64_t ld, float* ws, float* out, int M, int N, int accumulate,
                     void* stream);

/* sum split-K slabs: out[i] (+)= sum_z slabs[z*stride + i], i < n */
int reed_reduce_slabs(const float* slabs, int64_t stride, int nslabs, float* out, int64_t n,
                      int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm(eps, no affine) + modulate  (sit.py:26-27,113,119,132-135,143,152)
 *   h[m,:] = bf16( LN(x[m,:]) * bf16(1+scale[b,:]) + shift[b,:] ),  b = m / T
 * x f32 [M,D]; shift/scale bf16 with row stride ldmod; h bf16 [M,D]; mean/rstd f32 [M] (optional).
 * scale==NULL -> plain cast f32->bf16 of x (projector input cast, sit.py:292).
 * ------------------------------------------------------------------------------------------- */
int reed_ln_modulate_fwd(const float* x, const void* shift, const void* scale, int64_t ldmod,
                         void* h, float* mean, float* rstd, int M, int D, int T, float eps,
                         void* stream);
/* backward: dx[m,:] += LNbwd(dh * bf16(1+scale)); partial column sums over each 16-row chunk:
 *   part[(m/16), 0, :] = sum dh, part[(m/16), 1, :] = sum dh*xhat        (f32 [M/16, 2, D]) */
int reed_ln_modulate_bwd(const void* dh, const float* x, const float* mean, const float* rstd,
                         const void* scale, int64_t ldmod, float* dx, float* part, int M, int D,
                         int T, void* stream);

/* the same followed, in the same pass, by the gate backward of the next branch in backward order (the one that
 * consumes the dx just finished): see reed_gate_bwd for dy / part_g / part_dy. Saves one read of the fp32 dx. */
int reed_ln_modulate_bwd_gate(const void* dh, const float* x, const float* mean, const float* rstd,
                              const void* scale, int64_t ldmod, float* dx, float* part, const void* y,
                              const void* gate, int64_t ldgate, void* dy, float* part_g, float* part_dy,
                              int M, int D, int T, void* stream);

/* gate backward (sit.py:134-135): dg = bf16(dx); dy = bf16(dg*gate[b]); part[(m/16),:] = sum bf16(dg*y);
 * optional part_dy[(m/16),:] = sum dy  (bias gradient of the linear that produced y, reduce with reed_rowsum_f32) */
int reed_gate_bwd(const float* dx, const void* y, const void* gate, int64_t ldgate, void* dy,
                  float* part, float* part_dy, int M, int D, int T, void* stream);
/* dst bf16 [C,R] = src bf16 [R,C]^T (utility; the training path no longer needs transposed copies) */
int reed_transpose_bf16(const void* src, void* dst, int R, int C, void* stream);
/* out[n] (+)= sum_r part[r, n], f32 [R, N], fixed order; ws (optional, cdiv(R,64)*N floats) enables the two-stage
 * path for tall inputs */
int reed_rowsum_f32(const float* part, int R, float* ws, float* out, int N, int accumulate, void* stream);

/* reduce per-chunk partials to bf16 modulation grads:
 *   dmod[b, col0 + j*D + d] = bf16( sum_{c<T/16} part_j[(b*T/16 + c)*stride_j + d] ) for the listed parts */
int reed_reduce_mod_parts(const float* const* parts, const int64_t* strides, const int64_t* offs,
                          int nparts, void* dmod, int64_t lddmod, int B, int D, int chunks,
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention (timm Attention under sit.py:114-118; softmax(q k^T / sqrt(hd)) v), head_dim 64 or 72.
 * qkv bf16 [B, T, 3, H, hd] as produced by the qkv Linear; o bf16 [B, T, H*hd]; lse f32 [B,H,T].
 * ------------------------------------------------------------------------------------------- */
int reed_attention_fwd(const void* qkv, void* o, float* lse, int B, int T, int H, int hd,
                       void* stream);
int reed_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse,
                       void* dqkv, int B, int T, int H, int hd, void* stream);
/* The same backward with a caller-owned workspace of reed_attention_bwd_ws_floats(B, T, H) floats (delta = rowsum(dO * O) is
 * formed there by a row kernel, which lets the main kernel run persistently with the next (batch, head) item's operands in
 * flight under the current one: csrc/attention.hip); ws == NULL falls back to reed_attention_bwd.  Replaces the autograd of
 * F.scaled_dot_product_attention inside timm Attention (image/models/sit.py:114-118). */
int64_t reed_attention_bwd_ws_floats(int B, int T, int H);
int reed_attention_bwd_ws(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, float* ws,
                          int B, int T, int H, int hd, void* stream);
/* The same with delta taken from dpart f32 [H, S, B*T] (S = 1 for hd 64, 2 for hd 72): the per-row, per-head partial dot
 * products dO . O that reed_gemm's epilogue 13 leaves when it PRODUCES dO (the input gradient of the attention output
 * projection, layout NN, R = O, C2 = dpart, rows_per_gate = hd) — the 302 MB row pass over dO and O (b = 256) becomes a
 * 12 MB one, and O is not read by the backward at all.  16-bit builds; the fp32 build returns 1002. */
int reed_attention_bwd_dp(const void* qkv, const void* d_o, const float* lse, const float* dpart, void* dqkv, float* ws,
                          int B, int T, int H, int hd, void* stream);

/* qk_norm (timm Attention(qk_norm=True); reference flag --qk-norm, sit.py:114-116): LayerNorm over head_dim (eps, affine
 * f32 [hd]) on the q and k thirds of qkv bf16 [M,3,H,hd]; v copied through; stats f32 [M,2,H,2] = (mean, rstd).
 * backward: dpre = LNbwd(dn*w) (v copied); part f32 [nblocks,2(q,k),2(dw,db),hd] per-block partial parameter grads
 * (reed_qk_norm_bwd_part_floats floats), reduce with reed_rowsum_f32. */
int reed_qk_norm_fwd(const void* qkv, const float* qw, const float* qb, const float* kw, const float* kb,
                     void* out, float* stats, int M, int H, int hd, float eps, void* stream);
int64_t reed_qk_norm_bwd_part_floats(int M, int H, int hd);
int reed_qk_norm_bwd(const void* dn, const void* qkv, const float* stats, const float* qw, const float* kw,
                     void* dpre, float* part, int M, int H, int hd, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Embedders and the final layer
 * ------------------------------------------------------------------------------------------- */
/* PatchEmbed (timm, sit.py:198-200,279): tokens f32 [B,T,D] = bf16(conv_p(x bf16, W bf16)+b) + pos_embed f32 */
int reed_patch_embed_fwd(const float* x, const void* w, const void* bias, const float* pos,
                         float* tokens, int B, int C, int HW, int P, int D, void* stream);
/* patchify to bf16 rows: out bf16 [B*T, C*P*P]; order 0 = (c,pi,pj) (conv input, timm PatchEmbed),
 * order 1 = (pi,pj,c) (the unpatchify order of sit.py:256-269) */
int reed_patchify_bf16(const float* x, void* out, int B, int C, int HW, int P, int order, void* stream);
/* small-K weight gradient (patch-embed conv weight, final linear): deterministic two-stage reduction over M rows
 *   out[layout 0: d*KS+k | layout 1: k*Dw+d] (+)= sum_m wide[m,d] * small[m,k]   (wide f32 is bf16-rounded first)
 *   colsum_wide[d] (+)= sum_m wide[m,d];  colsum_small[k] (+)= sum_m small[m,k]   (either may be NULL)
 * ws: caller workspace of reed_smallk_wgrad_ws_floats(Dw, KS) floats. */
int64_t reed_smallk_wgrad_ws_floats(int Dw, int KS);
int reed_smallk_wgrad(const void* wide, int wide_is_f32, const void* small, float* ws, float* out,
                      float* colsum_wide, float* colsum_small, int M, int Dw, int KS, int layout,
                      int accumulate, void* stream);
/* TimestepEmbedder.positional_embedding (sit.py:45-64): out bf16 [B, dim] = [cos(t f) | sin(t f)] */
int reed_timestep_sinusoid(const float* t, void* out, int B, int dim, float max_period, void* stream);
/* LabelEmbedder (sit.py:84-99): labels_out = drop ? num_classes : labels; c_out f32 [B,D] = t_emb bf16 + table f32[label];
 * silu_c bf16 [B,D] = bf16(silu(c)) (input of every adaLN linear, sit.py:126-129). table has table_rows rows; a label
 * outside [0, table_rows) (nn.Embedding raises, sit.py:98) is replaced by row 0 and *err_flag (device int, may be NULL)
 * is set to 1 — the caller reads it at its next synchronisation point. */
int reed_label_cond(const int64_t* labels, const uint8_t* drop, int num_classes, int table_rows, const float* table,
                    const void* t_emb, int64_t* labels_out, float* c, void* silu_c, int* err_flag, int B, int D,
                    void* stream);
/* backward of the conditioning vector: dc f32 [B,D] -> dt_emb bf16, dtable f32 [rows,D] += (deterministic scatter;
 * dsilu_c is the f32 grad w.r.t. silu(c), bf16-rounded inside) */
int reed_label_cond_bwd(const float* dsilu_c, const float* c, const int64_t* labels_eff, void* dt_emb,
                        float* dtable, int B, int D, void* stream);
/* FinalLayer + unpatchify (sit.py:140-158,256-269): out f32 [B,C,HW,HW] */
int reed_final_layer_fwd(const float* x, const void* shift, const void* scale, int64_t ldmod,
                         const void* w, const void* bias, float* out, float* mean, float* rstd,
                         int B, int T, int D, int C, int P, float eps, void* stream);
/* backward, row part: recompute h = bf16(modulate(LN(x))) -> hbuf bf16 [M,D]; dlin bf16 [M, P*P*C] = patchify(dout);
 * dh bf16 [M,D] = dlin @ W.  Follow with reed_ln_modulate_bwd(dh, ...) and reed_smallk_wgrad(hbuf, dlin) */
int reed_final_layer_bwd_rows(const float* dout, const float* x, const float* mean, const float* rstd,
                              const void* shift, const void* scale, int64_t ldmod, const void* w,
                              void* hbuf, void* dlin, void* dh, int B, int T, int D, int C, int P,
                              void* stream);
/* mean over tokens for the text projector (sit.py:292,301): out bf16 [B,D] = bf16(mean_t x f32 [B,T,D]) */
int reed_token_mean_fwd(const float* x, void* out, int B, int T, int D, void* stream);
int reed_token_mean_bwd(const void* dmean, float* dx, int B, int T, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SILoss arithmetic (loss.py:49-64,153-237)
 * ------------------------------------------------------------------------------------------- */
/* sample_posterior (train.py:84-91): out [B,half] = (moments[:, :half] + moments[:, half:] * eps) * scale + bias */
int reed_sample_posterior(const float* moments, const float* eps, float* out, int B, int64_t half,
                          float scale, float bias, void* stream);
/* path_type 0 linear, 1 cosine: xt = a x + s n, target = da x + ds n */
int reed_interpolant(const float* x, const float* noise, const float* t, float* xt, float* target,
                     int B, int64_t per, int path_type, void* stream);
/* loss[b] = mean((out-target)^2) */
int reed_mse_fwd(const float* out, const float* target, float* loss, int B, int64_t per, void* stream);
/* dout = gscale[b] * 2 (out-target) / per */
int reed_mse_bwd(const float* out, const float* target, const float* gscale, float* dout, int B,
                 int64_t per, void* stream);
/* cosine alignment: loss[b] = -mean_t <normalize(z), normalize(zt)>;  zt bf16 [B*T,Z] (projector out), z f32 */
int reed_cosine_fwd(const void* zt, const float* z, float* rowdot, float* loss, int B, int T, int Z,
                    void* stream); /* rowdot: caller scratch f32 [B*T] */
int reed_cosine_bwd(const void* zt, const float* z, const float* gscale, void* dzt, int B, int T,
                    int Z, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser side (train.py:94-105 update_ema, :402-409 clip + AdamW)
 * ------------------------------------------------------------------------------------------- */
/* partial[i] = sum of squares of chunk i (nblocks partials); finalize: norm_clip[0]=||g||, [1]=min(1,max_norm/(||g||+1e-6)) */
int reed_grad_sqnorm(const float* g, int64_t n, float* partial, int nblocks, void* stream);
int reed_clip_finalize(const float* partial, int nblocks, float max_norm, float* norm_clip, void* stream);
/* the same with dynamic loss scaling on the device (fp16 training; torch.amp.GradScaler as accelerate drives it,
 * image/train.py:401-409): scaler_state f32[4] = [scale, growth_tracker, found_inf, steps_taken]; the arena holds scale x the
 * gradients; norm_clip[0] = the true norm (inf / nan on overflow), [1] = clip coefficient / scale, 0 on overflow; the state is
 * updated as GradScaler.update() does (x backoff on overflow; x growth after growth_interval clean steps) */
int reed_clip_finalize_scaled(const float* partial, int nblocks, float max_norm, float* norm_clip, float* scaler_state,
                              float growth_factor, float backoff_factor, float growth_interval, void* stream);
/* fused clip * AdamW + EMA + 16-bit shadow over a flat arena; norm_clip may be NULL (no clipping); scaler_state (may be
 * NULL): an overflowed step leaves p, m, v untouched (EMA and shadow still run) and the bias corrections count the steps
 * actually taken (bc1 / bc2 are then ignored) */
int reed_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, void* shadow,
                   int64_t n_train, int64_t n_total, const float* norm_clip, const float* scaler_state, float lr,
                   float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2, float ema_decay,
                   void* stream);
int reed_cast_bf16(const float* src, void* dst, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Samplers (samplers.py:46-104,107-187): fp64 state updates
 * ------------------------------------------------------------------------------------------- */
/* d = cfg ? d_u + s (d_c - d_u) : d_c  (model out f32 [2n or n]); x_next = x_cur + dt * (w0 d + w1 d_prev)
 * d_store (f64, optional) receives d for Heun's second stage. */
int reed_sampler_update(const double* x_cur, const float* model_out, const double* d_prev,
                        double* d_store, double* x_next, int64_t n_elems, int cfg, double cfg_scale,
                        double dt, double w0, double w1, void* stream);
/* SDE step (samplers.py:146-156): drift from velocity + score, Euler-Maruyama update */
int reed_sde_update(const double* x_cur, const float* model_out, const double* eps, double* x_next,
                    int64_t n_elems, int cfg, double cfg_scale, double t_cur, double dt,
                    int path_type, int last_step, void* stream);
/* model input f32 [2n or n] = f32(x f64) (duplicated when cfg) */
int reed_sampler_input(const double* x, float* out, int64_t n_elems, int dup, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gradient all-reduce over RCCL/xGMI (replaces accelerate->DDP bucketed all-reduce, train.py:293-295,401)
 * ------------------------------------------------------------------------------------------- */
int reed_comm_unique_id(void* out128);                       /* 128-byte ncclUniqueId */
int reed_comm_init(const void* id128, int rank, int world, void** comm_out);
int reed_comm_allreduce_avg(void* comm, float* buf, int64_t count, void* compute_stream); /* async on the comm stream, ordered after compute_stream */
/* the same average issued as ncclReduceScatter + ncclAllGather, in place (the direct form of SURVEY.md §5; count % world
 * tail elements through ncclAllReduce) */
int reed_comm_allreduce_avg_rsag(void* comm, float* buf, int64_t count, void* compute_stream);
int reed_comm_sync(void* comm, void* compute_stream);       /* compute_stream waits for all pending reductions */
int reed_comm_broadcast(void* comm, float* buf, int64_t count, int root, void* compute_stream);
/* all-gather of `bytes` bytes per rank (recv = world * bytes, rank-major), async on the comm stream, ordered after
 * compute_stream; reed_comm_sync_gather makes compute_stream wait for the gathers issued so far (not for reductions
 * queued behind them).  Used to exchange the K = local-batch factors of the adaLN weight gradient instead of
 * all-reducing the matrix (reed_amd/parallel.py). */
int reed_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes, void* compute_stream);
int reed_comm_sync_gather(void* comm, void* compute_stream);
int reed_comm_destroy(void* comm);

/* ---------------------------------------------------------------------------------------------
 * Frozen CLIP image encoder, forward only: the producer of the alignment targets that
 * image/train.py:351-357 runs under autocast every step (image/models/clip_vit.py:208-230 UpdatedVisionTransformer:
 * conv1 14x14/14 -> [class token | patches] + positional embedding -> ln_pre -> ResidualAttentionBlocks (:173-195,
 * nn.MultiheadAttention + QuickGELU MLP) -> tokens without the class token; no ln_post, no projection).
 * The contractions go through reed_gemm (epilogues 0, 9 QuickGELU, 10 + bf16 residual) and reed_attention_fwd (hd 64,
 * T = 257); these are the row passes around them.
 * ------------------------------------------------------------------------------------------- */
/* out bf16 [B*(S/P)^2, Kp]: row (b, gy, gx), column c*P*P + py*P + px of img f32 [B,3,S,S]; columns >= 3 P^2 zero */
int reed_clip_im2col(const float* img, void* out, int B, int S, int P, int Kp, void* stream);
/* out bf16 [B,T,D]: row 0 = bf16(bf16(cls) + bf16(pos[0])), row t = bf16(patches[b,t-1] + bf16(pos[t])); T = patches+1 */
int reed_clip_tokens(const void* patches, const float* cls, const float* pos, void* out, int B, int T, int D,
                     void* stream);
/* out bf16 = bf16(LayerNorm_fp32(float(x bf16 [M,D]); eps) * w + b)   (clip_vit.py:159-165) */
int reed_ln_affine_bf16(const void* x, const float* w, const float* b, void* out, int M, int D, float eps,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * The other frozen towers the reference can name (image/utils.py:73-82 mocov3, :133-147 mae, :149-160 jepa): plain pre-LN
 * ViTs (image/models/jepa.py:173-218 Attention / Block, :376-466 VisionTransformer; image/models/mae_vit.py:20-48 and
 * mocov3_vit.py:52-101 over timm's VisionTransformer) with an fp32 residual stream under autocast, LayerNorm(eps 1e-6,
 * affine), exact GELU (reed_gemm epilogue 11) and head_dim 64 or 80 (reed_attention_fwd).  Row passes:
 * ------------------------------------------------------------------------------------------- */
/* out (bf16 or f32, row stride ldo) = LayerNorm_fp32(x f32 [M,D]; eps) * w + b */
int reed_ln_affine_f32(const float* x, const float* w, const float* b, void* out, int out_is_f32, int M, int D,
                       int64_t ldo, float eps, void* stream);
/* out f32 [B,T,D]: rows t < nprefix = cls[t] + pos[t] (cls f32 [nprefix, D]: the class token, then DINOv2's register tokens
 * with zero pos rows), rows t >= nprefix = float(patches bf16[b, t - nprefix]) + pos[t]; T = nprefix + patches */
int reed_vit_tokens(const void* patches, const float* cls, int nprefix, const float* pos, float* out, int B, int T, int D,
                    void* stream);
/* preprocess_raw_image (image/train.py:53-74): raw u8 [B,3,R,R] -> out f32 [B,3,S,S]; order 0: /255 -> bicubic -> normalise
 * ('clip'), order 1: /255 -> normalise -> bicubic ('dinov2', 'jepa'); S == R: no resampling ('mocov3', 'mae').
 * mean3 / std3 are HOST pointers to 3 floats.  Bicubic = F.interpolate(mode='bicubic', align_corners=False). */
int reed_preprocess_image(const uint8_t* raw, float* out, int B, int R, int S, const float* mean3, const float* std3,
                          int order, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SD-VAE decoder: `vae.decode(latents / 0.18215).sample` of image/generate.py:87,156 and
 * image/train.py:446-447 (diffusers' AutoencoderKL, not vendored in the reference: reed_amd/vae.py restates its decoder).
 * Activations are fp32 NHWC = the token matrix [B*H*W, C]; every convolution / Linear is a reed_gemm call on it (NT, epilogue 6:
 * fp32 + bias, accumulate = the residual connection); these are the passes around the contractions (csrc/vae.hip).
 * ------------------------------------------------------------------------------------------- */
/* doubles of workspace reed_groupnorm_stats needs for x f32 [B, hw, C] */
int64_t reed_groupnorm_ws_doubles(int B, int64_t hw, int C);
/* stats f32 [B, G, 2] (optional) = (mean, rstd = 1 / sqrt(biased var + eps)) of x f32 [B, hw, C] per (image, group of C / G
 * channels), fp64 sums in a fixed order; table f32 [B, 3, C] (optional; needs gamma, beta f32 [C]) = per channel
 * (mean of its group, rstd * gamma[c], beta[c]) for reed_conv_rows.  C % 4 == 0, C <= 1024. */
int reed_groupnorm_stats(const float* x, int B, int64_t hw, int C, int G, float eps, const float* gamma, const float* beta,
                         double* ws, float* stats, float* table, void* stream);
/* The row operand of a convolution as a GEMM: out (operand type) [nrows, ldo], row r - row0 = output position r = (b, y, x) of
 * the [B, Hi << upsample, Wi << upsample] grid, columns tap * C + c = a(b, y + tap / 3 - 1, x + tap % 3 - 1, c) for taps = 9
 * (zero outside the grid: padding 1) or a(b, y, x, c) for taps = 1, columns [taps * C, kcols) zero; a = x f32 [B, Hi, Wi, C]
 * read through nearest x2 upsampling when `upsample`, GroupNorm-ed as fma(x - mean, rstd * gamma, beta) with reed_groupnorm_stats'
 * table when table != NULL, SiLU-ed when `silu`.  taps = 1 is the activation pass in front of reed_conv3x3 and of the
 * attention's Linear layers.  C, kcols, ldo multiples of 4; fewer than 2^31 output positions. */
int reed_conv_rows(const float* x, const float* table, int B, int Hi, int Wi, int C, int silu, int upsample, int taps,
                   int64_t row0, int64_t nrows, int kcols, void* out, int64_t ldo, void* stream);
/* 3x3 convolution, padding 1, as an implicit GEMM (no im2col matrix: the kernel's row operand is gathered from the activation,
 * zero outside the image; 16-bit builds: csrc/conv.hip on v_mfma_f32_16x16x32, LDS-DMA gather; fp32 build: csrc/gemm_f32.hip's
 * kernel with the window gather in its staging loads): out f32 [B*Ho*Wo, ldc] (+)= conv(a) + bias, a (operand type) NHWC
 * [B, Hi, Wi, C] read through nearest x2 upsampling when `upsample` (Ho = Hi << upsample), w (operand type) [N, 9 C] in
 * (ky, kx, ci) order, bias f32 [N] or NULL; accumulate != 0 adds onto what `out` holds (the residual connection, in place).
 * 16-bit builds: C % 64 == 0, N % 128 == 0, B*Hi*Wi*C*2 bytes < 2 GiB; fp32 build: C % 4 == 0, N % 4 == 0. */
int reed_conv3x3(const void* a, const void* w, const float* bias, float* out, int64_t ldc, int B, int Hi, int Wi, int C, int N,
                 int upsample, int accumulate, void* stream);
/* p (operand type) [rows, ldp] = softmax over the first `cols` columns of scale * s f32 [rows, lds] */
int reed_softmax_rows(const float* s, int64_t lds, void* p, int64_t ldp, int rows, int cols, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
