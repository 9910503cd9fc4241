/*
 * dctn_amd — C-ABI of the MI355X (gfx950) kernel library for dctn's hot path.
 *
 * Drop-in boundary.  The reference (philip-bl/dctn) is pure Python: it has no FFI for this path;
 * its arithmetic is delegated to opt_einsum -> torch.einsum.  The entry points below are what a
 * binding for this path binds (ctypes stub: INTEGRATION.md); each one names the reference
 * function whose arithmetic it replaces (file:line relative to the reference repository).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes and element strides; no torch / C++ types.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream).  No allocation, no synchronisation, no host<->device copies inside: calls can be
 *     captured into a hipGraph.  Scratch memory is supplied by the caller (`*_workspace_bytes`).
 *   - return value: 0 on success, negative DCTN_ERR_* otherwise (dctn_strerror()).  The Python
 *     host layer turns shape errors into AssertionError like the reference's `assert`s.
 *   - dtype codes: DCTN_F32 / DCTN_F64 / DCTN_BF16 (bf16 storage, fp32 accumulation).
 *   - re-entrant.  Nothing a call computes depends on process state: no environment variables are read, there are
 *     no setters; the one mutable global is the diagnostic name returned by dctn_last_kernel().
 */
#ifndef DCTN_AMD_H
#define DCTN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DCTN_F32 = 0, DCTN_F64 = 1, DCTN_BF16 = 2, DCTN_DTYPE_MASK = 0xFF };
/* OR-ed into the `dtype` argument of the dctn_convsbs_* entry points (any other bit above DCTN_DTYPE_MASK:
 * DCTN_ERR_UNSUPPORTED): strings with every bond <= 4 run on the matrix-core sweep instead of the register-resident
 * sweep that is their default (cross-check of the two kernel families; same results to f32 rounding) */
enum { DCTN_SBS_MATRIX_CORE_SWEEP = 1 << 8 };

enum {
  DCTN_OK = 0,
  DCTN_ERR_BAD_SHAPE = -1,    /* sizes inconsistent (reference: AssertionError) */
  DCTN_ERR_BAD_DTYPE = -2,    /* unknown dtype code */
  DCTN_ERR_UNSUPPORTED = -3,  /* valid request no kernel of this build covers */
  DCTN_ERR_WORKSPACE = -4,    /* workspace missing or too small */
  DCTN_ERR_LAUNCH = -5,       /* HIP reported a launch error */
  DCTN_ERR_NULL = -6          /* required pointer is NULL */
};
/* positive success codes of the entry points that document them */
enum {
  DCTN_SAVED = 1,             /* dctn_eps_fwd_save / dctn_convsbs_fwd: the forward also WROTE the buffer a following
                                 *_bwd_saved call reads; DCTN_OK (0) = forward done, buffer untouched */
  DCTN_PARTIAL = 2            /* DCTN_OPT_MAIN_KERNEL_ONLY was honoured: only the dominant kernel ran, the
                                 gradients are NOT complete */
};

/* `policy` argument of the EPS entry points: one DCTN_PREC_* value (precision policy for float32 tensors on the
 * MFMA paths), optionally OR-ed with DCTN_OPT_* flags.  A workspace query and the call it sizes take the same policy. */
enum {
  DCTN_PREC_EXACT = 0, /* f32 in / f32 accumulate (v_mfma_f32_32x32x2_f32 or VALU fma) */
  DCTN_PREC_BF16 = 1,  /* operands rounded to bf16, f32 accumulate (v_mfma_f32_32x32x16_bf16) */
  DCTN_PREC_MASK = 0xFF
};
enum {
  /* float32 shapes that both exact-f32 families cover run on the two-halves GEMM path instead of the LDS-streamed
   * large-core kernels (cross-check of the two designs; same results to f32 rounding) */
  DCTN_OPT_F32_PREFER_HALVES = 1 << 8,
  /* two-halves path: bound its per-chunk buffers to 256 KiB instead of 1 GiB, so that small inputs take many
   * chunks (tests of the chunk loop; results identical) */
  DCTN_OPT_SMALL_CHUNKS = 1 << 9,
  /* measurement only: dctn_eps_bwd / dctn_eps_head_bwd on the register-resident family launch their dominant
   * kernel (per-workgroup partial sums) and skip the small reduction kernel, so that the kernel can be timed
   * alone; the gradients are NOT written and the call returns DCTN_PARTIAL (2), not DCTN_OK */
  DCTN_OPT_MAIN_KERNEL_ONLY = 1 << 10,
  /* dctn_eps_fwd / dctn_eps_bwd on the generic kernels whatever the shape: one lane per window walks the core rows
   * one factor digit after the other, no matrix cores, no Khatri-Rao halves - the independent evaluation order
   * behind `eps_one_by_one` (dctn/eps.py:43-63), which the reference's tests use to cross-check `eps` */
  DCTN_OPT_GENERIC_KERNELS = 1 << 11,
  DCTN_OPT_ALL = (1 << 8) | (1 << 9) | (1 << 10) | (1 << 11)   /* any other bit above DCTN_PREC_MASK: DCTN_ERR_UNSUPPORTED */
};

int dctn_version(void);
const char* dctn_strerror(int code);
/* name of the kernel family the last successful call dispatched to
 * (diagnostics / tests: proves which HIP path ran; process-wide, last writer wins) */
const char* dctn_last_kernel(void);

/* ------------------------------------------------------------------------------------------
 * EPS — replaces dctn/eps.py:19-40 `eps(core, input)` (and :43-63 `eps_one_by_one`, same result)
 *   x    : (C, B, H, W, Q) with element strides x_strides[5]
 *   core : (Q,)*(K*K*C) + (O,) contiguous == row-major matrix (Q^(K*K*C), O); factor index
 *          n = pos*C + ch, pos row-major over (dh, dw)  (dctn/align.py:31-32,41-45)
 *   out  : (B, H-K+1, W-K+1, O) contiguous
 *   workspace : scratch of dctn_eps_fwd_workspace_bytes() bytes (0 for most shapes; the large-core
 *          family splits its row tiles over the grid and sums the slices in a fixed order)
 * ------------------------------------------------------------------------------------------ */
/* Which kernel family the forward (and backward) of a shape runs on, for a contiguous input: callers use
 * it to route bf16 tensors with large cores through float32 (the exact-f32 matrix-core family; bf16
 * storage, f32 arithmetic) instead of the generic kernels — see dctn_amd/eps.py.  -1 = invalid shape. */
#define DCTN_EPS_FAMILY_GENERIC 0
#define DCTN_EPS_FAMILY_Q2REG 1        /* bf16 MFMA, Q = 2, N in {8, 9} */
#define DCTN_EPS_FAMILY_BIGCORE_F32 2  /* exact f32 MFMA, LDS-streamed core */
#define DCTN_EPS_FAMILY_HALVES 3       /* two-halves GEMM path: f64 MFMA, and f32 MFMA for shapes 1 and 2 leave */
#define DCTN_EPS_FAMILY_Q2REG_F32 4    /* exact f32 MFMA, register-resident core: Q = 2, N in {8, 9}, O <= 4 */
int dctn_eps_family(int C, int B, int H, int W, int Q, int K, int O, int dtype, int policy);
size_t dctn_eps_fwd_workspace_bytes(int C, int B, int H, int W, int Q, int K, int O,
                                    int dtype, int policy);
int dctn_eps_fwd(const void* x, const int64_t x_strides[5], const void* core, void* out,
                 void* workspace, size_t workspace_bytes,
                 int C, int B, int H, int W, int Q, int K, int O,
                 int dtype, int policy, void* stream);

/* Forward whose output is only needed for its statistics - replaces the `transform_in_slices(...)` +
 * `output.std(unbiased=False)` of dctn/eps.py:163-181 `make_eps_unit_empirical_output_std` (and the statistics of
 * dctn/eps_plus_linear.py:161-196): stats[0] += sum y, stats[1] += sum y^2 over every value of eps(core, x), in
 * float64, values as the output tensor would hold them (rounded to the storage dtype).  `stats`: two float64 on the
 * device, ACCUMULATED (the caller zeroes them once and feeds slice after slice; count = B*H'*W'*O per call).
 * The register-resident family folds the sums into the forward kernel's epilogue and stores nothing; the other
 * families write the slice into `workspace` (dctn_eps_fwd_stats_workspace_bytes) and reduce it in a second pass -
 * the output of the whole data set is never materialised either way. */
size_t dctn_eps_fwd_stats_workspace_bytes(int C, int B, int H, int W, int Q, int K, int O,
                                          int dtype, int policy);
int dctn_eps_fwd_stats(const void* x, const int64_t x_strides[5], const void* core, void* stats,
                       void* workspace, size_t workspace_bytes,
                       int C, int B, int H, int W, int Q, int K, int O,
                       int dtype, int policy, void* stream);

/* Autograd of the above (reference: torch autograd through the 4 path steps, dctn/training.py:81).
 *   dY    : (B, H', W', O) contiguous
 *   dX    : (C, B, H, W, Q) contiguous, or NULL when the input needs no gradient
 *   dCore : same layout as core, or NULL
 *   Both are OVERWRITTEN (not accumulated).  `workspace` must hold dctn_eps_bwd_workspace_bytes().
 */
size_t dctn_eps_bwd_workspace_bytes(int C, int B, int H, int W, int Q, int K, int O,
                                    int dtype, int policy, int need_dx, int need_dcore);
int dctn_eps_bwd(const void* x, const int64_t x_strides[5], const void* core, const void* dY,
                 void* dX, void* dCore, void* workspace, size_t workspace_bytes,
                 int C, int B, int H, int W, int Q, int K, int O,
                 int dtype, int policy, void* stream);

/* Training forward / backward that keep the forward's GEMM result - what the reference's autograd does: torch saves
 * the result G of path step (0,1) (dctn/eps.py:25-30, `core . K-R_0`) and its backward runs TWO GEMMs of that size
 * (d core, d K-R_0) plus a product of G with dY; without the buffer dctn_eps_bwd has to run the forward GEMM a third
 * time.  Only the input gradient needs it (dCore does not): pass it when dX will be asked for.
 *   dctn_eps_saved_bytes : size of the buffer for this shape, 0 when the shape's kernel family keeps nothing
 *                          (register-resident and generic families: nothing to save; also beyond 16 GiB)
 *   dctn_eps_fwd_save    : dctn_eps_fwd + fills `saved`.  Returns DCTN_SAVED (1) when `saved` was written,
 *                          DCTN_OK (0) when the forward ran but kept nothing (then call dctn_eps_bwd), < 0 on error
 *   dctn_eps_bwd_saved   : dctn_eps_bwd reading `saved` as written by dctn_eps_fwd_save for the SAME x, core, shape,
 *                          dtype and policy.  Same outputs as dctn_eps_bwd to rounding (the sums run in another order).
 * large-core float32 family: Z[(b,o)][w] float32 in row-quad-major order, B*H'*W' * Q^n1 * O * 4 bytes (cfg3a layer 2:
 * 416 MB); two-halves family: both Khatri-Rao halves and Z in the storage dtype. */
size_t dctn_eps_saved_bytes(int C, int B, int H, int W, int Q, int K, int O, int dtype, int policy);
int dctn_eps_fwd_save(const void* x, const int64_t x_strides[5], const void* core, void* out,
                      void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                      int C, int B, int H, int W, int Q, int K, int O,
                      int dtype, int policy, void* stream);
int dctn_eps_bwd_saved(const void* x, const int64_t x_strides[5], const void* core, const void* dY,
                       const void* saved, size_t saved_bytes,
                       void* dX, void* dCore, void* workspace, size_t workspace_bytes,
                       int C, int B, int H, int W, int Q, int K, int O,
                       int dtype, int policy, void* stream);

/* Forward of (EPS layer -> "b h w q -> b (h w q)" -> nn.Linear), the tail of EPSesPlusLinear.forward (reference:
 * dctn/eps_plus_linear.py:144-147), as ONE kernel: a workgroup holds all window positions of a few samples, so the
 * head's sum over (position, output) closes inside the workgroup - no second launch, no re-read of the features.
 *   features : (B, H'*W'*O) contiguous, OVERWRITTEN (the layer's output; the backward reads it)
 *   logits   : (B, Cout) contiguous, OVERWRITTEN = features @ head_weight^T + head_bias, products of the stored
 *              (bf16-rounded) features with the bf16 weight, float32 sums
 * bfloat16, contiguous x, the register-resident family with O in {2, 4}, Cout <= 16 and at most 768 window positions
 * per sample; DCTN_ERR_UNSUPPORTED otherwise (the caller then runs dctn_eps_fwd + dctn_linear_head_fwd). */
int dctn_eps_head_fwd(const void* x, const int64_t x_strides[5], const void* core, const void* head_weight,
                      const void* head_bias, void* features, void* logits,
                      int C, int B, int H, int W, int Q, int K, int O, int Cout,
                      int dtype, int policy, void* stream);

/* Backward of (EPS layer -> "b h w q -> b (h w q)" -> nn.Linear), the tail of
 * EPSesPlusLinear.forward (reference: dctn/eps_plus_linear.py:144-147), in one pass over x: the
 * kernel forms dY[b,h,w,o] = sum_c dLogits[b,c] * head_weight[c, (h*W'+w)*O + o] on the fly, so the
 * gradient of the features is never written to or read from HBM, and accumulates the head's own
 * gradients beside dCore.
 *   features    : (B, H'*W'*O) contiguous, the layer's forward output (input of the linear head)
 *   dLogits     : (B, Cout) contiguous;  head_weight : (Cout, H'*W'*O) contiguous
 *   dCore       : same layout as core;  dWeight : like head_weight, or NULL;  dBias : (Cout), or NULL
 *   all three OVERWRITTEN; `workspace` must hold dctn_eps_head_bwd_workspace_bytes().
 * bfloat16 (register-resident bf16 family) and float32 (register-resident exact-f32 family, O in {2, 4}, Cout <= 16);
 * the layer's input gets no gradient.  Returns DCTN_ERR_UNSUPPORTED for shapes outside those two families (the
 * caller then composes dctn_linear_head_bwd or library GEMMs with dctn_eps_bwd; there is no CPU fallback). */
size_t dctn_eps_head_bwd_workspace_bytes(int C, int B, int H, int W, int Q, int K, int O, int Cout,
                                         int dtype, int policy);
int dctn_eps_head_bwd(const void* x, const int64_t x_strides[5], const void* features,
                      const void* dLogits, const void* head_weight, void* dCore, void* dWeight,
                      void* dBias, void* workspace, size_t workspace_bytes,
                      int C, int B, int H, int W, int Q, int K, int O, int Cout,
                      int dtype, int policy, void* stream);

/* ------------------------------------------------------------------------------------------
 * ConvSBS — replaces dctn/conv_sbs.py:258-304 `ConvSBS.forward`
 *   x         : (C, B, H, W, q) with element strides x_strides[5] (channel c = x[c])
 *   n_cores   : number of cores in the string (string order)
 *   cores[c]  : device pointer to core c, shape (out_sizes[c], bond_sizes[c],
 *               bond_sizes[(c+1)%n], q, ..., q [C times]) contiguous (dctn/conv_sbs_spec.py:24-27,65-80)
 *   pos_h/pos_w : position of core c inside the window (min must be 0, dctn/align.py:18-19)
 *   out       : (B, H-max_h, W-max_w, prod(out_sizes)) contiguous, out dims in string order
 * ------------------------------------------------------------------------------------------ */
size_t dctn_convsbs_workspace_bytes(int n_cores, const int* out_sizes, const int* bond_sizes,
                                    int C, int B, int H, int W, int q,
                                    const int* pos_h, const int* pos_w, int dtype, int backward);
int dctn_convsbs_fwd(const void* x, const int64_t x_strides[5], const void* const* cores,
                     void* out, int n_cores, const int* out_sizes, const int* bond_sizes,
                     const int* pos_h, const int* pos_w,
                     int C, int B, int H, int W, int q,
                     void* workspace, size_t workspace_bytes, int dtype, void* stream);
/* dX (C,B,H,W,q contiguous) and dCores[c] (same layout as cores[c]) are OVERWRITTEN; either
 * dX or the whole dCores array may be NULL. */
int dctn_convsbs_bwd(const void* x, const int64_t x_strides[5], const void* const* cores,
                     const void* dY, void* dX, void* const* dCores,
                     int n_cores, const int* out_sizes, const int* bond_sizes,
                     const int* pos_h, const int* pos_w,
                     int C, int B, int H, int W, int q,
                     void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* A training forward can leave its forward states for the backward instead of having the backward recompute them
 * (the backward's own forward sweep is a quarter of its time): pass dctn_convsbs_fwd a workspace of at least
 * dctn_convsbs_saved_states_bytes(...) bytes (0: this string always recomputes - rings, many-valued cores, bonds above
 * 16, float64 / bf16).  dctn_convsbs_fwd then returns DCTN_SAVED (1) when it WROTE the states - hand that buffer,
 * untouched, to dctn_convsbs_bwd_saved - and DCTN_OK (0) when the forward ran on a kernel that keeps nothing (the
 * buffer is then uninitialised: call dctn_convsbs_bwd).  NULL / too small saved_states: exactly dctn_convsbs_bwd.  Replaces nothing in the reference (autograd keeps every intermediate there, dctn/conv_sbs.py:268-303). */
size_t dctn_convsbs_saved_states_bytes(int n_cores, const int* out_sizes, const int* bond_sizes, int C, int B, int H,
                                       int W, int q, const int* pos_h, const int* pos_w, int dtype);
int dctn_convsbs_bwd_saved(const void* x, const int64_t x_strides[5], const void* const* cores,
                           const void* dY, void* dX, void* const* dCores, int n_cores,
                           const int* out_sizes, const int* bond_sizes, const int* pos_h,
                           const int* pos_w, int C, int B, int H, int W, int q, void* workspace,
                           size_t workspace_bytes, const void* saved_states, size_t saved_states_bytes, int dtype,
                           void* stream);

/* Several strings of one layer at once - replaces the loop of `ManyConvSBS.forward` (dctn/conv_sbs.py:367-370: every
 * string contracts the same input) for layers whose strings are all nine-core strings of one bond <= 4 over the same
 * window positions (the reference's layers: two snakes through one 3 x 3 window, mnist.py:189-252): ONE forward launch,
 * ONE backward launch (+ the small reduction), dX written once, already summed over the strings.  Since version 401
 * also layers of TWO strings of the band family (largest bond 5..16, same band geometry: the bond-16 layer of
 * mnist.py:224-242): one forward launch, one backward launch + one tail kernel that sums the strings' shares of dX.
 *   cores / dCores : n_strings * n_cores pointers, string-major;  out_sizes, bond_sizes, pos_h, pos_w likewise
 *   outs / dYs     : one (B, H', W', prod(out_sizes of the string)) tensor per string
 * DCTN_ERR_UNSUPPORTED for any other layer: call dctn_convsbs_fwd / _bwd per string (and add the dX). */
size_t dctn_convsbs_many_workspace_bytes(int n_strings, int n_cores, const int* out_sizes, const int* bond_sizes,
                                         int C, int B, int H, int W, int q, const int* pos_h, const int* pos_w, int dtype);
int dctn_convsbs_many_fwd(const void* x, const int64_t x_strides[5], const void* const* cores, void* const* outs,
                          int n_strings, int n_cores, const int* out_sizes, const int* bond_sizes,
                          const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q, int dtype, void* stream);
int dctn_convsbs_many_bwd(const void* x, const int64_t x_strides[5], const void* const* cores, const void* const* dYs,
                          void* dX, void* const* dCores, int n_strings, int n_cores,
                          const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                          int C, int B, int H, int W, int q, void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * logmatmulexp — replaces dctn/logmatmulexp.py:5-14 (and the checkpointed :17-22; nothing of
 * size Theta*R*I is ever materialised here, forward or backward).
 *   batched: logA (batch, Theta, R), logB (batch, R, I), out (batch, Theta, I), all contiguous;
 *   the reference's strictly 2-D call is batch == 1.  stride_*_batch in elements (0 = broadcast).
 * ------------------------------------------------------------------------------------------ */
/* Products that are not tiny run as "exp -> GEMM on the matrix cores -> log" (one exp per input
 * element, float32 MFMA) and need scratch: `workspace` of dctn_logmatmulexp_workspace_bytes() bytes
 * (0 = this shape uses the direct kernels; workspace may then be NULL).  With workspace == NULL the
 * direct kernels are always used.  Results keep torch.logsumexp semantics either way: output tiles
 * / batch elements where the factored form is not safe are recomputed by the direct kernels. */
size_t dctn_logmatmulexp_workspace_bytes(int64_t batch, int Theta, int R, int I,
                                         int64_t strideA_batch, int64_t strideB_batch, int dtype);
int dctn_logmatmulexp_fwd(const void* logA, const void* logB, void* out,
                          void* workspace, size_t workspace_bytes,
                          int64_t batch, int Theta, int R, int I,
                          int64_t strideA_batch, int64_t strideB_batch,
                          int dtype, void* stream);
/* dA / dB are OVERWRITTEN; either may be NULL.  With a broadcast operand (stride 0) its
 * gradient is summed over the batch. */
int dctn_logmatmulexp_bwd(const void* logA, const void* logB, const void* out, const void* dOut,
                          void* dA, void* dB,
                          void* workspace, size_t workspace_bytes,
                          int64_t batch, int Theta, int R, int I,
                          int64_t strideA_batch, int64_t strideB_batch,
                          int dtype, void* stream);

/* Left fold over L square log-matrices per window — the loop
 * `reduce(logmatmulexp, matrices)` of small_experiments/logmatmulexp_benchmark/benchmark.py:30,
 * batched over windows (BASELINE config 5).
 *   mats : (Wn, L, D, D) contiguous;  out : (Wn, D, D)
 * Backward recomputes the prefix folds; `workspace` holds dctn_logmatmulexp_fold_workspace_bytes(). */
size_t dctn_logmatmulexp_fold_workspace_bytes(int64_t Wn, int L, int D, int dtype, int backward);
int dctn_logmatmulexp_fold_fwd(const void* mats, void* out, int64_t Wn, int L, int D,
                               int dtype, void* stream);
int dctn_logmatmulexp_fold_bwd(const void* mats, const void* dOut, void* dMats,
                               void* workspace, size_t workspace_bytes,
                               int64_t Wn, int L, int D, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Linear classifier head — replaces `self.linear(features)` of dctn/eps_plus_linear.py:147
 * (nn.Linear(H'*W'*Q, 10)) for skinny outputs.
 *   feat (B, F), weight (Cout, F), bias (Cout), out (B, Cout), all contiguous;
 *   float32, float64 and bf16 (float32 accumulation; float64 for float64), Cout <= 16, any F.  bf16 with F % 8 == 0 and
 *   16-byte aligned pointers runs the vectorised matrix-core kernels, everything else the scalar streaming kernels
 *   (DCTN_ERR_UNSUPPORTED only for Cout > 16: the host layer then uses the framework's library GEMM).
 *   Backward: dFeat (B, F), dWeight (Cout, F), dBias (Cout) are OVERWRITTEN; dFeat may be NULL;
 *   dWeight and dBias are produced together (dBias may be NULL).
 * ------------------------------------------------------------------------------------------ */
int dctn_linear_head_fwd(const void* feat, const void* weight, const void* bias, void* out,
                         int64_t B, int F, int Cout, int dtype, void* stream);
size_t dctn_linear_head_bwd_workspace_bytes(int64_t B, int F, int Cout, int dtype);
int dctn_linear_head_bwd(const void* feat, const void* weight, const void* dOut, void* dFeat,
                         void* dWeight, void* dBias, void* workspace, size_t workspace_bytes,
                         int64_t B, int F, int Cout, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Window statistics of the input feature map (reference: calc_scaling_factor,
 * dctn/dataset_loading.py:79-94 = make_windows (dctn/align.py:49-61) + RankOneTensorsBatch
 * .mean_over_batch / .var_over_batch (dctn/rank_one_tensor.py:53-100)).  For the rank-one tensor
 * T_w = (x)_n x_n[w,:] of every K x K window w of x (C,B,H,W,Q):
 *   sums[0] = sum_w sum(T_w)   = sum_w prod_n sum_q x_n[w,q]
 *   sums[1] = sum_w ||T_w||^2  = sum_w prod_n sum_q x_n[w,q]^2
 * `sums`: two float64 values on the device, OVERWRITTEN.  The K*K-fold window tensor of the
 * reference is never materialised.
 * ------------------------------------------------------------------------------------------ */
int dctn_window_stats(const void* x, const int64_t x_strides[5], void* sums,
                      int C, int B, int H, int W, int Q, int K, int dtype, void* stream);

/* The feature map itself on the device - replaces `phi_cos_sin_squared_1` as applied to the whole data set in
 * dctn/dataset_loading.py:33-36,63 (u -> (2 sin^2(pi u / 2), 2 cos^2(pi u / 2)), float32 arithmetic like the reference's):
 *   dctn_phi_window_stats : the two sums of dctn_window_stats straight from the RAW images (B, H, W) float32 contiguous,
 *       phi applied once per pixel inside the kernel: neither the expanded (1, B, H, W, 2) tensor nor the K*K stacked
 *       window copies of calc_scaling_factor (dataset_loading.py:79-94) exist.  sums: two float64, OVERWRITTEN.
 *       DCTN_ERR_UNSUPPORTED for images whose per-pixel table does not fit LDS (beyond ~97 x 97).
 *   dctn_phi_expand       : x[0, b, h, w, :] = scale * phi(images[b, h, w]) written once in the model's dtype
 *       (`dtype` of x: f32 / f64 / bf16), the scaling factor folded in; n_pixels = B * H * W. */
int dctn_phi_window_stats(const void* images, void* sums, int B, int H, int W, int K, void* stream);
int dctn_phi_expand(const void* images, void* x, int64_t n_pixels, float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Tail of a training iteration (reference: dctn/training.py:77-84 with
 * F.cross_entropy, the L2 regularisers of dctn/eps_plus_linear.py:149-159 and torch.optim.SGD).
 *   dctn_ce_loss_fwd : loss[0] (float32, OVERWRITTEN) = mean_b ( logsumexp(logits[b,:]) - logits[b, labels[b]] )
 *   dctn_ce_loss_bwd : dlogits = (softmax(logits) - onehot(labels)) * dloss[0] / B, dtype of logits
 *   dctn_sgd_l2_step : over one flat parameter buffer (n values; the first n_reg are regularised):
 *                      g = grads + 2*l2*w (regularised prefix), buf = first_step ? g : momentum*buf + g,
 *                      w -= lr*buf; sq_sum (optional, float32 array of dctn_sgd_l2_num_partials(n) slots,
 *                      OVERWRITTEN): slot b = workgroup b's part of the sum of w^2 over the prefix BEFORE the
 *                      update; their sum is the regulariser's value / its coefficient (stored, not
 *                      accumulated: no fill launch and no atomics in the iteration).
 *                      momentum_buf is float32 whatever the parameter dtype.
 * logits (B, C) contiguous, labels int64; dtypes DCTN_F32 / DCTN_BF16.  Rows labelled -100 (F.cross_entropy's default
 * ignore_index) add nothing to the loss, get a zero gradient row and do not count in the mean (n = the other rows; n = 0:
 * NaN, as torch).  Any other label outside [0, C) makes the loss and that sample's gradient row NaN (F.cross_entropy
 * raises on it).
 * ------------------------------------------------------------------------------------------ */
int dctn_ce_loss_fwd(const void* logits, const void* labels, void* loss, int64_t B, int C, int dtype, void* stream);
/* forward that also leaves dlogits_unit = (softmax - onehot) / n (logits' dtype): the backward for an incoming gradient of 1,
 * so that a caller who knows its gradient seed is 1 needs no second kernel */
int dctn_ce_loss_fwd_grad(const void* logits, const void* labels, void* loss, void* dlogits_unit,
                          int64_t B, int C, int dtype, void* stream);
int dctn_ce_loss_bwd(const void* logits, const void* labels, const void* dloss, void* dlogits,
                     int64_t B, int C, int dtype, void* stream);
int dctn_sgd_l2_num_partials(int64_t n);
int dctn_sgd_l2_step(void* params, const void* grads, void* momentum_buf, void* sq_sum, int64_t n, int64_t n_reg,
                     float lr, float momentum, float l2, int first_step, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Tensor-network inner product of two stacks of EPS cores - replaces the contractions of
 * dctn/epses_composition.py:21-58 `inner_product` (Gram of the first pair of cores over their input legs:
 * dctn/eps.py:106-112 `contract_on_input_dims`; that matrix absorbed into every input leg of the next core:
 * the N-operand einsum at :46-56; the closing dot product: dctn/eps.py:120-123), which the reference evaluates
 * every training iteration (dctn/eps_plus_linear.py:156-159).  Forward and backward are compositions of two
 * primitives over a contiguous tensor viewed as (pre, q, post):
 *   dctn_mode_product : out[pre, j, post] = sum_i in[pre, i, post] * M[i, j]      M: (q, q2) contiguous
 *   dctn_fiber_gram   : out[i, j] = sum_(pre, post) A[pre, i, post] * B[pre, j, post]     out: (qa, qb)
 * (Gram of two (rows, O) matrices: post = 1; dot product: qa = qb = 1.)  q, q2, qa, qb <= 32.  Outputs are
 * OVERWRITTEN, in the tensors' dtype (f32 / f64 / bf16 storage with f32 accumulation); sums are combined in a
 * fixed order (no atomics).  `out` must not alias `in`.
 * ------------------------------------------------------------------------------------------ */
int dctn_mode_product(const void* in, const void* M, void* out, int64_t pre, int q, int q2, int64_t post,
                      int dtype, void* stream);
size_t dctn_fiber_gram_workspace_bytes(int64_t pre, int qa, int qb, int64_t post, int dtype);
int dctn_fiber_gram(const void* A, const void* B, void* out, void* workspace, size_t workspace_bytes,
                    int64_t pre, int qa, int qb, int64_t post, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Latency-first gradient all-reduce over peer-mapped buffers (the reference has no
 * collective; the build's one collective is the mean of the parameter gradients, 58 KB .. 7.5 MB: latency-bound).
 * One-shot direct algorithm for the ranks of ONE node: every rank owns an uncached device block (flag lines + two
 * staging buffers), exported once as an IPC handle and mapped by every peer; per step ONE kernel per rank copies the
 * rank's values to its staging buffer, publishes its step number to every peer, waits (bounded, ~2 s) for theirs,
 * then sums every rank's staging buffer in rank order (float32 / float64 accumulation) and writes the (scaled)
 * result over `buf` in place: bitwise identical on all ranks.  Replayable from a HIP graph (the step counter lives
 * in device memory).  dctn_ar_create / _connect / _status / _destroy allocate, map or synchronise and are NOT
 * capturable; dctn_ar_allreduce only enqueues.  Host-side pairing: dctn_amd/ddp.py `DirectAllReducer`.
 *   create(world <= 16, rank, max_bytes)  -> opaque state;  export -> dctn_ar_handle_bytes() bytes for the peers;
 *   connect(handles of ALL ranks, rank-major);  allreduce(buf, n elements, dtype, average);
 *   status: 0 = every wait completed, r + 1 = a wait for rank r timed out (the results of that step are invalid).
 * Two-shot form for large buckets (version 402): rank r reduces chunk r only (reads (P - 1) N / P bytes), leaves it in
 * its result area and publishes a second flag; every rank copies the other chunks from their owners ((P - 1) N / P
 * more) instead of reading (P - 1) N bytes - bitwise the one-shot values.  dctn_ar_allreduce picks it for world >= 4
 * and >= 512 KiB; dctn_ar_allreduce_algo(..., algorithm: 0 = that rule, 1 = one-shot, 2 = two-shot) forces a form.
 * ------------------------------------------------------------------------------------------ */
size_t dctn_ar_handle_bytes(void);
int dctn_ar_create(int world, int rank, size_t max_bytes, void** state_out);
int dctn_ar_export(void* state, void* handle_out);
int dctn_ar_connect(void* state, const void* handles);
int dctn_ar_allreduce(void* state, void* buf, int64_t n, int dtype, int average, void* stream);
int dctn_ar_allreduce_algo(void* state, void* buf, int64_t n, int dtype, int average, int algorithm, void* stream);
int dctn_ar_status(void* state);
int dctn_ar_destroy(void* state);

#ifdef __cplusplus
}
#endif
#endif /* DCTN_AMD_H */
