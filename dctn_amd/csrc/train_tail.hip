// The tail of a training iteration: everything around the contraction
// path that the reference's loop runs per iteration (dctn/training.py:77-84) — cross-entropy on the
// (batch, classes) logits, the L2 regulariser (dctn/eps_plus_linear.py:149-159,
// dctn/epses_composition.py:144-146) and the optimizer update — is a few kilobytes of data, but as
// library calls it is ~30 launches of 4-5 us each, 3x the time of the forward + backward kernels of
// BASELINE config 2.  Three kernels replace them:
//   ce_fwd_k   : mean cross-entropy of bf16/f32 logits (max-shifted log-sum-exp, float32; rows labelled -100 are skipped
//                and do not count in the mean, as F.cross_entropy's default ignore_index)
//   ce_bwd_k   : dLogits = (softmax - onehot) * dLoss / n, in the logits' dtype
//   sgd_l2_k   : over ONE flat buffer holding all parameters: g += 2 * l2 * w on the regularised
//                prefix, buf = momentum * buf + g, w -= lr * buf; the regulariser's value
//                sum w^2 is left as one partial sum per workgroup (float32 master arithmetic, storage dtype kept)
#include "common.h"

namespace {

constexpr long long DCTN_CE_IGNORE = -100;   // torch.nn.functional.cross_entropy's default ignore_index

// Number of rows that count in the mean: every workgroup scans all labels itself (8 bytes per row out of L2; the mean's
// divisor is needed before the first gradient row is written, and a workgroup cannot wait for the others)
__device__ __forceinline__ float ce_count_valid(const long long* __restrict__ labels, long long B, float* red) {
  float n = 0.f;
  for (long long b = threadIdx.x; b < B; b += blockDim.x) n += labels[b] != DCTN_CE_IGNORE ? 1.f : 0.f;
  for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
  __syncthreads();
  float total = 0.f;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) total += red[k];
  __syncthreads();   // `red` is reused by the caller
  return total;
}

// SINGLE: the whole batch in one workgroup of 1024 threads — the result is stored, not accumulated, so no fill
// precedes the kernel (one graph node instead of two; these kernels are launch-latency, not work)
template <typename S, bool SINGLE>
__global__ __launch_bounds__(SINGLE ? 1024 : 256) void ce_fwd_k(const S* __restrict__ logits,
                                                                const long long* __restrict__ labels,
                                                                float* __restrict__ loss, S* __restrict__ dunit,
                                                                long long B, int C) {
  // dunit (optional): (softmax - onehot) / n, the gradient of the mean loss for an incoming gradient of 1; n = the number
  // of rows whose label is not DCTN_CE_IGNORE (F.cross_entropy's default ignore_index): such rows add nothing to the loss,
  // get a zero gradient row and do not count in the mean
  __shared__ float red[16];
  const float n_valid = ce_count_valid(labels, B, red);
  float part = 0.f;
  for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (long long)gridDim.x * blockDim.x) {
    const S* row = logits + b * C;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, (float)row[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf((float)row[c] - m);
    const long long y = labels[b];
    // any OTHER label outside [0, C) poisons the loss and its row of the gradient with NaN: F.cross_entropy raises /
    // device-asserts on such labels; a plausible-looking number would hide the bug
    const bool skip = y == DCTN_CE_IGNORE, valid = y >= 0 && y < C;
    part += skip ? 0.f : valid ? (m + logf(s)) - (float)row[y] : __builtin_nanf("");
    if (dunit) {
      const float inv = 1.f / s, scale = skip ? 0.f : valid ? 1.f / n_valid : __builtin_nanf("");
      for (int c = 0; c < C; ++c)
        dunit[b * C + c] = (S)((expf((float)row[c] - m) * inv - (c == y ? 1.f : 0.f)) * scale);
    }
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) total += red[k];
    if (SINGLE)
      loss[0] = total / n_valid;   // (no row counted: 0 / 0 = NaN, as torch)
    else
      atomicAdd(loss, total / n_valid);
  }
}

template <typename S>
__global__ __launch_bounds__(256) void ce_bwd_k(const S* __restrict__ logits, const long long* __restrict__ labels,
                                                const float* __restrict__ dloss, S* __restrict__ dlogits, long long B,
                                                int C) {
  __shared__ float red[16];
  const float scale = dloss[0] / ce_count_valid(labels, B, red);
  for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (long long)gridDim.x * blockDim.x) {
    const S* row = logits + b * C;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, (float)row[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf((float)row[c] - m);
    const float inv = 1.f / s;
    const long long y = labels[b];
    const float rs = y == DCTN_CE_IGNORE ? 0.f : (y >= 0 && y < C) ? scale : __builtin_nanf("");   // ignored: zero row; invalid: NaN row
    for (int c = 0; c < C; ++c) {
      const float p = expf((float)row[c] - m) * inv;
      dlogits[b * C + c] = (S)((p - (c == y ? 1.f : 0.f)) * rs);
    }
  }
}

template <typename S>
__global__ __launch_bounds__(256) void sgd_l2_k(S* __restrict__ w, const S* __restrict__ g, float* __restrict__ buf,
                                                float* __restrict__ sq_sum, long long n, long long n_reg, float lr,
                                                float momentum, float l2, int first_step) {
  __shared__ float red[4];
  float part = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float wi = (float)w[i];
    float gi = (float)g[i];
    if (i < n_reg) {
      gi += 2.f * l2 * wi;
      part += wi * wi;
    }
    const float bi = first_step ? gi : momentum * buf[i] + gi;   // torch.optim.SGD: the first step copies g
    buf[i] = bi;
    w[i] = (S)(wi - lr * bi);
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  // one slot per workgroup, stored (not accumulated): no fill before the kernel, no atomics, a fixed summation order;
  // the caller adds the dctn_sgd_l2_num_partials(n) slots when it wants the regulariser's value
  if (threadIdx.x == 0 && sq_sum) sq_sum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

unsigned blocks_for(long long n) {
  long long b = (n + 255) / 256;
  if (b > 1024) b = 1024;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

static int ce_fwd_launch(const void* logits, const void* labels, void* loss, void* dunit, int64_t B, int C, int dtype,
                         hipStream_t st) {
  if (!logits || !labels || !loss) return DCTN_ERR_NULL;
  if (B < 1 || C < 1) return DCTN_ERR_BAD_SHAPE;
  if (dtype != DCTN_F32 && dtype != DCTN_BF16) return DCTN_ERR_BAD_DTYPE;
  if (B <= 8192) {   // one workgroup, no fill
    if (dtype == DCTN_F32)
      hipLaunchKernelGGL((ce_fwd_k<float, true>), dim3(1), dim3(1024), 0, st, (const float*)logits, (const long long*)labels, (float*)loss, (float*)dunit, (long long)B, C);
    else
      hipLaunchKernelGGL((ce_fwd_k<bf16_t, true>), dim3(1), dim3(1024), 0, st, (const bf16_t*)logits, (const long long*)labels, (float*)loss, (bf16_t*)dunit, (long long)B, C);
    DCTN_CHECK_LAUNCH();
    return DCTN_OK;
  }
  if (dctn_zero_async(loss, sizeof(float), st) != DCTN_OK) return DCTN_ERR_LAUNCH;
  const dim3 g(blocks_for(B)), b(256);
  if (dtype == DCTN_F32)
    hipLaunchKernelGGL((ce_fwd_k<float, false>), g, b, 0, st, (const float*)logits, (const long long*)labels, (float*)loss, (float*)dunit, (long long)B, C);
  else
    hipLaunchKernelGGL((ce_fwd_k<bf16_t, false>), g, b, 0, st, (const bf16_t*)logits, (const long long*)labels, (float*)loss, (bf16_t*)dunit, (long long)B, C);
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

int dctn_ce_loss_fwd(const void* logits, const void* labels, void* loss, int64_t B, int C, int dtype, void* stream) {
  return ce_fwd_launch(logits, labels, loss, nullptr, B, C, dtype, (hipStream_t)stream);
}

int dctn_ce_loss_fwd_grad(const void* logits, const void* labels, void* loss, void* dlogits_unit, int64_t B, int C,
                          int dtype, void* stream) {
  if (!dlogits_unit) return DCTN_ERR_NULL;
  return ce_fwd_launch(logits, labels, loss, dlogits_unit, B, C, dtype, (hipStream_t)stream);
}

int dctn_ce_loss_bwd(const void* logits, const void* labels, const void* dloss, void* dlogits, int64_t B, int C,
                     int dtype, void* stream) {
  if (!logits || !labels || !dloss || !dlogits) return DCTN_ERR_NULL;
  if (B < 1 || C < 1) return DCTN_ERR_BAD_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(blocks_for(B)), b(256);
  switch (dtype) {
    case DCTN_F32: hipLaunchKernelGGL(ce_bwd_k<float>, g, b, 0, st, (const float*)logits, (const long long*)labels, (const float*)dloss, (float*)dlogits, (long long)B, C); break;
    case DCTN_BF16: hipLaunchKernelGGL(ce_bwd_k<bf16_t>, g, b, 0, st, (const bf16_t*)logits, (const long long*)labels, (const float*)dloss, (bf16_t*)dlogits, (long long)B, C); break;
    default: return DCTN_ERR_BAD_DTYPE;
  }
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

int dctn_sgd_l2_num_partials(int64_t n) { return n < 1 ? 0 : (int)blocks_for(n); }

int dctn_sgd_l2_step(void* params, const void* grads, void* momentum_buf, void* sq_sum, int64_t n, int64_t n_reg,
                     float lr, float momentum, float l2, int first_step, int dtype, void* stream) {
  if (!params || !grads || !momentum_buf) return DCTN_ERR_NULL;
  if (n < 1 || n_reg < 0 || n_reg > n) return DCTN_ERR_BAD_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype != DCTN_F32 && dtype != DCTN_BF16) return DCTN_ERR_BAD_DTYPE;
  const dim3 g(blocks_for(n)), b(256);
  if (dtype == DCTN_F32)
    hipLaunchKernelGGL(sgd_l2_k<float>, g, b, 0, st, (float*)params, (const float*)grads, (float*)momentum_buf, (float*)sq_sum, (long long)n, (long long)n_reg, lr, momentum, l2, first_step);
  else
    hipLaunchKernelGGL(sgd_l2_k<bf16_t>, g, b, 0, st, (bf16_t*)params, (const bf16_t*)grads, (float*)momentum_buf, (float*)sq_sum, (long long)n, (long long)n_reg, lr, momentum, l2, first_step);
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

}  // extern "C"
