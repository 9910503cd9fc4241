// logmatmulexp kernels — replace dctn/logmatmulexp.py:5-22.
//
// out[t,i] = log sum_r exp(A[t,r] + B[r,i]) with the exact max-shift semantics of
// torch.logsumexp (an infinite maximum is replaced by 0 before shifting, so all -inf rows give
// -inf and +inf gives +inf).  The (Theta,R,I) tensor of the reference is never materialised:
// forward streams over r twice (max, then sum of exps); backward recomputes the weights
// exp(A+B-out) on the fly (what logmatmulexp_lowmem buys with checkpointing, for free).
//
// The fold kernels run the loop `reduce(logmatmulexp, matrices)` of
// small_experiments/logmatmulexp_benchmark/benchmark.py:30 for one window per workgroup with
// every prefix kept in LDS (BASELINE config 5).
#include "common.h"

#include <math.h>

namespace {

template <typename A> __device__ __forceinline__ A neg_inf();
template <> __device__ __forceinline__ float neg_inf<float>() { return -INFINITY; }
template <> __device__ __forceinline__ double neg_inf<double>() { return -(double)INFINITY; }

__device__ __forceinline__ float xexp(float v) { return expf(v); }
__device__ __forceinline__ double xexp(double v) { return exp(v); }
__device__ __forceinline__ float xlog(float v) { return logf(v); }
__device__ __forceinline__ double xlog(double v) { return log(v); }
__device__ __forceinline__ float xmax(float a, float b) { return (a != a || b != b) ? (a + b) : fmaxf(a, b); }
__device__ __forceinline__ double xmax(double a, double b) { return (a != a || b != b) ? (a + b) : fmax(a, b); }
__device__ __forceinline__ bool xisinf(float v) { return isinf(v); }
__device__ __forceinline__ bool xisinf(double v) { return isinf(v); }

template <typename S, typename A>
__global__ void lme_fwd_k(const S* __restrict__ lA, const S* __restrict__ lB, S* __restrict__ out,
                          long long batch, int T, int R, int I, long long sA, long long sB) {
  const long long total = batch * T * I;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(idx % I);
    const long long t2 = idx / I;
    const int th = (int)(t2 % T);
    const long long b = t2 / T;
    const S* a = lA + b * sA + (long long)th * R;
    const S* bm = lB + b * sB + i;
    A m = neg_inf<A>();
    for (int r = 0; r < R; ++r) m = xmax(m, (A)a[r] + (A)bm[(long long)r * I]);
    const A mm = xisinf(m) ? A(0) : m;
    A s = A(0);
    for (int r = 0; r < R; ++r) s += xexp((A)a[r] + (A)bm[(long long)r * I] - mm);
    out[idx] = (S)(xlog(s) + mm);
  }
}

// dA[bA,t,r] = sum_{b in group} sum_i dO[b,t,i] exp(A[t,r] + B[r,i] - out[b,t,i])
template <typename S, typename A>
__global__ void lme_bwd_dA_k(const S* __restrict__ lA, const S* __restrict__ lB,
                             const S* __restrict__ out, const S* __restrict__ dO,
                             S* __restrict__ dA, long long batch, int T, int R, int I,
                             long long sA, long long sB) {
  const long long nbA = sA == 0 ? 1 : batch;
  const long long total = nbA * T * R;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx % R);
    const long long t2 = idx / R;
    const int th = (int)(t2 % T);
    const long long bA = t2 / T;
    const long long b0 = sA == 0 ? 0 : bA, b1 = sA == 0 ? batch : bA + 1;
    A acc = A(0);
    for (long long b = b0; b < b1; ++b) {
      const A av = (A)lA[b * sA + (long long)th * R + r];
      const S* brow = lB + b * sB + (long long)r * I;
      const S* orow = out + (b * T + th) * I;
      const S* grow = dO + (b * T + th) * I;
      for (int i = 0; i < I; ++i) acc += (A)grow[i] * xexp(av + (A)brow[i] - (A)orow[i]);
    }
    dA[idx] = (S)acc;
  }
}

template <typename S, typename A>
__global__ void lme_bwd_dB_k(const S* __restrict__ lA, const S* __restrict__ lB,
                             const S* __restrict__ out, const S* __restrict__ dO,
                             S* __restrict__ dB, long long batch, int T, int R, int I,
                             long long sA, long long sB) {
  const long long nbB = sB == 0 ? 1 : batch;
  const long long total = nbB * R * I;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(idx % I);
    const long long t2 = idx / I;
    const int r = (int)(t2 % R);
    const long long bB = t2 / R;
    const long long b0 = sB == 0 ? 0 : bB, b1 = sB == 0 ? batch : bB + 1;
    A acc = A(0);
    for (long long b = b0; b < b1; ++b) {
      const A bv = (A)lB[b * sB + (long long)r * I + i];
      for (int th = 0; th < T; ++th) {
        const long long oi = (b * T + th) * I + i;
        acc += (A)dO[oi] * xexp((A)lA[b * sA + (long long)th * R + r] + bv - (A)out[oi]);
      }
    }
    dB[idx] = (S)acc;
  }
}

// ------------------------------------------------------------------------------ fold kernels
// one workgroup (D*D threads) per window; thread (t,i) owns acc[t][i]
template <typename S, typename A>
__global__ void lme_fold_fwd_k(const S* __restrict__ mats, S* __restrict__ out, long long Wn,
                               int L, int D) {
  extern __shared__ __align__(16) unsigned char smem[];
  A* acc = reinterpret_cast<A*>(smem);  // [D][D+1]
  A* cur = acc + D * (D + 1);           // [D][D]
  const int tid = threadIdx.x;
  const int t = tid / D, i = tid - t * D;
  const int DD = D * D;
  for (long long w = blockIdx.x; w < Wn; w += gridDim.x) {
    const S* base = mats + w * (long long)L * DD;
    A v = (A)base[tid];
    for (int l = 1; l < L; ++l) {
      __syncthreads();
      acc[t * (D + 1) + i] = v;
      cur[tid] = (A)base[(long long)l * DD + tid];
      __syncthreads();
      A m = neg_inf<A>();
      for (int r = 0; r < D; ++r) m = xmax(m, acc[t * (D + 1) + r] + cur[r * D + i]);
      const A mm = xisinf(m) ? A(0) : m;
      A s = A(0);
      for (int r = 0; r < D; ++r) s += xexp(acc[t * (D + 1) + r] + cur[r * D + i] - mm);
      v = xlog(s) + mm;
    }
    out[w * DD + tid] = (S)v;
  }
}

template <typename S, typename A>
__global__ void lme_fold_bwd_k(const S* __restrict__ mats, const S* __restrict__ dOut,
                               S* __restrict__ dMats, long long Wn, int L, int D) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int DD = D * D, DP = D + 1;
  A* pre = reinterpret_cast<A*>(smem);  // [L][D][D+1]  prefix folds, pre[0] = mats[0]
  A* cur = pre + (size_t)L * D * DP;    // [D][D+1]     current right operand
  A* g = cur + D * DP;                  // [D][D+1]     gradient wrt the current prefix
  const int tid = threadIdx.x;
  const int t = tid / D, i = tid - t * D;
  for (long long w = blockIdx.x; w < Wn; w += gridDim.x) {
    const S* base = mats + w * (long long)L * DD;
    __syncthreads();
    A v = (A)base[tid];
    pre[t * DP + i] = v;
    for (int l = 1; l < L; ++l) {
      __syncthreads();
      cur[t * DP + i] = (A)base[(long long)l * DD + tid];
      __syncthreads();
      const A* ap = pre + (size_t)(l - 1) * D * DP;
      A m = neg_inf<A>();
      for (int r = 0; r < D; ++r) m = xmax(m, ap[t * DP + r] + cur[r * DP + i]);
      const A mm = xisinf(m) ? A(0) : m;
      A s = A(0);
      for (int r = 0; r < D; ++r) s += xexp(ap[t * DP + r] + cur[r * DP + i] - mm);
      v = xlog(s) + mm;
      pre[(size_t)l * D * DP + t * DP + i] = v;
    }
    __syncthreads();
    g[t * DP + i] = (A)dOut[w * DD + tid];
    for (int l = L - 1; l >= 1; --l) {
      __syncthreads();
      cur[t * DP + i] = (A)base[(long long)l * DD + tid];
      __syncthreads();
      const A* ap = pre + (size_t)(l - 1) * D * DP;  // left operand
      const A* op = pre + (size_t)l * D * DP;        // result of this step
      // thread (t,i) read as (theta=t, r=i) for dA and as (r=t, i=i) for dB
      A dA = A(0), dB = A(0);
      for (int k = 0; k < D; ++k) {
        // dA[t][r=i] += g[t][k] * exp(A[t][i] + B[i][k] - out[t][k])
        dA += g[t * DP + k] * xexp(ap[t * DP + i] + cur[i * DP + k] - op[t * DP + k]);
        // dB[r=t][i] += g[k][i] * exp(A[k][t] + B[t][i] - out[k][i])
        dB += g[k * DP + i] * xexp(ap[k * DP + t] + cur[t * DP + i] - op[k * DP + i]);
      }
      dMats[(w * L + l) * (long long)DD + tid] = (S)dB;
      __syncthreads();
      g[t * DP + i] = dA;
    }
    __syncthreads();
    dMats[(w * L) * (long long)DD + tid] = (S)g[t * DP + i];
  }
}

// ---------------------------------------------------------------- factored fold, D = 16, float32
// out[t,i] = a_t + b_i + log( sum_r exp(acc[t,r] - a_t) * exp(M[r,i] - b_i) ),  a_t = max_r acc[t,r],
// b_i = max_r M[r,i]: 2*256 exps + 256 logs + one 16x16x16 matrix product per step instead of 4096
// exps.  One wave per window; the product runs on v_mfma_f32_16x16x4_f32 (exact f32) in TRANSPOSED
// form, out^T = E2^T x E1^T, with the k index ordered r = 4*kq + s: the accumulator layout of one
// step (lane (t, g) holds acc[t][4g..4g+3]) is then exactly the B-operand layout of the next step,
// so the whole chain stays in registers (no LDS, no transposes); HBM sees each matrix once.
// When the dynamic range is unsafe for the factorisation (row / column range > 40, infinities,
// NaN) the wave takes the exact max-shifted path for that step (same semantics as torch.logsumexp).
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float wave16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float wave16_min(float v) {
  v = fminf(v, __shfl_xor(v, 16, 64));
  return fminf(v, __shfl_xor(v, 32, 64));
}

__global__ __launch_bounds__(256) void lme_fold16_fwd_mfma_k(const float* __restrict__ mats,
                                                             float* __restrict__ out, long long Wn,
                                                             int L) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  for (long long w = wave; w < Wn; w += nwaves) {
    const float* base = mats + w * (long long)L * 256;
    // state: v[s] = acc[t = c][4g + s]
    float4 v4 = *reinterpret_cast<const float4*>(base + c * 16 + 4 * g);
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
    float mrow[4];
    if (L > 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s) mrow[s] = base[256 + (4 * g + s) * 16 + c];
    }
    for (int l = 1; l < L; ++l) {
      float mnext[4];
      const float* nb = base + (long long)(l + 1 < L ? l + 1 : l) * 256;
#pragma unroll
      for (int s = 0; s < 4; ++s) mnext[s] = nb[(4 * g + s) * 16 + c];   // prefetch next matrix
      const float amax = wave16_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
      const float amin = wave16_min(fminf(fminf(v[0], v[1]), fminf(v[2], v[3])));
      const float bmax = wave16_max(fmaxf(fmaxf(mrow[0], mrow[1]), fmaxf(mrow[2], mrow[3])));
      const float bmin = wave16_min(fminf(fminf(mrow[0], mrow[1]), fminf(mrow[2], mrow[3])));
      const bool safe = (amax - amin <= 40.f) && (bmax - bmin <= 40.f);   // false for inf / NaN
      if (__all(safe)) {
        f32x4_t S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
          S = __builtin_amdgcn_mfma_f32_16x16x4f32(expf(mrow[s] - bmax), expf(v[s] - amax), S, 0, 0, 0);
        // S[reg] = sum for (i = 4g + reg, t = c); b_i lives in the lanes whose c equals i
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) v[reg] = logf(S[reg]) + amax + __shfl(bmax, 4 * g + reg, 64);
      } else {
        // exact path: gather acc[t][0..15] from the 4 lanes of column t, read M[r][4g..4g+3] directly
        float arow[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) arow[r] = __shfl(v[r & 3], c + 16 * (r >> 2), 64);
        const float* mb = base + (long long)l * 256;
        float res[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int i = 4 * g + reg;
          float m = neg_inf<float>();
          for (int r = 0; r < 16; ++r) m = xmax(m, arow[r] + mb[r * 16 + i]);
          const float mm = xisinf(m) ? 0.f : m;
          float sacc = 0.f;
          for (int r = 0; r < 16; ++r) sacc += expf(arow[r] + mb[r * 16 + i] - mm);
          res[reg] = logf(sacc) + mm;
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) v[reg] = res[reg];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) mrow[s] = mnext[s];
    }
    *reinterpret_cast<float4*>(out + w * 256 + c * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

unsigned grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 65536 * 4) g = 65536 * 4;
  if (g < 1) g = 1;
  return (unsigned)g;
}

template <typename S, typename A>
int fwd_launch(const void* lA, const void* lB, void* out, long long batch, int T, int R, int I,
               long long sA, long long sB, hipStream_t st) {
  const long long total = batch * T * I;
  hipLaunchKernelGGL((lme_fwd_k<S, A>), dim3(grid_for(total, 256)), dim3(256), 0, st,
                     (const S*)lA, (const S*)lB, (S*)out, batch, T, R, I, sA, sB);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("logmatmulexp_fwd");
  return DCTN_OK;
}

template <typename S, typename A>
int bwd_launch(const void* lA, const void* lB, const void* out, const void* dO, void* dA, void* dB,
               long long batch, int T, int R, int I, long long sA, long long sB, hipStream_t st) {
  if (dA) {
    const long long total = (sA == 0 ? 1 : batch) * T * R;
    hipLaunchKernelGGL((lme_bwd_dA_k<S, A>), dim3(grid_for(total, 256)), dim3(256), 0, st,
                       (const S*)lA, (const S*)lB, (const S*)out, (const S*)dO, (S*)dA, batch, T, R,
                       I, sA, sB);
    DCTN_CHECK_LAUNCH();
  }
  if (dB) {
    const long long total = (sB == 0 ? 1 : batch) * R * I;
    hipLaunchKernelGGL((lme_bwd_dB_k<S, A>), dim3(grid_for(total, 256)), dim3(256), 0, st,
                       (const S*)lA, (const S*)lB, (const S*)out, (const S*)dO, (S*)dB, batch, T, R,
                       I, sA, sB);
    DCTN_CHECK_LAUNCH();
  }
  dctn_set_last_kernel("logmatmulexp_bwd");
  return DCTN_OK;
}

template <typename S, typename A>
int fold_fwd_launch(const void* mats, void* out, long long Wn, int L, int D, hipStream_t st) {
  const size_t lds = ((size_t)D * (D + 1) + (size_t)D * D) * sizeof(A);
  const unsigned grid = (unsigned)(Wn < 256 * 64 ? Wn : 256 * 64);
  hipLaunchKernelGGL((lme_fold_fwd_k<S, A>), dim3(grid), dim3(D * D), lds, st, (const S*)mats,
                     (S*)out, Wn, L, D);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("logmatmulexp_fold_fwd");
  return DCTN_OK;
}

template <typename S, typename A>
int fold_bwd_launch(const void* mats, const void* dOut, void* dMats, long long Wn, int L, int D,
                    hipStream_t st) {
  const size_t lds = ((size_t)(L + 2) * D * (D + 1)) * sizeof(A);
  if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
  (void)hipFuncSetAttribute((const void*)lme_fold_bwd_k<S, A>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const unsigned grid = (unsigned)(Wn < 256 * 32 ? Wn : 256 * 32);
  hipLaunchKernelGGL((lme_fold_bwd_k<S, A>), dim3(grid), dim3(D * D), lds, st, (const S*)mats,
                     (const S*)dOut, (S*)dMats, Wn, L, D);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("logmatmulexp_fold_bwd");
  return DCTN_OK;
}

}  // namespace

extern "C" {

int dctn_logmatmulexp_fwd(const void* logA, const void* logB, void* out, int64_t batch, int Theta,
                          int R, int I, int64_t strideA_batch, int64_t strideB_batch, int dtype,
                          void* stream) {
  if (!logA || !logB || !out) return DCTN_ERR_NULL;
  if (batch < 1 || Theta < 1 || R < 1 || I < 1) return DCTN_ERR_BAD_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case DCTN_F32:
      return fwd_launch<float, float>(logA, logB, out, batch, Theta, R, I, strideA_batch, strideB_batch, st);
    case DCTN_F64:
      return fwd_launch<double, double>(logA, logB, out, batch, Theta, R, I, strideA_batch, strideB_batch, st);
    case DCTN_BF16:
      return fwd_launch<bf16_t, float>(logA, logB, out, batch, Theta, R, I, strideA_batch, strideB_batch, st);
  }
  return DCTN_ERR_BAD_DTYPE;
}

int dctn_logmatmulexp_bwd(const void* logA, const void* logB, const void* out, const void* dOut,
                          void* dA, void* dB, int64_t batch, int Theta, int R, int I,
                          int64_t strideA_batch, int64_t strideB_batch, int dtype, void* stream) {
  if (!logA || !logB || !out || !dOut) return DCTN_ERR_NULL;
  if (batch < 1 || Theta < 1 || R < 1 || I < 1) return DCTN_ERR_BAD_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case DCTN_F32:
      return bwd_launch<float, float>(logA, logB, out, dOut, dA, dB, batch, Theta, R, I, strideA_batch, strideB_batch, st);
    case DCTN_F64:
      return bwd_launch<double, double>(logA, logB, out, dOut, dA, dB, batch, Theta, R, I, strideA_batch, strideB_batch, st);
    case DCTN_BF16:
      return bwd_launch<bf16_t, float>(logA, logB, out, dOut, dA, dB, batch, Theta, R, I, strideA_batch, strideB_batch, st);
  }
  return DCTN_ERR_BAD_DTYPE;
}

size_t dctn_logmatmulexp_fold_workspace_bytes(int64_t Wn, int L, int D, int dtype, int backward) {
  (void)Wn; (void)L; (void)D; (void)dtype; (void)backward;
  return 256;  // every prefix fold lives in LDS
}

int dctn_logmatmulexp_fold_fwd(const void* mats, void* out, int64_t Wn, int L, int D, int dtype,
                               void* stream) {
  if (!mats || !out) return DCTN_ERR_NULL;
  if (Wn < 1 || L < 1 || D < 1) return DCTN_ERR_BAD_SHAPE;
  if (D > 32) return DCTN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCTN_F32 && D == 16 && ((uintptr_t)mats % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
    long long blocks = (Wn + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(lme_fold16_fwd_mfma_k, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)mats,
                       (float*)out, (long long)Wn, L);
    DCTN_CHECK_LAUNCH();
    dctn_set_last_kernel("logmatmulexp_fold_fwd_mfma16");
    return DCTN_OK;
  }
  switch (dtype) {
    case DCTN_F32: return fold_fwd_launch<float, float>(mats, out, Wn, L, D, st);
    case DCTN_F64: return fold_fwd_launch<double, double>(mats, out, Wn, L, D, st);
    case DCTN_BF16: return fold_fwd_launch<bf16_t, float>(mats, out, Wn, L, D, st);
  }
  return DCTN_ERR_BAD_DTYPE;
}

int dctn_logmatmulexp_fold_bwd(const void* mats, const void* dOut, void* dMats, void* workspace,
                               size_t workspace_bytes, int64_t Wn, int L, int D, int dtype,
                               void* stream) {
  (void)workspace; (void)workspace_bytes;
  if (!mats || !dOut || !dMats) return DCTN_ERR_NULL;
  if (Wn < 1 || L < 1 || D < 1) return DCTN_ERR_BAD_SHAPE;
  if (D > 32) return DCTN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case DCTN_F32: return fold_bwd_launch<float, float>(mats, dOut, dMats, Wn, L, D, st);
    case DCTN_F64: return fold_bwd_launch<double, double>(mats, dOut, dMats, Wn, L, D, st);
    case DCTN_BF16: return fold_bwd_launch<bf16_t, float>(mats, dOut, dMats, Wn, L, D, st);
  }
  return DCTN_ERR_BAD_DTYPE;
}

}  // extern "C"
