// Tensor-network inner product of two stacks of EPS cores — the regulariser the reference evaluates every training
// iteration (dctn/epses_composition.py:21-58 through dctn/eps_plus_linear.py:156-159 and dctn/training.py:79).
// The reference's recursion is: Gram matrix of the first pair of cores over all their input legs
// (contract_on_input_dims, dctn/eps.py:106-112), absorbed into EVERY input leg of the next core of stack 1 (one
// N-operand einsum), recurse; the last pair ends in a dot product (dctn/eps.py:120-123).  All of it — and all of its
// backward — are two primitives over a tensor viewed as (pre, q, post):
//   mode_product : out[pre, j, post] = sum_i in[pre, i, post] * M[i, j]              (one leg through a small matrix)
//   fiber_gram   : out[i, j] = sum_(pre, post) A[pre, i, post] * B[pre, j, post]     (Gram / dot over all fibers)
// Gram of two (rows x O) core matrices = fiber_gram with post = 1; the final dot = fiber_gram with q = 1; the backward
// of mode_product is a mode_product with M^T plus a fiber_gram; the backward of fiber_gram is two mode_products.
// Both are bandwidth work (the 6 MiB cfg3a core is touched once per leg): coalesced streaming, float32 / float64
// accumulation on the vector ALU, deterministic (per-workgroup partial sums combined in a fixed order, no atomics).
#include "common.h"

namespace {

constexpr int TN_MAXQ = 32;
constexpr int TN_THREADS = 256;
constexpr int TN_TILE = 256;   // fibers staged per tile

template <typename S> struct TnAcc { typedef float type; };
template <> struct TnAcc<double> { typedef double type; };

__device__ __forceinline__ long long fiber_base(long long f, int q, long long post) {
  const long long pre = f / post, po = f - pre * post;
  return pre * (long long)q * post + po;
}

// ------------------------------------------------------------------------------------------------ mode product
template <typename S, int QP>
__global__ __launch_bounds__(TN_THREADS) void tn_mode_product_k(const S* __restrict__ in, const S* __restrict__ M,
                                                                S* __restrict__ out, long long nfib, int q, int q2,
                                                                long long post) {
  typedef typename TnAcc<S>::type A;
  __shared__ A sm[QP * QP];
  for (int e = threadIdx.x; e < QP * QP; e += TN_THREADS) {
    const int i = e / QP, j = e - i * QP;
    sm[e] = (i < q && j < q2) ? (A)M[i * q2 + j] : (A)0;
  }
  __syncthreads();
  for (long long f = (long long)blockIdx.x * TN_THREADS + threadIdx.x; f < nfib; f += (long long)gridDim.x * TN_THREADS) {
    const long long bi = fiber_base(f, q, post), bo = fiber_base(f, q2, post);
    A v[QP];
#pragma unroll
    for (int i = 0; i < QP; ++i) {   // clamped index + select: no branch between the loads
      const A ld = (A)in[bi + (i < q ? i : 0) * post];
      v[i] = i < q ? ld : (A)0;
    }
#pragma unroll
    for (int j = 0; j < QP; ++j) {
      if (j < q2) {
        A s = (A)0;
#pragma unroll
        for (int i = 0; i < QP; ++i) s += v[i] * sm[i * QP + j];
        out[bo + j * post] = (S)s;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ fiber Gram
// partial[blk][i*qb + j] = sum over the block's fibers of A[f, i] * B[f, j]
template <typename S>
__global__ __launch_bounds__(TN_THREADS) void tn_fiber_gram_k(const S* __restrict__ Ap, const S* __restrict__ Bp,
                                                              typename TnAcc<S>::type* __restrict__ partial,
                                                              long long nfib, int qa, int qb, long long post,
                                                              long long fib_per_block) {
  typedef typename TnAcc<S>::type A;
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_raw[];
  A* sA = reinterpret_cast<A*>(tn_raw);                 // [TN_TILE][qa]
  A* sB = sA + TN_TILE * qa;                            // [TN_TILE][qb]
  A* red = sB + TN_TILE * qb;                           // [TN_THREADS]
  const int tid = threadIdx.x, P = qa * qb;
  const int nsub = P <= TN_THREADS ? TN_THREADS / P : 1;
  const int npp = (P + TN_THREADS - 1) / TN_THREADS;    // pairs per thread when P > 256 (<= 4 for q <= 32)
  const bool active = P <= TN_THREADS ? tid < P * nsub : true;
  const int pair0 = P <= TN_THREADS ? tid % P : tid, sub = P <= TN_THREADS ? tid / P : 0;
  A acc[4] = {(A)0, (A)0, (A)0, (A)0};
  const long long f0 = (long long)blockIdx.x * fib_per_block;
  long long f1 = f0 + fib_per_block;
  if (f1 > nfib) f1 = nfib;
  for (long long t0 = f0; t0 < f1; t0 += TN_TILE) {
    const int nt = (int)((f1 - t0) < TN_TILE ? (f1 - t0) : TN_TILE);
    __syncthreads();
    // stage the tile: walk memory in its own order (post == 1: a fiber is contiguous; else fibers are contiguous)
    for (int e = tid; e < TN_TILE * qa; e += TN_THREADS) {
      const int fl = post == 1 ? e / qa : e % TN_TILE, i = post == 1 ? e % qa : e / TN_TILE;
      sA[fl * qa + i] = fl < nt ? (A)Ap[fiber_base(t0 + fl, qa, post) + i * post] : (A)0;
    }
    for (int e = tid; e < TN_TILE * qb; e += TN_THREADS) {
      const int fl = post == 1 ? e / qb : e % TN_TILE, j = post == 1 ? e % qb : e / TN_TILE;
      sB[fl * qb + j] = fl < nt ? (A)Bp[fiber_base(t0 + fl, qb, post) + j * post] : (A)0;
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int pair = pair0 + k * TN_THREADS;
        if (k < npp && pair < P) {
          const int i = pair / qb, j = pair - i * qb;
          A s = (A)0;
          for (int fl = sub; fl < TN_TILE; fl += nsub) s += sA[fl * qa + i] * sB[fl * qb + j];
          acc[k] += s;
        }
      }
    }
  }
  // sum the `nsub` row subsets of a pair in a fixed order
  A* dst = partial + (long long)blockIdx.x * P;
  if (P <= TN_THREADS) {
    __syncthreads();
    red[tid] = active ? acc[0] : (A)0;
    __syncthreads();
    if (tid < P) {
      A s = (A)0;
      for (int k = 0; k < nsub; ++k) s += red[k * P + tid];
      dst[tid] = s;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pair = pair0 + k * TN_THREADS;
      if (k < npp && pair < P) dst[pair] = acc[k];
    }
  }
}

template <typename S>
__global__ __launch_bounds__(TN_THREADS) void tn_gram_finish_k(const typename TnAcc<S>::type* __restrict__ partial,
                                                               S* __restrict__ out, int nblk, int P) {
  typedef typename TnAcc<S>::type A;
  for (int pair = threadIdx.x; pair < P; pair += TN_THREADS) {
    A s = (A)0;
    for (int b = 0; b < nblk; ++b) s += partial[(long long)b * P + pair];
    out[pair] = (S)s;
  }
}

int gram_blocks(long long nfib) {
  long long b = (nfib + 4 * TN_TILE - 1) / (4 * TN_TILE);   // >= 4 tiles per block
  if (b > 512) b = 512;
  return (int)(b < 1 ? 1 : b);
}

template <typename S, int QP>
void launch_mode(const void* in, const void* M, void* out, long long nfib, int q, int q2, long long post, hipStream_t st) {
  long long blocks = (nfib + TN_THREADS - 1) / TN_THREADS;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((tn_mode_product_k<S, QP>), dim3((unsigned)blocks), dim3(TN_THREADS), 0, st, (const S*)in,
                     (const S*)M, (S*)out, nfib, q, q2, post);
}

template <typename S>
int mode_dispatch(const void* in, const void* M, void* out, long long nfib, int q, int q2, long long post, hipStream_t st) {
  const int m = q > q2 ? q : q2;
  if (m <= 2) launch_mode<S, 2>(in, M, out, nfib, q, q2, post, st);
  else if (m <= 4) launch_mode<S, 4>(in, M, out, nfib, q, q2, post, st);
  else if (m <= 8) launch_mode<S, 8>(in, M, out, nfib, q, q2, post, st);
  else if (m <= 16) launch_mode<S, 16>(in, M, out, nfib, q, q2, post, st);
  else launch_mode<S, 32>(in, M, out, nfib, q, q2, post, st);
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

template <typename S>
int gram_launch(const void* A, const void* B, void* out, void* ws, long long nfib, int qa, int qb, long long post,
                hipStream_t st) {
  typedef typename TnAcc<S>::type Acc;
  const int nblk = gram_blocks(nfib), P = qa * qb;
  const long long per = ((nfib + nblk - 1) / nblk + TN_TILE - 1) / TN_TILE * TN_TILE;
  const size_t lds = ((size_t)TN_TILE * (qa + qb) + TN_THREADS) * sizeof(Acc);
  if (lds > 60 * 1024)
    (void)hipFuncSetAttribute((const void*)tn_fiber_gram_k<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((tn_fiber_gram_k<S>), dim3(nblk), dim3(TN_THREADS), lds, st, (const S*)A, (const S*)B, (Acc*)ws,
                     nfib, qa, qb, post, per);
  DCTN_CHECK_LAUNCH();
  hipLaunchKernelGGL((tn_gram_finish_k<S>), dim3(1), dim3(TN_THREADS), 0, st, (const Acc*)ws, (S*)out, nblk, P);
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

bool tn_shape_ok(int64_t pre, int qa, int qb, int64_t post) {
  return pre >= 1 && post >= 1 && qa >= 1 && qb >= 1 && qa <= TN_MAXQ && qb <= TN_MAXQ &&
         pre <= (1LL << 40) / post;
}

}  // namespace

extern "C" {

size_t dctn_fiber_gram_workspace_bytes(int64_t pre, int qa, int qb, int64_t post, int dtype) {
  if (!tn_shape_ok(pre, qa, qb, post)) return 0;
  const size_t acc = dtype == DCTN_F64 ? 8 : 4;
  return (size_t)gram_blocks(pre * post) * (size_t)(qa * qb) * acc + 256;
}

int dctn_fiber_gram(const void* A, const void* B, void* out, void* workspace, size_t workspace_bytes, int64_t pre,
                    int qa, int qb, int64_t post, int dtype, void* stream) {
  if (!A || !B || !out) return DCTN_ERR_NULL;
  if (!tn_shape_ok(pre, qa, qb, post)) return pre < 1 || post < 1 || qa < 1 || qb < 1 ? DCTN_ERR_BAD_SHAPE : DCTN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < dctn_fiber_gram_workspace_bytes(pre, qa, qb, post, dtype)) return DCTN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const long long nfib = pre * post;
  int rc;
  switch (dtype) {
    case DCTN_F32: rc = gram_launch<float>(A, B, out, workspace, nfib, qa, qb, post, st); break;
    case DCTN_F64: rc = gram_launch<double>(A, B, out, workspace, nfib, qa, qb, post, st); break;
    case DCTN_BF16: rc = gram_launch<bf16_t>(A, B, out, workspace, nfib, qa, qb, post, st); break;
    default: return DCTN_ERR_BAD_DTYPE;
  }
  if (rc == DCTN_OK) dctn_set_last_kernel("tn_fiber_gram");
  return rc;
}

int dctn_mode_product(const void* in, const void* M, void* out, int64_t pre, int q, int q2, int64_t post, int dtype,
                      void* stream) {
  if (!in || !M || !out) return DCTN_ERR_NULL;
  if (!tn_shape_ok(pre, q, q2, post)) return pre < 1 || post < 1 || q < 1 || q2 < 1 ? DCTN_ERR_BAD_SHAPE : DCTN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long long nfib = pre * post;
  int rc;
  switch (dtype) {
    case DCTN_F32: rc = mode_dispatch<float>(in, M, out, nfib, q, q2, post, st); break;
    case DCTN_F64: rc = mode_dispatch<double>(in, M, out, nfib, q, q2, post, st); break;
    case DCTN_BF16: rc = mode_dispatch<bf16_t>(in, M, out, nfib, q, q2, post, st); break;
    default: return DCTN_ERR_BAD_DTYPE;
  }
  if (rc == DCTN_OK) dctn_set_last_kernel("tn_mode_product");
  return rc;
}

}  // extern "C"
