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
#include "q2_common.h"

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
                             long long sA, long long sB, const int* __restrict__ only_batch) {
  const long long nbA = sA == 0 ? 1 : batch;
  const long long total = nbA * T * R;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx % R);
    const long long t2 = idx / R;
    const int th = (int)(t2 % T);
    const long long bA = t2 / T;
    if (only_batch && !only_batch[bA]) continue;   // (never combined with a broadcast operand)
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
                             long long sA, long long sB, const int* __restrict__ only_batch) {
  const long long nbB = sB == 0 ? 1 : batch;
  const long long total = nbB * R * I;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(idx % I);
    const long long t2 = idx / I;
    const int r = (int)(t2 % R);
    const long long bB = t2 / R;
    if (only_batch && !only_batch[bB]) continue;
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
                               S* __restrict__ dMats, long long Wn, int L, int D,
                               const int* __restrict__ only_flagged) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int DD = D * D, DP = D + 1;
  A* pre = reinterpret_cast<A*>(smem);  // [L][D][D+1]  prefix folds, pre[0] = mats[0]
  A* cur = pre + (size_t)L * D * DP;    // [D][D+1]     current right operand
  A* g = cur + D * DP;                  // [D][D+1]     gradient wrt the current prefix
  const int tid = threadIdx.x;
  const int t = tid / D, i = tid - t * D;
  auto window = [&](long long w) {
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
  };
  if (!only_flagged) {
    for (long long w = blockIdx.x; w < Wn; w += gridDim.x) window(w);
    return;
  }
  // flagged windows only: the workgroup reads blockDim.x flags at a time (one coalesced load; a flag per turn of a
  // per-window loop was a dependent memory round trip per window: 72 us for 692 224 clean windows) and walks those set
  __shared__ int nlist;
  __shared__ int list[1024];
  const int nthr = blockDim.x;
  for (long long base0 = (long long)blockIdx.x * nthr; base0 < Wn; base0 += (long long)gridDim.x * nthr) {
    __syncthreads();
    if (tid == 0) nlist = 0;
    __syncthreads();
    if (base0 + tid < Wn && only_flagged[base0 + tid] != 0) list[atomicAdd(&nlist, 1)] = tid;   // (order is irrelevant: windows are independent)
    __syncthreads();
    const int n = nlist;
    for (int k = 0; k < n; ++k) window(base0 + list[k]);
  }
}

// ---------------------------------------------------------------- factored fold, D = 16, float32
// out[t,i] = a_t + b_i + log( sum_r exp(acc[t,r] - a_t) * exp(M[r,i] - b_i) ),  a_t = max_r acc[t,r],
// b_i = max_r M[r,i]: 2*256 exps + 256 logs + one 16x16x16 matrix product per step instead of 4096
// exps.  One wave per window; the product runs on v_mfma_f32_16x16x4_f32 (exact f32) in TRANSPOSED
// form, out^T = E2^T x E1^T, with the k index ordered r = 4*kq + s: the accumulator layout of one
// step (lane (t, g) holds acc[t][4g..4g+3]) is then exactly the B-operand layout of the next step,
// so the chain of prefixes stays in registers; HBM sees each matrix once.
//
// The kernels are bounded by VALU issue, not by HBM or the matrix core, so the arithmetic around the
// product is kept minimal: the whole chain runs in the base-2 log domain (state and matrices scaled
// by log2 e once, the result by ln 2 once) so that every exp / log is the single native
// v_exp_f32 / v_log_f32; the two 16-lane-stride reductions use v_permlane16/32_swap instead of LDS
// shuffles; and safety of the factorisation is judged on its result: a step is accepted when every
// S[t,i] >= 2^-100 (terms the shifts could have flushed are then < 2^-26 of the sum); NaN / +inf /
// all -inf rows make S NaN and fail the test.  A rejected step takes the exact max-shifted path
// (same semantics as torch.logsumexp) in the forward kernel; the backward kernel flags the window
// for the exact recomputing kernel.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef int lme_int2v __attribute__((ext_vector_type(2)));

constexpr float LME_LOG2E = 1.4426950408889634f;
constexpr float LME_LN2 = 0.6931471805599453f;
constexpr float LME_SMIN = 7.888609052210118e-31f;   // 2^-100

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32

// v_max_f32 / v_max3_f32 as they are: fmaxf() on a value that came through a lane swap is preceded by a canonicalising
// v_max(x, x) per operand (8 of a reduction's 17 instructions).  A NaN operand may be dropped here - harmless: the NaN
// still reaches the product through its own exp and the step is rejected on S.
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// max over the four lanes (c, 0..3) that share c; result in all four
__device__ __forceinline__ float wave16_max(float v) {
  int iv = __float_as_int(v);
  lme_int2v r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
  v = vmax(__int_as_float(r[0]), __int_as_float(r[1]));
  iv = __float_as_int(v);
  r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
  return vmax(__int_as_float(r[0]), __int_as_float(r[1]));
}

// max over the 16 lanes of a row (lanes (0..15, g)), in every lane: four rotate steps
// (one v_max_f32_dpp per step; the builtin form is v_mov 0 + v_mov_dpp + v_max.  A DPP operand needs two wait states behind
// the vector instruction that wrote it, and the compiler does not look into the text.)
__device__ __forceinline__ float row16_max(float v) {
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
      : "+v"(v));
  return v;
}

constexpr int LME_PST = 20;                 // padded row stride of a 16x16 tile in LDS
constexpr int LME_TILE = 16 * LME_PST;      // floats per tile

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Matrices are fetched with one coalesced 16-byte load per lane (lane (c, g): row c, columns
// 4g..4g+3) and turned into the k-major operand layout (lane (c, g): rows 4g..4g+3 of column c) by a
// b128 write + 4 b32 reads of a per-wave LDS tile (conflict-free at row stride 20).
__device__ __forceinline__ void lds_transpose16(float* scr, int c, int g, const float4& row, float (&col)[4]) {
  wave_lds_sync();   // earlier reads of the tile are done
  *reinterpret_cast<float4*>(scr + c * LME_PST + 4 * g) = row;
  wave_lds_sync();
#pragma unroll
  for (int s = 0; s < 4; ++s) col[s] = scr[(4 * g + s) * LME_PST + c];
}

__device__ __forceinline__ bool lme_step_ok(const f32x4_t& S) {
  return (S[0] >= LME_SMIN) & (S[1] >= LME_SMIN) & (S[2] >= LME_SMIN) & (S[3] >= LME_SMIN);
}

// The running prefix is carried in SCALED form, not in the log domain: P[t][r] = a_t + log2 E[t][r] with max_r E[t][r] = 1
// (the "scaled forward algorithm").  E IS the left operand of the next product (2^(P - a_t)), so a step needs no exp for
// it and no log for its result:
//   S = E x EM (EM = 2^(M - b_i), b_i = max_r M[r][i]);   T[t][i] = S[t][i] * 2^(b_i - b_0);   E' = T / max_i T,
//   a' = a + b_0 + log2 max_i T:   per lane 4 + 1 v_exp, 1 v_rcp, 1 v_log per step instead of 8 v_exp + 4 v_log
// (these are quarter-rate instructions: 40 % of the vector time of the log-domain version).  Entries more than 2^-126
// below their row maximum flush to zero exactly where the log-domain version's 2^(P - a_t) did; the LAST step still forms
// its result as a + b_i + log2 S, so nothing flushes on the way out.  2^(b_i - b_0) overflows only when column maxima of
// one matrix differ by 2^127: the step is then rejected like any other the factorisation cannot represent.
constexpr float LME_BIG = 1.0e38f;

__device__ __forceinline__ float max4(const float (&v)[4]) { return vmax(vmax3(v[0], v[1], v[2]), v[3]); }

// The matrices stay in the NATURAL-log domain as loaded; log2(e) rides in the multiply-add in front of every v_exp_f32
// (2^(m log2e - b log2e)) instead of a multiply per loaded value.

// natural-log row (state layout) -> scaled form (a in the base-2 domain); a row of -inf gets shift 0 and E = 0, a row
// holding +inf or NaN gets E = NaN: either way the next product rejects the step
__device__ __forceinline__ void lme_to_scaled(const float (&v)[4], float (&E)[4], float& a) {
  const float m = wave16_max(max4(v));
  a = (m == neg_inf<float>()) ? 0.f : m * LME_LOG2E;
#pragma unroll
  for (int s = 0; s < 4; ++s) E[s] = ex2(__builtin_fmaf(v[s], LME_LOG2E, -a));
}

// S = E x 2^((M - b) log2e) for the k-major matrix mcol (natural log); nb = -b_i log2e in the lanes whose c equals i
__device__ __forceinline__ f32x4_t lme_step16s(const float (&E)[4], const float (&mcol)[4], float& nb) {
  nb = wave16_max(max4(mcol)) * -LME_LOG2E;
  f32x4_t S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s)
    S = __builtin_amdgcn_mfma_f32_16x16x4f32(ex2(__builtin_fmaf(mcol[s], LME_LOG2E, nb)), E[s], S, 0, 0, 0);
  return S;
}

// The same product with ONE shift for the whole matrix (nb = -max M log2e, the same number in every lane): EM is formed
// once, in the ROW layout the matrix was loaded in (row c, columns 4g..4g+3 - the layout the way back multiplies with), and
// taken through the per-wave LDS tile into the k-major operand layout - where the column shifts needed the exponentials of
// both layouts (8 v_exp_f32 per lane and step) and 2^(b_i - b_0) factors behind the product.  Columns whose largest entry
// lies 2^-126 below the matrix maximum flush to zero; their S is 0 and the step is rejected like any other the
// factorisation cannot represent.
__device__ __forceinline__ float lme_matrix_shift(const float (&mrow)[4]) {
  return row16_max(wave16_max(max4(mrow))) * -LME_LOG2E;
}
__device__ __forceinline__ f32x4_t lme_step16g(float* scr, int c, int g, const float (&E)[4], const float (&emrow)[4]) {
  float emcol[4];
  lds_transpose16(scr, c, g, make_float4(emrow[0], emrow[1], emrow[2], emrow[3]), emcol);
  f32x4_t S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) S = __builtin_amdgcn_mfma_f32_16x16x4f32(emcol[s], E[s], S, 0, 0, 0);
  return S;
}
// E' = S / max_i S and the row's new shift: a' = a + b + log2 max_i S  (da = log2 rmax - nb)
__device__ __forceinline__ bool lme_rescale_g(const f32x4_t& S, float (&En)[4], float& rinv) {
  const float T[4] = {S[0], S[1], S[2], S[3]};
  const float rmax = wave16_max(max4(T));
  rinv = __builtin_amdgcn_rcpf(rmax);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) En[reg] = T[reg] * rinv;
  return rmax < LME_BIG;
}

// E' and the row's new shift (base-2 domain) from S; returns false (in any lane of the row) when 2^(b_i - b_0) overflowed
__device__ __forceinline__ bool lme_rescale(const f32x4_t& S, float nb, int g, float (&En)[4], float& da) {
  const float nref = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, nb)));   // -b_0 log2e
  const float cb = ex2(nref - nb);
  float T[4];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) T[reg] = S[reg] * __shfl(cb, 4 * g + reg, 64);
  const float rmax = wave16_max(max4(T));
  const float rinv = __builtin_amdgcn_rcpf(rmax);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) En[reg] = T[reg] * rinv;
  da = lg2(rmax) - nref;
  return rmax < LME_BIG;
}

// The max-shifted sums of a whole window in the natural-log domain (same semantics as torch.logsumexp), rolled loops:
// the way out for a window with a step the factorisation cannot represent.  The scaled prefix has already flushed
// the entries far below their row maximum, and in such a step they may be the ones that matter - so the window starts
// again from its first matrix.  acc[t][r] comes from the 4 lanes of row t, M[r][4g..4g+3] from memory.
__device__ __forceinline__ void lme_exact_window(const float* __restrict__ base, float* __restrict__ outw, int L, int c, int g) {
  const float4 v4 = *reinterpret_cast<const float4*>(base + c * 16 + 4 * g);
  float vn[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll 1
  for (int l2 = 1; l2 < L; ++l2) {
    const float* mb = base + (long long)l2 * 256;
    float res[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = 4 * g + reg;
      float m = neg_inf<float>();
#pragma unroll 1
      for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) m = xmax(m, __shfl(vn[s2], c + 16 * kq, 64) + mb[(4 * kq + s2) * 16 + i]);
      }
      const float mm = xisinf(m) ? 0.f : m;
      float sacc = 0.f;
#pragma unroll 1
      for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) sacc += expf(__shfl(vn[s2], c + 16 * kq, 64) + mb[(4 * kq + s2) * 16 + i] - mm);
      }
      res[reg] = logf(sacc) + mm;
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) vn[reg] = res[reg];
  }
  *reinterpret_cast<float4*>(outw + c * 16 + 4 * g) = make_float4(vn[0], vn[1], vn[2], vn[3]);
}

// 16-byte global accesses of the fold kernels.  NT: the streaming ("nt") policy, for folds whose matrices are far larger
// than the 256 MiB last-level cache - every byte is used once per pass, so it should not displace anything; measured on
// the access pattern alone (tools/probes/window_stream2.hip, 692 224 windows of 9 matrices): 4.97 -> 5.39 TB/s.
typedef float lme_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 lme_ld4(const float* p) {
  if (NT) {
    const lme_v4f v = __builtin_nontemporal_load(reinterpret_cast<const lme_v4f*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
  }
  return *reinterpret_cast<const float4*>(p);
}
template <bool NT>
__device__ __forceinline__ void lme_st4(float* p, const float4& q) {
  if (NT) {
    const lme_v4f v = {q.x, q.y, q.z, q.w};
    __builtin_nontemporal_store(v, reinterpret_cast<lme_v4f*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = q;
  }
}

// One step of the forward fold on the state (E, a); `last`: the result goes out instead of becoming the next state.
// Returns false when the step was rejected (uniform over the wave).
template <bool NT = false>
__device__ __forceinline__ bool lme_fwd_step(float* scr, int c, int g, const float4& qm, float (&E)[4], float& a, bool last,
                                             float* __restrict__ outp) {
  float mcol[4];
  lds_transpose16(scr, c, g, qm, mcol);   // mcol[s] = M_l[4g + s][c]
  float nb, En[4], da;
  const f32x4_t S = lme_step16s(E, mcol, nb);
  bool ok = lme_step_ok(S);
  if (!last) ok = lme_rescale(S, nb, g, En, da) && ok;
  if (!__all(ok)) return false;
  if (last) {
    // S[reg] = sum for (i = 4g + reg, t = c); b_i lives in the lanes whose c equals i
    float r[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) r[reg] = (lg2(S[reg]) + a - __shfl(nb, 4 * g + reg, 64)) * LME_LN2;
    lme_st4<NT>(outp, make_float4(r[0], r[1], r[2], r[3]));
  } else {
    a += da;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) E[reg] = En[reg];
  }
  return true;
}

// Slot l holds matrix l of the wave's current window until step l has used it and is refilled at once with matrix l of the
// wave's NEXT window - the steps are unrolled, so every slot is a fixed set of registers (a rotating queue of four cost 12
// v_mov per step) and up to L KiB per wave are in flight across the window boundary.
//
// 2 <= L <= 16, the chain length a template parameter (round 5; what the backward kernel below taught): no step is
// conditional, window bases are scalars in buffer descriptors (one vector register of addresses), and a window whose
// steps the factorisation cannot represent is only NOTED in the loop (one bit per iteration in a per-wave LDS mask) and
// redone in the log domain behind it - a second way round the loop with memory operations in it joins the loop top with
// another counter state and costs every window its wait counts.
constexpr int LME_FWD_ITMAX = 4096;   // iterations a wave can note (the host grows the grid beyond that)

template <int L, bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(L <= 9 ? 4 : (L <= 12 ? 3 : 2)))) void lme_fold16_fwd_k(
    const float* __restrict__ mats, float* __restrict__ out, long long Wn) {
  __shared__ __align__(16) float scratch[4][LME_TILE];
  __shared__ unsigned rej[4][LME_FWD_ITMAX / 32];
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* scr = scratch[wv];
  const long long wave = (long long)blockIdx.x * 4 + wv;
  const long long nwaves = (long long)gridDim.x * 4;
  if (wave >= Wn) return;
  for (int i = lane; i < LME_FWD_ITMAX / 32; i += 64) rej[wv][i] = 0u;
  typedef __attribute__((ext_vector_type(4))) unsigned lme_u4;
  constexpr int AUX = NT ? 2 : 0;   // nt
  const unsigned voff = (unsigned)(c * 16 + 4 * g) * 4u;
  constexpr unsigned wbytes = (unsigned)L * 1024u;
  auto load4 = [&](float (&dst)[4], __amdgpu_buffer_rsrc_t rs, unsigned soff) {
    const lme_u4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned bits = q[e];   // (through a scalar: see the backward kernel)
      dst[e] = __uint_as_float(bits);
    }
  };
  float Mq[L][4];
  {
    const __amdgpu_buffer_rsrc_t r0 = q2_make_rsrc(mats + wave * (long long)L * 256, wbytes);
#pragma unroll
    for (int l = 0; l < L; ++l) load4(Mq[l], r0, (unsigned)l * 1024u);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the loop top joins the back edge with an empty counter state
  int it = 0;
  for (long long w = wave; w < Wn; w += nwaves, ++it) {
    const long long wn = w + nwaves < Wn ? w + nwaves : w;   // (the last window re-reads itself)
    const __amdgpu_buffer_rsrc_t rn = q2_make_rsrc(mats + wn * (long long)L * 256, wbytes);
    float E[4], a;
    lme_to_scaled(Mq[0], E, a);
    load4(Mq[0], rn, 0u);
    bool ok = true;
    float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 1; l < L; ++l) {
      float mcol[4], nb;
      lds_transpose16(scr, c, g, make_float4(Mq[l][0], Mq[l][1], Mq[l][2], Mq[l][3]), mcol);   // mcol[s] = M_l[4g + s][c]
      load4(Mq[l], rn, (unsigned)l * 1024u);
      const f32x4_t S = lme_step16s(E, mcol, nb);
      ok = ok & lme_step_ok(S);
      if (l + 1 < L) {
        float En[4], da;
        ok = lme_rescale(S, nb, g, En, da) & ok;
        a += da;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) E[reg] = En[reg];
      } else {
        // S[reg] = sum for (i = 4g + reg, t = c); b_i lives in the lanes whose c equals i
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) r[reg] = (lg2(S[reg]) + a - __shfl(nb, 4 * g + reg, 64)) * LME_LN2;
      }
    }
    {
      const lme_u4 v = {__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]), __float_as_uint(r[3])};
      __builtin_amdgcn_raw_buffer_store_b128(v, q2_make_rsrc(out + w * 256, 1024u), voff, 0u, AUX);
    }
    if (!__all(ok) && lane == 0) rej[wv][it >> 5] |= 1u << (it & 31);
  }
  // the noted windows, in the log domain from their first matrix (the scaled prefix has flushed what lies far below a row
  // maximum, and in such a step that may be what matters)
  wave_lds_sync();
  for (int wd = 0; wd * 32 < it; ++wd) {
    unsigned m = rej[wv][wd];   // (wave-uniform)
    while (m) {
      const int b = __builtin_ctz(m);
      m &= m - 1;
      const long long w = wave + (long long)(wd * 32 + b) * nwaves;
      lme_exact_window(mats + w * (long long)L * 256, out + w * 256, L, c, g);
    }
  }
}

// any L: four matrices of the wave's stream in flight in a rotating queue
__global__ __launch_bounds__(256) void lme_fold16_fwd_mfma_k(const float* __restrict__ mats,
                                                             float* __restrict__ out, long long Wn,
                                                             int L) {
  __shared__ __align__(16) float scratch[4][LME_TILE];
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  float* scr = scratch[threadIdx.x >> 6];
  const int gl_off = c * 16 + 4 * g;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  long long pw = wave;   // stream position of the next matrix to request: window pw, matrix pl
  int pl = 0;
  auto fetch = [&]() {
    const bool in = pw < Wn;
    const float4 q = *reinterpret_cast<const float4*>(mats + ((in ? pw : wave) * (long long)L + (in ? pl : 0)) * 256 + gl_off);
    if (++pl == L) { pl = 0; pw += nwaves; }
    return q;
  };
  if (wave >= Wn) return;
  float4 q0 = fetch(), q1 = fetch(), q2 = fetch(), q3 = fetch();
  for (long long w = wave; w < Wn; w += nwaves) {
    const float4 v4 = q0;
    q0 = q1; q1 = q2; q2 = q3; q3 = fetch();
    if (L == 1) {   // nothing to fold
      *reinterpret_cast<float4*>(out + w * 256 + gl_off) = v4;
      continue;
    }
    float E[4], a;
    {
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
      lme_to_scaled(v, E, a);
    }
    bool good = true;
    for (int l = 1; l < L; ++l) {
      const float4 qm = q0;
      q0 = q1; q1 = q2; q2 = q3; q3 = fetch();   // (a rejected window keeps the stream's place)
      if (good) good = lme_fwd_step(scr, c, g, qm, E, a, l == L - 1, out + w * 256 + gl_off);
    }
    if (!good) lme_exact_window(mats + w * (long long)L * 256, out + w * 256, L, c, g);
  }
}

// Factored backward of the same fold (D = 16, float32), one wave per window.  Pass 1 recomputes the scaled prefixes
// E_1..E_{L-2} and keeps, per step, E, the row factor 1 / max_i S and EM = 2^(M - b) (in the place of the matrix) in
// registers; pass 2 walks back with, per step (EP = E_{l-1}, EM, S of step l):
//   H = G / S              dP = EP .* (H x EM^T)          dM = EM .* (EP^T x H)
// (the shifts 2^(a_t), 2^b cancel between the numerator and S), i.e. two 16x16x16 products on v_mfma_f32_16x16x4_f32 plus
// 4 v_rcp_f32 per lane - every exp of the step was taken in pass 1.  S itself is not kept: S_l = E_l / Rinv_l (E_l is S
// scaled to a row maximum of 1), so H = G * Rinv * rcp(E); only the last step, which is not rescaled, keeps its S.
// dP^T comes out of the matrix core in the state layout (lane (t, g) holds columns 4g..4g+3 of row t) so it feeds the
// next step directly; the operands of dM need t on the k index, so EP and H take one trip through a per-wave LDS tile.
//
// b is ONE shift per matrix here (the forward kernel keeps one per column): EM is formed once, in the row layout the
// way back multiplies with, and transposed through LDS for the recomputed product - 4 v_exp_f32 per lane and step
// where the column shifts took 8 + 1 and a ds_bpermute per value.  A step whose matrix has columns 2^-100 below its
// maximum is rejected here (S = 0) even where the forward accepted it: the window is flagged and left to the exact
// recomputing kernel, which rewrites its gradients behind this launch - it still walks back here on whatever its steps
// produced, because a second way round the loop, taken or not, costs every window its wait counts (below).
//
// What the memory side is built to (round 5; each item measured on cfg5, 692 224 windows, 2.87 -> 2.50 ms):
//  * the chain length is a TEMPLATE parameter: with `if (l < L)` around unrolled steps the compiler merged the counter
//    state of the skipped path at every join and each step of the way back waited for the load it had just issued
//    (`s_waitcnt vmcnt(1)`: a memory round trip per step).  Any other edge into the loop top does the same (a `continue`
//    for flagged windows; the prologue, which is therefore drained by hand).
//  * loads and stores share ONE in-order counter on this part (vmcnt): a wait for a load also waits for every store issued
//    before it.  The window's dM tiles therefore wait in LDS (the lane reads back what it wrote) and go out in one burst
//    BEHIND the last load of the next window's matrices: every wait of the next window then covers only stores that are a
//    whole window old.  (With the stores in the steps the first steps of a window stood behind the acknowledgement of
//    stores a few hundred cycles old.)
//  * window bases are scalars (the wave index through readfirstlane) in raw buffer descriptors: one vector register of
//    addresses for the whole kernel, ~25 registers and a third of the vector instructions fewer than with 64-bit pointers
//    per access - 124 registers at L = 9, four waves per SIMD without a spill (a spill is a scratch access: it shares vmcnt
//    and turned every wait into vmcnt(0)).
constexpr int LME_BWD_WAVES = 4;

template <int L, bool NT>
__global__ __launch_bounds__(64 * LME_BWD_WAVES) __attribute__((amdgpu_waves_per_eu(L <= 9 ? 4 : (L <= 12 ? 3 : 2)))) void lme_fold16_bwd_mfma_k(
    const float* __restrict__ mats, const float* __restrict__ dOut, float* __restrict__ dMats,
    int* __restrict__ flags, long long Wn) {
  constexpr int LO = L <= 15 ? 2 : 3;   // tiles LO .. L - 1 are staged in LDS (64 KiB per workgroup), the last ones formed stay in registers
  constexpr int NST = L > LO ? L - LO : 1;
  __shared__ __align__(16) float scratch[LME_BWD_WAVES][2][LME_TILE];
  __shared__ __align__(16) float stage[LME_BWD_WAVES][NST][256];
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // a scalar: so are the window bases
  float* scrE = scratch[wv][0];
  float* scrH = scratch[wv][1];
  float* stg = &stage[wv][0][0] + 4 * lane - LO * 256;
  const int st_off = c * LME_PST + 4 * g;       // state layout: row c, columns 4g..4g+3
  const long long wave = (long long)blockIdx.x * LME_BWD_WAVES + wv;
  const long long nwaves = (long long)gridDim.x * LME_BWD_WAVES;
  if (wave >= Wn) return;
  // Slot l: matrix l of the current window (natural log) until pass 1 has turned it into EM_l, EM_l until pass 2 has used
  // it, then matrix l of the wave's NEXT window at once - the loads of a window are spread over the previous window's
  // way back instead of standing in front of its first step.  Matrix 1, whose slot frees last and is needed first, waits in N1.
  float Mr[L][4], N1[4] = {0.f, 0.f, 0.f, 0.f}, G[4];
  typedef __attribute__((ext_vector_type(4))) unsigned lme_u4;
  constexpr int AUX = NT ? 2 : 0;   // nt
  const unsigned voff = (unsigned)(c * 16 + 4 * g) * 4u;
  constexpr unsigned wbytes = (unsigned)L * 1024u;
  auto load4 = [&](float (&dst)[4], __amdgpu_buffer_rsrc_t rs, unsigned soff) {
    const lme_u4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX);
    // (through scalars: __builtin_bit_cast on a vector ELEMENT reads element 0 for every index with this compiler)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned bits = q[e];
      dst[e] = __uint_as_float(bits);
    }
  };
  auto store4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned soff, const float4& q) {
    const lme_u4 v = {__float_as_uint(q.x), __float_as_uint(q.y), __float_as_uint(q.z), __float_as_uint(q.w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, AUX);
  };
  {
    const __amdgpu_buffer_rsrc_t r0 = q2_make_rsrc(mats + wave * (long long)L * 256, wbytes);
#pragma unroll
    for (int l = 0; l < L; ++l) load4(Mr[l], r0, (unsigned)l * 1024u);
    load4(G, q2_make_rsrc(dOut + wave * 256, 1024u), 0u);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the loop top joins the back edge with an empty counter state
  for (long long w = wave; w < Wn; w += nwaves) {
    const long long wn = w + nwaves < Wn ? w + nwaves : w;   // (the last window re-reads itself)
    const __amdgpu_buffer_rsrc_t rn = q2_make_rsrc(mats + wn * (long long)L * 256, wbytes);    // the next window's matrices
    const __amdgpu_buffer_rsrc_t rd = q2_make_rsrc(dMats + w * (long long)L * 256, wbytes);   // this window's gradients
    // ---------------- pass 1
    float Es[L][4];        // E_l in the state layout
    float Rinv[L];         // 1 / max_i S of step l
    float Slast[4];        // S of the last step
    {
      float a0;
      lme_to_scaled(Mr[0], Es[0], a0);
      load4(Mr[0], rn, 0u);
    }
    bool ok = true;
#pragma unroll
    for (int l = 1; l < L; ++l) {
      const float nb = lme_matrix_shift(Mr[l]);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Mr[l][reg] = ex2(__builtin_fmaf(Mr[l][reg], LME_LOG2E, nb));   // EM[r = c][i = 4g + reg]
      const f32x4_t S = lme_step16g(scrE, c, g, Es[l - 1], Mr[l]);
      ok = ok & lme_step_ok(S);
      if (l + 1 < L) {
        ok = lme_rescale_g(S, Es[l], Rinv[l]) & ok;
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Slast[reg] = S[reg];
      }
    }
    const bool all_ok = __all(ok);
    if (lane == 0) flags[w] = all_ok ? 0 : 1;
    // ---------------- pass 2: walk back
    if (L > 1) load4(N1, rn, 1024u);
    float4 dmr[LO];
#pragma unroll
    for (int l = L - 1; l >= 1; --l) {
      const float* EM = Mr[l];
      const float* EP = Es[l - 1];
      float H[4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        H[s] = l + 1 < L ? G[s] * Rinv[l] * __builtin_amdgcn_rcpf(Es[l][s]) : G[s] * __builtin_amdgcn_rcpf(Slast[s]);   // G / S
      wave_lds_sync();   // the previous step's reads of the scratch tiles are done
      *reinterpret_cast<float4*>(scrE + st_off) = make_float4(EP[0], EP[1], EP[2], EP[3]);
      *reinterpret_cast<float4*>(scrH + st_off) = make_float4(H[0], H[1], H[2], H[3]);
      f32x4_t T1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) T1 = __builtin_amdgcn_mfma_f32_16x16x4f32(EM[s], H[s], T1, 0, 0, 0);
      wave_lds_sync();
      float EPc[4], Hc[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        EPc[s] = scrE[(4 * g + s) * LME_PST + c];                    // EP[t = 4g+s][r = c]
        Hc[s] = scrH[(4 * g + s) * LME_PST + c];                     // H[t = 4g+s][i = c]
      }
      f32x4_t T2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) T2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Hc[s], EPc[s], T2, 0, 0, 0);
      const float4 dm = make_float4(T2[0] * EM[0], T2[1] * EM[1], T2[2] * EM[2], T2[3] * EM[3]);
      if (l >= LO) *reinterpret_cast<float4*>(stg + l * 256) = dm;
      else dmr[l] = dm;
#pragma unroll
      for (int s = 0; s < 4; ++s) G[s] = T1[s] * EP[s];
      if (l >= 2) load4(Mr[l], rn, (unsigned)l * 1024u);
    }
    const float4 g0 = make_float4(G[0], G[1], G[2], G[3]);
    load4(G, q2_make_rsrc(dOut + wn * 256, 1024u), 0u);   // the next window's, needed behind its pass 1
    asm volatile("" ::: "memory");                          // every load of the next window is issued: now the burst
#pragma unroll
    for (int l = L - 1; l >= LO; --l) store4(rd, (unsigned)l * 1024u, *reinterpret_cast<const float4*>(stg + l * 256));
#pragma unroll
    for (int l = LO - 1; l >= 1; --l)
      if (l < L) store4(rd, (unsigned)l * 1024u, dmr[l]);
    store4(rd, 0u, g0);
    if (L > 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s) Mr[L > 1 ? 1 : 0][s] = N1[s];
    }
  }
}

// Second tier of the factored backward: the windows the kernel above flagged, with one shift per COLUMN of every matrix
// (the forward kernel's factorisation: it represents matrices whose columns lie hundreds of nats apart, which one shift per
// matrix flushes).  One wave per flagged window, nothing pipelined - a handful of windows in a batch at most; a window it
// can represent has its flag cleared, the others are left to the exact recomputing kernel, whose float32 log-domain
// prefixes carry the error of torch's own float32 logsumexp (1e-4 relative at magnitudes of a few hundred).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1))) void lme_fold16_bwd_cols_k(
    const float* __restrict__ mats, const float* __restrict__ dOut, float* __restrict__ dMats,
    int* __restrict__ flags, long long Wn, int L) {
  constexpr int LMAX = 16;
  __shared__ __align__(16) float scratch[4][2][LME_TILE];
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  float* scrE = scratch[wv][0];
  float* scrH = scratch[wv][1];
  const int st_off = c * LME_PST + 4 * g;
  const int gl_off = c * 16 + 4 * g;
  const long long nwaves = (long long)gridDim.x * 4;
  // a wave reads 64 flags at a time (a flag per turn was a dependent round trip per window: 118 us for 692 224 clean windows)
  for (long long w0 = ((long long)blockIdx.x * 4 + wv) * 64; w0 < Wn; w0 += nwaves * 64) {
   unsigned long long todo = __ballot(w0 + lane < Wn && flags[w0 + lane] != 0);
   while (todo) {
    const long long w = w0 + __builtin_ctzll(todo);
    todo &= todo - 1;
    const float* base = mats + w * (long long)L * 256 + gl_off;
    float Mr[LMAX][4], Es[LMAX][4], Ss[LMAX][4], G[4];
    auto load4 = [&](float (&dst)[4], const float* src) {
      const float4 q = *reinterpret_cast<const float4*>(src);
      dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
    };
#pragma unroll
    for (int l = 0; l < LMAX; ++l)
      if (l < L) load4(Mr[l], base + (long long)l * 256);
    load4(G, dOut + w * 256 + gl_off);
    float a0;
    lme_to_scaled(Mr[0], Es[0], a0);
    bool ok = true;
#pragma unroll
    for (int l = 1; l < LMAX; ++l) {
      if (l < L) {
        float mcol[4], nb, da;
        lds_transpose16(scrE, c, g, make_float4(Mr[l][0], Mr[l][1], Mr[l][2], Mr[l][3]), mcol);
        const f32x4_t S = lme_step16s(Es[l - 1], mcol, nb);
        ok = ok & lme_step_ok(S);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          Ss[l][reg] = S[reg];
          Mr[l][reg] = ex2(__builtin_fmaf(Mr[l][reg], LME_LOG2E, __shfl(nb, 4 * g + reg, 64)));   // EM[r = c][i = 4g + reg]
        }
        if (l + 1 < L) ok = lme_rescale(S, nb, g, Es[l], da) & ok;
      }
    }
    if (!__all(ok)) continue;   // stays flagged
#pragma unroll
    for (int l = LMAX - 1; l >= 1; --l) {
      if (l < L) {
        const float* EM = Mr[l];
        const float* EP = Es[l - 1];
        float H[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) H[s] = G[s] * __builtin_amdgcn_rcpf(Ss[l][s]);
        wave_lds_sync();
        *reinterpret_cast<float4*>(scrE + st_off) = make_float4(EP[0], EP[1], EP[2], EP[3]);
        *reinterpret_cast<float4*>(scrH + st_off) = make_float4(H[0], H[1], H[2], H[3]);
        f32x4_t T1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) T1 = __builtin_amdgcn_mfma_f32_16x16x4f32(EM[s], H[s], T1, 0, 0, 0);
        wave_lds_sync();
        float EPc[4], Hc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          EPc[s] = scrE[(4 * g + s) * LME_PST + c];
          Hc[s] = scrH[(4 * g + s) * LME_PST + c];
        }
        f32x4_t T2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) T2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Hc[s], EPc[s], T2, 0, 0, 0);
        *reinterpret_cast<float4*>(dMats + (w * L + l) * 256 + gl_off) =
            make_float4(T2[0] * EM[0], T2[1] * EM[1], T2[2] * EM[2], T2[3] * EM[3]);
#pragma unroll
        for (int s = 0; s < 4; ++s) G[s] = T1[s] * EP[s];
      }
    }
    *reinterpret_cast<float4*>(dMats + (w * L) * 256 + gl_off) = make_float4(G[0], G[1], G[2], G[3]);
    if (lane == 0) flags[w] = 0;
   }
  }
}

// workgroups of `threads` that are resident at once on the whole device
long long resident_blocks(const void* fn, int threads) {
  int per_cu = 0, dev = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return (long long)per_cu * cus;
}

template <int L, bool NT>
void fold16_bwd_launch_t(const float* mats, const float* dOut, float* dMats, int* flags, long long Wn, hipStream_t st) {
  long long blocks = (Wn + LME_BWD_WAVES - 1) / LME_BWD_WAVES;
  const long long cap = resident_blocks((const void*)lme_fold16_bwd_mfma_k<L, NT>, 64 * LME_BWD_WAVES);
  if (blocks > cap) blocks = cap;   // persistent waves: exactly one resident round
  hipLaunchKernelGGL((lme_fold16_bwd_mfma_k<L, NT>), dim3((unsigned)blocks), dim3(64 * LME_BWD_WAVES), 0, st, mats, dOut, dMats, flags, Wn);
}
template <bool NT>
void fold16_bwd_launch(const float* mats, const float* dOut, float* dMats, int* flags, long long Wn, int L, hipStream_t st) {
  switch (L) {
#define LME_BWD_CASE(LL) case LL: return fold16_bwd_launch_t<LL, NT>(mats, dOut, dMats, flags, Wn, st);
    LME_BWD_CASE(2) LME_BWD_CASE(3) LME_BWD_CASE(4) LME_BWD_CASE(5) LME_BWD_CASE(6) LME_BWD_CASE(7) LME_BWD_CASE(8) LME_BWD_CASE(9)
    LME_BWD_CASE(10) LME_BWD_CASE(11) LME_BWD_CASE(12) LME_BWD_CASE(13) LME_BWD_CASE(14) LME_BWD_CASE(15) LME_BWD_CASE(16)
#undef LME_BWD_CASE
    default: break;   // (the caller sends 2 <= L <= 16)
  }
}

// the fold's matrices do not fit the last-level cache (256 MiB): stream them (lme_ld4<true>)
static bool lme_streams(long long Wn, int L) { return Wn * (long long)L * 1024 > (512ll << 20); }

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
               long long batch, int T, int R, int I, long long sA, long long sB, hipStream_t st,
               const int* only_batch = nullptr) {
  if (dA) {
    const long long total = (sA == 0 ? 1 : batch) * T * R;
    hipLaunchKernelGGL((lme_bwd_dA_k<S, A>), dim3(grid_for(total, 256)), dim3(256), 0, st,
                       (const S*)lA, (const S*)lB, (const S*)out, (const S*)dO, (S*)dA, batch, T, R,
                       I, sA, sB, only_batch);
    DCTN_CHECK_LAUNCH();
  }
  if (dB) {
    const long long total = (sB == 0 ? 1 : batch) * R * I;
    hipLaunchKernelGGL((lme_bwd_dB_k<S, A>), dim3(grid_for(total, 256)), dim3(256), 0, st,
                       (const S*)lA, (const S*)lB, (const S*)out, (const S*)dO, (S*)dB, batch, T, R,
                       I, sA, sB, only_batch);
    DCTN_CHECK_LAUNCH();
  }
  if (!only_batch) dctn_set_last_kernel("logmatmulexp_bwd");
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
                    hipStream_t st, const int* only_flagged = nullptr) {
  const size_t lds = ((size_t)(L + 2) * D * (D + 1)) * sizeof(A);
  if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
  (void)hipFuncSetAttribute((const void*)lme_fold_bwd_k<S, A>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  unsigned grid = (unsigned)(Wn < 256 * 32 ? Wn : 256 * 32);
  if (only_flagged) {   // one turn of D * D flags per workgroup
    const long long g = (Wn + (long long)D * D - 1) / ((long long)D * D);
    grid = (unsigned)(g < 256 * 32 ? g : 256 * 32);
  }
  hipLaunchKernelGGL((lme_fold_bwd_k<S, A>), dim3(grid), dim3(D * D), lds, st, (const S*)mats,
                     (const S*)dOut, (S*)dMats, Wn, L, D, only_flagged);
  DCTN_CHECK_LAUNCH();
  if (!only_flagged) dctn_set_last_kernel("logmatmulexp_fold_bwd");
  return DCTN_OK;
}

}  // namespace

// logmatmulexp_gemm.hip
bool lme_gemm_wanted(long long batch, int T, int R, int I, long long sA, long long sB, int dtype);
size_t lme_gemm_workspace(long long batch, int T, int R, int I);
int lme_gemm_fwd(const void* A, const void* B, void* out, void* ws, long long batch, int T, int R, int I,
                 long long sA, long long sB, hipStream_t st);
int lme_gemm_bwd(const void* A, const void* B, const void* out, const void* dO, void* dA, void* dB, void* ws,
                 long long batch, int T, int R, int I, long long sA, long long sB, hipStream_t st,
                 const int** unsafe_out);

extern "C" {

size_t dctn_logmatmulexp_workspace_bytes(int64_t batch, int Theta, int R, int I, int64_t strideA_batch,
                                         int64_t strideB_batch, int dtype) {
  if (batch < 1 || Theta < 1 || R < 1 || I < 1) return 0;
  if (!lme_gemm_wanted(batch, Theta, R, I, strideA_batch, strideB_batch, dtype)) return 0;
  return lme_gemm_workspace(batch, Theta, R, I);
}

int dctn_logmatmulexp_fwd(const void* logA, const void* logB, void* out, void* workspace,
                          size_t workspace_bytes, int64_t batch, int Theta, int R, int I,
                          int64_t strideA_batch, int64_t strideB_batch, int dtype, void* stream) {
  if (!logA || !logB || !out) return DCTN_ERR_NULL;
  if (batch < 1 || Theta < 1 || R < 1 || I < 1) return DCTN_ERR_BAD_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (workspace && lme_gemm_wanted(batch, Theta, R, I, strideA_batch, strideB_batch, dtype) &&
      workspace_bytes >= lme_gemm_workspace(batch, Theta, R, I))
    return lme_gemm_fwd(logA, logB, out, workspace, batch, Theta, R, I, strideA_batch, strideB_batch, st);
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
                          void* dA, void* dB, void* workspace, size_t workspace_bytes, int64_t batch,
                          int Theta, int R, int I, int64_t strideA_batch, int64_t strideB_batch, int dtype,
                          void* stream) {
  if (!logA || !logB || !out || !dOut) return DCTN_ERR_NULL;
  if (batch < 1 || Theta < 1 || R < 1 || I < 1) return DCTN_ERR_BAD_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (workspace && lme_gemm_wanted(batch, Theta, R, I, strideA_batch, strideB_batch, dtype) &&
      workspace_bytes >= lme_gemm_workspace(batch, Theta, R, I)) {
    const int* unsafe = nullptr;
    const int rc = lme_gemm_bwd(logA, logB, out, dOut, dA, dB, workspace, batch, Theta, R, I, strideA_batch,
                                strideB_batch, st, &unsafe);
    if (rc != DCTN_OK) return rc;
    // batch elements whose factored form would overflow: exact kernels, those elements only
    return bwd_launch<float, float>(logA, logB, out, dOut, dA, dB, batch, Theta, R, I, strideA_batch,
                                    strideB_batch, st, unsafe);
  }
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
  (void)L;
  // prefix folds live in LDS; the factored backward keeps one "needs the exact kernel" flag per window
  if (backward && dtype == DCTN_F32 && D == 16 && Wn > 0) return 256 + (size_t)Wn * sizeof(int);
  return 256;
}

int dctn_logmatmulexp_fold_fwd(const void* mats, void* out, int64_t Wn, int L, int D, int dtype,
                               void* stream) {
  if (!mats || !out) return DCTN_ERR_NULL;
  if (Wn < 1 || L < 1 || D < 1) return DCTN_ERR_BAD_SHAPE;
  if (D > 32) return DCTN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCTN_F32 && D == 16 && ((uintptr_t)mats % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
    const bool nt = lme_streams(Wn, L);
    long long blocks = (Wn + 3) / 4;
    const dim3 b(256);
    if (L == 1) {   // nothing to fold
      if (hipMemcpyAsync(out, mats, (size_t)Wn * 1024, hipMemcpyDeviceToDevice, st) != hipSuccess) return DCTN_ERR_LAUNCH;
      dctn_set_last_kernel("logmatmulexp_fold_fwd_mfma16");
      return DCTN_OK;
    }
    auto exact = [&](auto kern) {
      const long long cap = resident_blocks((const void*)kern, 256);
      if (blocks > cap) blocks = cap;   // persistent waves: exactly one resident round ...
      const long long need = (Wn + 4ll * LME_FWD_ITMAX - 1) / (4ll * LME_FWD_ITMAX);   // ... unless a wave would run out of note bits
      if (blocks < need) blocks = need;
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), b, 0, st, (const float*)mats, (float*)out, (long long)Wn);
    };
    switch (L) {
#define LME_FWD_CASE(LL) case LL: nt ? exact(lme_fold16_fwd_k<LL, true>) : exact(lme_fold16_fwd_k<LL, false>); break;
      LME_FWD_CASE(2) LME_FWD_CASE(3) LME_FWD_CASE(4) LME_FWD_CASE(5) LME_FWD_CASE(6) LME_FWD_CASE(7) LME_FWD_CASE(8) LME_FWD_CASE(9)
      LME_FWD_CASE(10) LME_FWD_CASE(11) LME_FWD_CASE(12) LME_FWD_CASE(13) LME_FWD_CASE(14) LME_FWD_CASE(15) LME_FWD_CASE(16)
#undef LME_FWD_CASE
      default: {
        const long long cap = resident_blocks((const void*)lme_fold16_fwd_mfma_k, 256);
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(lme_fold16_fwd_mfma_k, dim3((unsigned)blocks), b, 0, st, (const float*)mats, (float*)out, (long long)Wn, L);
      }
    }
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
  if (!mats || !dOut || !dMats) return DCTN_ERR_NULL;
  if (Wn < 1 || L < 1 || D < 1) return DCTN_ERR_BAD_SHAPE;
  if (D > 32) return DCTN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCTN_F32 && D == 16 && L >= 2 && L <= 16 && workspace &&
      workspace_bytes >= 256 + (size_t)Wn * sizeof(int) && ((uintptr_t)mats % 16 == 0) &&
      ((uintptr_t)dOut % 16 == 0) && ((uintptr_t)dMats % 16 == 0) && ((uintptr_t)workspace % 4 == 0) &&
      (size_t)(L + 2) * 16 * 17 * sizeof(float) <= DCTN_LDS_BUDGET) {
    int* flags = reinterpret_cast<int*>(static_cast<unsigned char*>(workspace) + 256);
    const float* m = (const float*)mats;
    const float* dy = (const float*)dOut;
    float* dm = (float*)dMats;
    const bool nt = lme_streams(Wn, L);
    nt ? fold16_bwd_launch<true>(m, dy, dm, flags, Wn, L, st) : fold16_bwd_launch<false>(m, dy, dm, flags, Wn, L, st);
    DCTN_CHECK_LAUNCH();
    {   // second tier (column shifts) over the flagged windows; what it cannot represent either stays flagged
      long long blocks = (Wn + 255) / 256;   // a wave per 64 flags
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(lme_fold16_bwd_cols_k, dim3((unsigned)blocks), dim3(256), 0, st, m, dy, dm, flags, (long long)Wn, L);
      DCTN_CHECK_LAUNCH();
    }
    const int rc = fold_bwd_launch<float, float>(mats, dOut, dMats, Wn, L, D, st, flags);  // flagged windows only
    if (rc != DCTN_OK) return rc;
    dctn_set_last_kernel("logmatmulexp_fold_bwd_mfma16");
    return DCTN_OK;
  }
  switch (dtype) {
    case DCTN_F32: return fold_bwd_launch<float, float>(mats, dOut, dMats, Wn, L, D, st);
    case DCTN_F64: return fold_bwd_launch<double, double>(mats, dOut, dMats, Wn, L, D, st);
    case DCTN_BF16: return fold_bwd_launch<bf16_t, float>(mats, dOut, dMats, Wn, L, D, st);
  }
  return DCTN_ERR_BAD_DTYPE;
}

}  // extern "C"
