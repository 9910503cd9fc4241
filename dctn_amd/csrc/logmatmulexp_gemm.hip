// logmatmulexp for matrices that are not tiny: the factored form on the matrix cores.
//
//   out[t,i] = log sum_r exp(A[t,r] + B[r,i])
//            = a_t + b_i + log( sum_r exp(A[t,r] - a_t) * exp(B[r,i] - b_i) ),   a_t = max_r A[t,r], b_i = max_r B[r,i]
// i.e. one exp per INPUT element and one log per output element around a plain GEMM (exact float32 on
// v_mfma_f32_32x32x2_f32), instead of R exps per output element: for the reference's own benchmark
// (small_experiments/logmatmulexp_benchmark: reduce(logmatmulexp, 6 x (256 x 256))) 0.6 M exps instead
// of 84 M.  Everything runs in the base-2 domain (v_exp_f32 / v_log_f32 are base 2).
// The factorisation is accepted where its result shows it is safe: S[t,i] >= 2^-100 (terms the shifts
// could have flushed are then < 2^-26 of the sum); NaN, +inf and all -inf rows / columns make S NaN.
// Rejected 64 x 64 output tiles are recomputed by the exact max-shifted kernel (logmatmulexp.hip), so
// torch.logsumexp semantics are kept.
//
// Backward, with E_A = 2^(A' - a), E_B = 2^(B' - b), H = G * 2^(a + b - out')  (= G / S):
//   dA = E_A .* (H x E_B^T)        dB = E_B .* (E_A^T x H)
// the same tiled kernel with other loaders / epilogues.  A batch element whose H would overflow
// (a + b - out' > 100, or non-finite inputs) is left to the exact backward kernels.
#include "common.h"

#include <math.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr float L2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float SMIN = 7.888609052210118e-31f;   // 2^-100
constexpr int TM = 64, TN = 64, TK = 32;           // workgroup tile, k chunk
constexpr int LA_P = TK + 1, LB_P = TN + 1;        // padded LDS rows

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }

// amax[b][t] = log2(e) * max_r A[b][t][r] (one wave per row), bmax[b][i] = log2(e) * max_r B[b][r][i]
// (64 columns x 4 row groups per workgroup); also clears the `nflags` flags the later kernels set.
__global__ __launch_bounds__(256) void lme_max_k(const float* __restrict__ A, const float* __restrict__ B,
                                                 float* __restrict__ amax, float* __restrict__ bmax,
                                                 int* __restrict__ flags, long long nflags, long long batch, int T,
                                                 int R, int I, long long sA, long long sB, int a_blocks) {
  for (long long f = (long long)blockIdx.x * 256 + threadIdx.x; f < nflags; f += (long long)gridDim.x * 256) flags[f] = 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if ((int)blockIdx.x < a_blocks) {
    const long long row = (long long)blockIdx.x * 4 + wv;   // (b, t)
    if (row >= batch * T) return;
    const long long b = row / T;
    const int t = (int)(row - b * T);
    const float* a = A + b * sA + (long long)t * R;
    float m = -INFINITY;
    bool bad = false;   // fmaxf drops NaN: carry it explicitly so that the product sees it
#pragma unroll 4
    for (int r = lane; r < R; r += 64) {
      const float v = a[r];
      bad = bad | (v != v);
      m = fmaxf(m, v);
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const bool anyb = __any(bad);
    if (lane == 0) amax[row] = anyb ? NAN : m * L2E;
    return;
  }
  __shared__ float red[4][64];
  __shared__ int redbad[4][64];
  const int cpb = (I + 63) / 64;                       // column blocks per batch element
  const int cb = (int)blockIdx.x - a_blocks;
  const long long b = cb / cpb;
  const int i = (cb - (int)b * cpb) * 64 + lane;
  const int ic = i < I ? i : I - 1;
  const float* bp = B + b * sB + ic;
  float m = -INFINITY;
  bool bad = false;
#pragma unroll 8
  for (int r = wv; r < R; r += 4) {
    const float v = bp[(long long)r * I];
    bad = bad | (v != v);
    m = fmaxf(m, v);
  }
  red[wv][lane] = m;
  redbad[wv][lane] = bad;
  __syncthreads();
  if (wv == 0 && i < I) {
    m = fmaxf(fmaxf(red[0][lane], red[1][lane]), fmaxf(red[2][lane], red[3][lane]));
    const int anyb = redbad[0][lane] | redbad[1][lane] | redbad[2][lane] | redbad[3][lane];
    bmax[b * I + i] = anyb ? NAN : m * L2E;
  }
}

// per batch element: is the factored backward representable?  (H exponent <= 100, everything finite)
__global__ __launch_bounds__(256) void lme_bwd_check_k(const float* __restrict__ out, const float* __restrict__ amax,
                                                       const float* __restrict__ bmax, int* __restrict__ unsafe,
                                                       long long batch, int T, int I) {
  const long long total = batch * T * I;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(idx % I);
    const long long t2 = idx / I;
    const int t = (int)(t2 % T);
    const long long b = t2 / T;
    const float e = amax[b * T + t] + bmax[b * I + i] - out[idx] * L2E;
    if (!(e <= 100.f) || !(e >= -1e30f)) atomicOr(&unsafe[b], 1);
  }
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

enum { G_FWD = 0, G_DA = 1, G_DB = 2 };

struct GemmP {
  long long batch;
  int T, R, I;
  long long sA, sB;
  int M, N, K;          // C is M x N, reduction over K
  int tiles_n, tiles_m, tile;   // output tiling of this launch (tile = 32 or 64)
};

// C[m,n] = epilogue( sum_k LA(m,k) * LB(k,n) ), one 32 x 32 accumulator tile per wave, two workgroup shapes:
//   WV = 2, KS = 1: 2 x 2 waves on a 64 x 64 tile, operands staged by the whole workgroup;
//   WV = 1, KS = 4: products too small to fill the chip with 64 x 64 tiles — a 32 x 32 tile whose k range is
//                   split over 4 waves (chunks c = wave mod 4, wave-private LDS, no workgroup barriers in the
//                   loop); the four partial tiles are summed through LDS before the epilogue.
// The raw operands of a wave's next k-chunk are fetched into registers before the MFMAs of the current one; the
// exponentials are applied on the way into LDS.  Loads are unconditional on clamped indices (no branches).
template <int MODE, int WV, int KS>
__global__ __launch_bounds__(64 * WV * WV * KS) void lme_gemm_k(const float* __restrict__ A, const float* __restrict__ B,
                                                           const float* __restrict__ out_in,
                                                           const float* __restrict__ G, const float* __restrict__ amax,
                                                           const float* __restrict__ bmax, float* __restrict__ C,
                                                           int* __restrict__ flags, GemmP p) {
  static_assert((WV == 2 && KS == 1) || (WV == 1 && KS == 4), "workgroup shapes");
  constexpr int NT = 64 * WV * WV, BM = 32 * WV, BN = 32 * WV;   // threads staging one operand tile, tile
  constexpr int LAP = TK + 1, LBP = BN + 1;
  constexpr int NU = BM * TK / NT;                               // elements per thread per operand per chunk (8 / 16)
  constexpr int LOG_BM = WV == 2 ? 6 : 5;
  __shared__ float la_all[KS * BM * LAP];
  __shared__ float lb_all[KS * TK * LBP];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
  const int tid = KS > 1 ? lane : (int)threadIdx.x;              // index within the staging group
  const int wm = KS > 1 ? 0 : wv / WV, wn = KS > 1 ? 0 : wv % WV;
  float* la = la_all + (KS > 1 ? wv * BM * LAP : 0);
  float* lb = lb_all + (KS > 1 ? wv * TK * LBP : 0);
  const long long b = blockIdx.z;
  if (MODE != G_FWD && flags[b]) return;   // unsafe batch element: the exact backward kernels own it
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const float* Ab = A + b * p.sA;
  const float* Bb = B + b * p.sB;
  const float* ob = out_in ? out_in + b * (long long)p.T * p.I : nullptr;
  const float* gb = G ? G + b * (long long)p.T * p.I : nullptr;
  const float* am = amax + b * p.T;
  const float* bm = bmax + b * p.I;
  // element e = tid + NT u of a staged operand, "k fastest": k = e & 31, x = e >> 5; "x fastest": x = e & (BM-1), k = e >> LOG_BM
  // FWD: LA k fastest (A[t][r]),  LB x fastest (B[r][i]).   DA: LA k fastest (H[t][i]), LB k fastest (B[r][i]).
  // DB:  LA x fastest (A[t][r]),  LB x fastest (H[t][i]).
  constexpr bool LA_KF = MODE != G_DB, LB_KF = MODE == G_DA;
  const int kf_k = tid & (TK - 1), kf_x = tid >> 5;              // + (NT / 32) u
  const int xf_x = tid & (BM - 1), xf_k = tid >> LOG_BM;         // + (NT / BM) u
  // tile-constant shifts
  float rowc[NU];
  float colc = 0.f;
  if (MODE == G_FWD || MODE == G_DA) {
#pragma unroll
    for (int u = 0; u < NU; ++u) rowc[u] = am[min(m0 + kf_x + (NT / 32) * u, p.T - 1)];
  }
  if (MODE == G_FWD || MODE == G_DB) colc = bm[min(n0 + xf_x, p.I - 1)];
  float a1[NU], a2[NU], b1[NU], b2[NU], kv[NU];
  auto fetch = [&](int k0) {
    if (MODE == G_FWD) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        a1[u] = Ab[(long long)min(m0 + kf_x + (NT / 32) * u, p.T - 1) * p.R + min(k0 + kf_k, p.R - 1)];
        b1[u] = Bb[(long long)min(k0 + xf_k + (NT / BM) * u, p.R - 1) * p.I + min(n0 + xf_x, p.I - 1)];
      }
    } else if (MODE == G_DA) {
      kv[0] = bm[min(k0 + kf_k, p.I - 1)];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const long long o = (long long)min(m0 + kf_x + (NT / 32) * u, p.T - 1) * p.I + min(k0 + kf_k, p.I - 1);
        a1[u] = gb[o];
        a2[u] = ob[o];
        b1[u] = Bb[(long long)min(n0 + kf_x + (NT / 32) * u, p.R - 1) * p.I + min(k0 + kf_k, p.I - 1)];
      }
    } else {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int t = min(k0 + xf_k + (NT / BM) * u, p.T - 1);
        kv[u] = am[t];
        a1[u] = Ab[(long long)t * p.R + min(m0 + xf_x, p.R - 1)];
        const long long o = (long long)t * p.I + min(n0 + xf_x, p.I - 1);
        b1[u] = gb[o];
        b2[u] = ob[o];
      }
    }
  };
  auto stage = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (MODE == G_FWD) {
        const int m = kf_x + (NT / 32) * u, k = xf_k + (NT / BM) * u;
        const bool va = (m0 + m < p.T) & (k0 + kf_k < p.R), vb = (k0 + k < p.R) & (n0 + xf_x < p.I);
        la[m * LAP + kf_k] = va ? ex2(a1[u] * L2E - rowc[u]) : 0.f;
        lb[k * LBP + xf_x] = vb ? ex2(b1[u] * L2E - colc) : 0.f;
      } else if (MODE == G_DA) {
        const int m = kf_x + (NT / 32) * u;   // also the n (= r) index of LB
        const bool va = (m0 + m < p.T) & (k0 + kf_k < p.I), vb = (n0 + m < p.R) & (k0 + kf_k < p.I);
        la[m * LAP + kf_k] = va ? a1[u] * ex2(rowc[u] + kv[0] - a2[u] * L2E) : 0.f;
        lb[kf_k * LBP + m] = vb ? ex2(b1[u] * L2E - kv[0]) : 0.f;
      } else {
        const int k = xf_k + (NT / BM) * u;
        const bool va = (m0 + xf_x < p.R) & (k0 + k < p.T), vb = (k0 + k < p.T) & (n0 + xf_x < p.I);
        la[xf_x * LAP + k] = va ? ex2(a1[u] * L2E - kv[u]) : 0.f;
        lb[k * LBP + xf_x] = vb ? b1[u] * ex2(kv[u] + colc - b2[u] * L2E) : 0.f;
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  const int kbeg = KS > 1 ? wv * TK : 0;
  if (kbeg < p.K) fetch(kbeg);
  for (int k0 = kbeg; k0 < p.K; k0 += KS * TK) {
    if (KS > 1) wave_lds_sync(); else __syncthreads();
    stage(k0);
    if (KS > 1) wave_lds_sync(); else __syncthreads();
    if (k0 + KS * TK < p.K) fetch(k0 + KS * TK);
    const float* pa = la + (32 * wm + c) * LAP + h;
    const float* pb = lb + h * LBP + 32 * wn + c;
#pragma unroll
    for (int s = 0; s < TK / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * s], pb[2 * s * LBP], acc, 0, 0, 0);
  }
  // ---- epilogue: accumulator register v is row (v&3) + 8 (v>>2) + 4 h, column c of the wave's sub-tile
  constexpr int NV = 16 / KS;   // k-split: wave w finishes registers [4 w, 4 w + 4) of the summed tile
  if (KS > 1) {
    __shared__ float red[KS * 16 * 64];
#pragma unroll
    for (int v = 0; v < 16; ++v) red[(wv * 16 + v) * 64 + lane] = acc[v];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < KS; ++w) sum += red[(w * 16 + wv * NV + j) * 64 + lane];
      acc[j] = sum;
    }
  }
  const int n = n0 + 32 * wn + c;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = KS > 1 ? wv * NV + j : j;
    const int m = m0 + 32 * wm + (v & 3) + 8 * (v >> 2) + 4 * h;
    if (m < p.M && n < p.N) {
      if (MODE == G_FWD) {
        const float S = acc[j];
        bad = bad || !(S >= SMIN);
        C[b * (long long)p.T * p.I + (long long)m * p.I + n] = (lg2(S) + am[m] + bm[n]) * LN2;
      } else if (MODE == G_DA) {
        const long long o = (long long)m * p.R + n;
        C[b * p.sA + o] = acc[j] * ex2(Ab[o] * L2E - am[m]);
      } else {
        const long long o = (long long)m * p.I + n;
        C[b * p.sB + o] = acc[j] * ex2(Bb[o] * L2E - bm[n]);
      }
    }
  }
  const bool anybad = __any(bad);
  if (MODE == G_FWD && anybad && lane == 0)
    atomicOr(&flags[(b * p.tiles_m + blockIdx.y) * p.tiles_n + blockIdx.x], 1);
}

// exact recomputation of the flagged output tiles (one workgroup per tile)
__global__ __launch_bounds__(256) void lme_fwd_fix_k(const float* __restrict__ A, const float* __restrict__ B,
                                                     float* __restrict__ out, const int* __restrict__ flags, GemmP p) {
  const long long b = blockIdx.z;
  if (!flags[(b * p.tiles_m + blockIdx.y) * p.tiles_n + blockIdx.x]) return;
  const float* a0 = A + b * p.sA;
  const float* b0 = B + b * p.sB;
  for (int e = threadIdx.x; e < p.tile * p.tile; e += 256) {
    const int t = blockIdx.y * p.tile + e / p.tile, i = blockIdx.x * p.tile + e % p.tile;
    if (t >= p.T || i >= p.I) continue;
    const float* a = a0 + (long long)t * p.R;
    const float* bp = b0 + i;
    float m = -INFINITY;
    for (int r = 0; r < p.R; ++r) {
      const float s = a[r] + bp[(long long)r * p.I];
      m = (m != m || s != s) ? (m + s) : fmaxf(m, s);
    }
    const float mm = isinf(m) ? 0.f : m;
    float s = 0.f;
    for (int r = 0; r < p.R; ++r) s += expf(a[r] + bp[(long long)r * p.I] - mm);
    out[b * (long long)p.T * p.I + (long long)t * p.I + i] = logf(s) + mm;
  }
}

GemmP gemm_plan(long long batch, int T, int R, int I, long long sA, long long sB, int M, int N, int K) {
  GemmP p{batch, T, R, I, sA, sB, M, N, K, 0, 0, 64};
  // 64 x 64 tiles (2 x 2 waves) when they fill the chip, else 32 x 32 with the k range split over 4 waves
  if ((long long)((M + 63) / 64) * ((N + 63) / 64) * batch < 256) p.tile = 32;
  p.tiles_m = (M + p.tile - 1) / p.tile;
  p.tiles_n = (N + p.tile - 1) / p.tile;
  return p;
}

template <int MODE>
void gemm_launch(const GemmP& p, hipStream_t st, const float* A, const float* B, const float* out, const float* G,
                 const float* amax, const float* bmax, float* C, int* flags) {
  const dim3 grid(p.tiles_n, p.tiles_m, (unsigned)p.batch);
  if (p.tile == 64)
    hipLaunchKernelGGL((lme_gemm_k<MODE, 2, 1>), grid, dim3(256), 0, st, A, B, out, G, amax, bmax, C, flags, p);
  else
    hipLaunchKernelGGL((lme_gemm_k<MODE, 1, 4>), grid, dim3(256), 0, st, A, B, out, G, amax, bmax, C, flags, p);
}

void max_launch(const void* A, const void* B, float* amax, float* bmax, int* flags, long long nflags, long long batch,
                int T, int R, int I, long long sA, long long sB, hipStream_t st) {
  const int a_blocks = (int)((batch * T + 3) / 4), b_blocks = (int)(batch * ((I + 63) / 64));
  hipLaunchKernelGGL(lme_max_k, dim3(a_blocks + b_blocks), dim3(256), 0, st, (const float*)A, (const float*)B, amax,
                     bmax, flags, nflags, batch, T, R, I, sA, sB, a_blocks);
}

}  // namespace

// The factored path is taken for float32 products that are not tiny and whose operands are not
// broadcast over the batch (the exact kernels keep those).
bool lme_gemm_wanted(long long batch, int T, int R, int I, long long sA, long long sB, int dtype) {
  if (dtype != DCTN_F32) return false;
  if ((sA == 0 || sB == 0) && batch > 1) return false;
  if (R < 16 || (long long)T * I < 1024) return false;
  return batch <= 65535 && batch * (long long)(T > I ? T : I) < (1ll << 30);
}

size_t lme_gemm_workspace(long long batch, int T, int R, int I) {
  (void)R;
  const long long tiles = (long long)((T + 31) / 32) * ((I + 31) / 32);
  return (size_t)(batch * T + batch * I) * sizeof(float) + (size_t)(batch * tiles + batch) * sizeof(int) + 256;
}

static void lme_ws_split(void* ws, long long batch, int T, int I, float*& amax, float*& bmax, int*& flags) {
  amax = static_cast<float*>(ws);
  bmax = amax + batch * T;
  flags = reinterpret_cast<int*>(bmax + batch * I);
}

int lme_gemm_fwd(const void* A, const void* B, void* out, void* ws, long long batch, int T, int R, int I,
                 long long sA, long long sB, hipStream_t st) {
  float *amax, *bmax;
  int* flags;
  lme_ws_split(ws, batch, T, I, amax, bmax, flags);
  const GemmP p = gemm_plan(batch, T, R, I, sA, sB, T, I, R);
  max_launch(A, B, amax, bmax, flags, batch * p.tiles_m * p.tiles_n, batch, T, R, I, sA, sB, st);
  DCTN_CHECK_LAUNCH();
  gemm_launch<G_FWD>(p, st, (const float*)A, (const float*)B, nullptr, nullptr, amax, bmax, (float*)out, flags);
  DCTN_CHECK_LAUNCH();
  hipLaunchKernelGGL(lme_fwd_fix_k, dim3(p.tiles_n, p.tiles_m, (unsigned)batch), dim3(256), 0, st, (const float*)A,
                     (const float*)B, (float*)out, flags, p);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("logmatmulexp_fwd_mfma_gemm");
  return DCTN_OK;
}

// Returns in `unsafe_out` a device pointer to per-batch flags (1 = left to the exact kernels).
int lme_gemm_bwd(const void* A, const void* B, const void* out, const void* dO, void* dA, void* dB, void* ws,
                 long long batch, int T, int R, int I, long long sA, long long sB, hipStream_t st,
                 const int** unsafe_out) {
  float *amax, *bmax;
  int* flags;
  lme_ws_split(ws, batch, T, I, amax, bmax, flags);
  max_launch(A, B, amax, bmax, flags, batch, batch, T, R, I, sA, sB, st);
  DCTN_CHECK_LAUNCH();
  {
    long long blocks = (batch * T * I + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(lme_bwd_check_k, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)out, amax, bmax, flags,
                       batch, T, I);
    DCTN_CHECK_LAUNCH();
  }
  if (dA) {
    const GemmP p = gemm_plan(batch, T, R, I, sA, sB, T, R, I);
    gemm_launch<G_DA>(p, st, (const float*)A, (const float*)B, (const float*)out, (const float*)dO, amax, bmax,
                      (float*)dA, flags);
    DCTN_CHECK_LAUNCH();
  }
  if (dB) {
    const GemmP p = gemm_plan(batch, T, R, I, sA, sB, R, I, T);
    gemm_launch<G_DB>(p, st, (const float*)A, (const float*)B, (const float*)out, (const float*)dO, amax, bmax,
                      (float*)dB, flags);
    DCTN_CHECK_LAUNCH();
  }
  *unsafe_out = flags;
  dctn_set_last_kernel("logmatmulexp_bwd_mfma_gemm");
  return DCTN_OK;
}
