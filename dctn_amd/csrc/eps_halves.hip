// EPS on float64 for cores that are not tiny: the two-halves path on the f64 matrix cores.
//
// Replaces dctn/eps.py:19-40 for float64 (the dtype of the reference's own tests and of its EPS
// micro-benchmark, BASELINE cfg1: B = 64, K = 4, Q = 2, O = 2 -> core 65 536 x 2).  The generic kernels
// (one lane per window, scalar FMAs) run that shape at 0.6 TFLOP/s; here the work is what the reference's
// contraction path asks for (eps.py:25-40), as GEMMs on v_mfma_f64_16x16x4_f64:
//
//   P0[w, i0] = prod of the first n0 factors, P1[w, i1] = prod of the last n1 factors   (materialised per
//   chunk of windows: float64 workloads are small and HBM is 288 GB; the chunk is bounded to ~1 GiB)
//   forward : Z = P0 x Core[(i0), (i1 o)]            out[w,o]  = sum_i1 Z[w,i1,o] P1[w,i1]
//   dCore   : dCore[(i0), (i1 o)] = P0^T x T          T[w,(i1 o)] = P1[w,i1] dY[w,o]   (formed in the loader)
//   dX      : dP0 = T x Core^T,  Z again, dP1[w,i1] = sum_o dY[w,o] Z[w,i1,o], then the leave-one-out
//             products per factor -> gxw[(n q)][w] and the deterministic gather of eps_generic.hip.
//
// One GEMM kernel (64 x 64 tile, 2 x 2 waves x 2 x 2 MFMA tiles, k chunks of 16 staged through LDS with the
// next chunk's operands prefetched into registers), loaders for the four operand layouts, split-k over
// grid.z for the dCore product (K = windows).  Accumulator layout of the instruction on gfx950 (measured,
// tools/mfma64probe.hip): register v of lane l is D[4 v + l / 16][l % 16].
#include "common.h"

#include <stdlib.h>

#include <type_traits>

typedef __attribute__((ext_vector_type(4))) double f64x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// The two 16 x 16 x 4 matrix instructions share the operand layout (A: row = lane % 16, k = lane / 16; B: column =
// lane % 16, k = lane / 16) and differ in the accumulator rows: f64 register v of lane l is row 4 v + l / 16
// (measured, tools/mfma64probe.hip), f32 register v is row 4 (l / 16) + v.
template <typename T> struct Mma;
template <> struct Mma<double> {
  typedef f64x4 acc_t;
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int v, int lk) { return 4 * v + lk; }
};
template <> struct Mma<float> {
  typedef f32x4 acc_t;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int v, int lk) { return 4 * lk + v; }
};

namespace {

constexpr int GT = 64, GK = 16;                 // tile, k chunk
constexpr size_t CHUNK_BYTES = (size_t)1 << 30;        // bound of the per-chunk buffers
constexpr size_t SMALL_CHUNK_BYTES = (size_t)256 << 10;  // ... under DCTN_OPT_SMALL_CHUNKS (tests: many chunks on small inputs)
static inline size_t chunk_budget(const EpsP& p) { return (p.opts & DCTN_OPT_SMALL_CHUNKS) ? SMALL_CHUNK_BYTES : CHUNK_BYTES; }

struct HalfP {
  EpsP p;
  int n0, n1;
  long long A, Bn, NB;    // Q^n0, Q^n1, Bn * O
  long long wc;           // windows per chunk (multiple of 64)
  int ksplit;             // grid.z of the dCore product
};

// big_first: the larger half is the first one (the bf16 products: K = A of the forward GEMM is then the long
// dimension and its fused epilogue touches O Bn, not O A, columns per window)
HalfP make_half(const EpsP& p, size_t esz, bool big_first = false) {
  HalfP h;
  h.p = p;
  h.n0 = big_first ? p.N - p.N / 2 : p.N / 2;
  h.n1 = p.N - h.n0;
  h.A = ipow_ll(p.Q, h.n0);
  h.Bn = ipow_ll(p.Q, h.n1);
  h.NB = h.Bn * p.O;
  const long long per_win = (2 * h.A + 2 * h.Bn + h.NB) * (long long)esz;
  const size_t budget = chunk_budget(p);
  long long wc = (long long)(budget / (size_t)per_win);
  wc = wc / 64 * 64;
  if (wc < 64) wc = 64;
  const long long wn64 = (p.Wn + 63) / 64 * 64;
  h.wc = wc < wn64 ? wc : wn64;
  // split the window sum of the dCore product until ~2 workgroups per CU exist
  const long long tiles = ((h.A + GT - 1) / GT) * ((h.NB + GT - 1) / GT);
  long long ks = (512 + tiles - 1) / tiles;
  const long long max_ks = (h.wc + 255) / 256;   // >= 256 windows per slice
  if (ks > max_ks) ks = max_ks;
  if (ks < 1) ks = 1;
  if (ks > 64) ks = 64;
  h.ksplit = (int)ks;
  return h;
}

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------ P0 / P1 of a chunk of windows
// One wave per window (4 per workgroup).  Each half is itself a Kronecker product of two quarter tables
// (its leading ceil(nd/2) factors x its trailing ones, <= 32 entries each for halves <= 1024): the tables cost
// nd/2 multiplies per entry, every entry of P0 / P1 then one multiply and two LDS reads.
template <int LOGQ, typename T, typename S = T>   // T: arithmetic / LDS type, S: storage type of x, P0, P1
__global__ __launch_bounds__(256) void halves_k(const S* __restrict__ x, S* __restrict__ P0,
                                                    S* __restrict__ P1, HalfP h, long long w0, long long nw) {
  extern __shared__ __align__(16) unsigned char sm_raw[];
  T* sm = reinterpret_cast<T*>(sm_raw);
  const EpsP& p = h.p;
  const int Q = p.Q;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // quarter tables: half s has nh_s leading and nl_s trailing factors, sizes H_s = Q^nh_s, L_s = Q^nl_s
  const int nl0 = h.n0 / 2, nh0 = h.n0 - nl0, nl1 = h.n1 / 2, nh1 = h.n1 - nl1;
  int H0 = 1, L0 = 1, H1 = 1, L1 = 1;
  for (int d = 0; d < nh0; ++d) H0 *= Q;
  for (int d = 0; d < nl0; ++d) L0 *= Q;
  for (int d = 0; d < nh1; ++d) H1 *= Q;
  for (int d = 0; d < nl1; ++d) L1 *= Q;
  const int off1 = H0, off2 = H0 + L0, off3 = off2 + H1, ntab = off3 + L1;
  T* xs = sm + (size_t)wv * (p.N * Q + ntab);   // [N][Q]
  T* tab = xs + p.N * Q;                        // hi0 | lo0 | hi1 | lo1
  const int hw = p.Ho * p.Wo;
  const int A = (int)h.A, Bn = (int)h.Bn;
  for (long long wl = (long long)blockIdx.x * 4 + wv; wl < nw; wl += (long long)gridDim.x * 4) {
    const long long w = w0 + wl;
    const long long b = w / hw;
    const int rem = (int)(w - b * hw), ho = rem / p.Wo, wo = rem - ho * p.Wo;
    wave_sync_lds();
    for (int e = lane; e < p.N * Q; e += 64) {
      const int n = e / Q, q = e - n * Q;
      const int pos = n / p.C, ch = n - pos * p.C, dh = pos / p.K, dw = pos - dh * p.K;
      xs[e] = (T)x[ch * p.s[0] + b * p.s[1] + (long long)(ho + dh) * p.s[2] + (long long)(wo + dw) * p.s[3] + q * p.s[4]];
    }
    wave_sync_lds();
    for (int e = lane; e < ntab; e += 64) {
      // table of this entry: hi0 | lo0 | hi1 | lo1 -> its first factor, number of factors, index
      const bool s1 = e >= off2, low = s1 ? e >= off3 : e >= off1;
      int t = e - (s1 ? (low ? off3 : off2) : (low ? off1 : 0));
      const int nd = s1 ? (low ? nl1 : nh1) : (low ? nl0 : nh0);
      const int base = (s1 ? h.n0 : 0) + (low ? (s1 ? nh1 : nh0) : 0);
      T pr = 1.0;
      for (int d = nd - 1; d >= 0; --d) {
        int digit;
        if (LOGQ > 0) {
          digit = t & ((1 << LOGQ) - 1);
          t >>= LOGQ;
        } else {
          digit = t % Q;
          t /= Q;
        }
        pr *= xs[(base + d) * Q + digit];
      }
      tab[e] = pr;
    }
    wave_sync_lds();
    for (int e = lane; e < A; e += 64) {
      int u, l;
      if (LOGQ > 0) {
        u = e >> (LOGQ * nl0);
        l = e & (L0 - 1);
      } else {
        u = e / L0;
        l = e - u * L0;
      }
      P0[wl * h.A + e] = (S)(tab[u] * tab[off1 + l]);
    }
    for (int e = lane; e < Bn; e += 64) {
      int u, l;
      if (LOGQ > 0) {
        u = e >> (LOGQ * nl1);
        l = e & (L1 - 1);
      } else {
        u = e / L1;
        l = e - u * L1;
      }
      P1[wl * h.Bn + e] = (S)(tab[off2 + u] * tab[off3 + l]);
    }
  }
}

// ------------------------------------------------------------------------------- the GEMM kernel
enum { A_KFAST = 0, A_MFAST = 1, A_T = 2 };   // A[m][k] k-contiguous / stored [k][m] / T[w, (i1 o)] formed from P1, dY
enum { B_NFAST = 0, B_KFAST = 1, B_T = 2 };   // B[k][n] n-contiguous / stored [n][k] / T (k = window)

template <typename T>
struct GemmD {
  int M, N, K;
  long long lda, ldb, ldc;
  long long kslice;       // k range per grid.z slice
  long long cslice;       // elements between the C of two slices
  const T* p1;       // T operand: P1 (ld Bn) and dY (ld O) of the chunk
  const T* dy;
  long long Bn;
  int O;
  int slices;             // k slices (set by gemm_launch)
  T* zstore;              // EPI_FWD of a training forward: Z[m][n] (ld ldc) is ALSO stored for the backward; NULL: not
};

// WTM MFMA tiles per wave along m (block tile 32 WTM x 64), KC = k chunk
enum { EPI_STORE = 0, EPI_FWD = 1, EPI_DP1 = 2 };
// Epilogues for Z = P0 x Core when O is a power of two <= 16 (a tile's 64 columns then hold whole groups of O):
//   EPI_FWD : partial[tile_n][w][o] = sum over the tile's columns n = (i1, o) of Z[w, n] P1[w, i1]   (Z never stored)
//   EPI_DP1 : dP1[w, i1] = sum_o dY[w, o] Z[w, (i1, o)]
template <int LA, int LB, int WTM, int KC, int EPI, typename T>
__global__ __launch_bounds__(256) void halves_gemm_k(const T* __restrict__ Ag, const T* __restrict__ Bg,
                                                  T* __restrict__ Cg, GemmD<T> g) {
  constexpr int BM = 32 * WTM, BN = GT;
  constexpr int APk = KC + 1, BPn = BN + 1;
  constexpr int UA = BM * KC / 256, UB = BN * KC / 256;   // staged elements per thread
  constexpr int LOGK = KC == 16 ? 4 : 5, LOGBM = BM == 64 ? 6 : 7;
  __shared__ T As[BM * APk];
  __shared__ T Bs[KC * BPn];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int lr = lane & 15, lk = lane >> 4;
  // XCD-aware tile order: workgroups go round-robin to the 8 XCDs (id % 8), each with its own L2.  XCD x takes
  // the contiguous tile range [x per, (x+1) per) with the n tile fastest, so the workgroups that share an A
  // row block (and are resident together) hit the same L2 instead of fetching it 8 times from memory.
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int total = tiles_n * tiles_m * g.slices, per = (total + 7) / 8;
  const int t = ((int)blockIdx.x % 8) * per + (int)blockIdx.x / 8;
  if (t >= total) return;
  const int bz = t / (tiles_n * tiles_m), trem = t - bz * tiles_n * tiles_m;
  const int m0 = (trem / tiles_n) * BM, n0 = (trem % tiles_n) * BN;
  const long long kbeg = (long long)bz * g.kslice;
  const long long kend = kbeg + g.kslice < g.K ? kbeg + g.kslice : g.K;
  constexpr bool AKF = LA != A_MFAST, BKF = LB == B_KFAST;
  // element e = tid + 256 u of an X x KC operand tile: k fastest: k = e & (KC-1), x = e >> LOGK; x fastest: x = e & (X-1), k = e / X
  const int kf_k = tid & (KC - 1), kf_x = tid >> LOGK;                     // x step 256 / KC
  const int axf_x = tid & (BM - 1), axf_k = tid >> LOGBM;                  // k step 256 / BM
  const int bxf_x = tid & (BN - 1), bxf_k = tid >> 6;                      // k step 4
  constexpr int KFS = 256 / KC, AXS = 256 / BM;
  T ra[UA], rb[UB], ra2[LA == A_T ? UA : 1], rb2[LB == B_T ? UB : 1];
  // The k-invariant part of every load address is formed once per lane (rows / columns clamped into range:
  // what the clamped rows produce lands in rows / columns of C that are never stored, so nothing is masked
  // outside the last, partial k chunk); inside the loop the wave-uniform k0 term is all that changes.
  long long aoff[UA], boff[UB];
  long long aoff2[LA == A_T ? UA : 1];
  long long boff2[LB == B_T ? UB : 1];
  int bt_i1 = 0, bt_o = 0;   // T operand on the B side: this thread's column n = (i1, o) is fixed
  if (LB == B_T) {
    const int n = min(n0 + bxf_x, g.N - 1);
    bt_i1 = n / g.O;
    bt_o = n - bt_i1 * g.O;
  }
#pragma unroll
  for (int u = 0; u < UA; ++u) {
    if (LA == A_KFAST) {
      aoff[u] = (long long)min(m0 + kf_x + KFS * u, g.M - 1) * g.lda + kf_k;            // + k0
    } else if (LA == A_MFAST) {
      aoff[u] = (long long)(axf_k + AXS * u) * g.lda + min(m0 + axf_x, g.M - 1);       // + k0 lda
    } else {
      const long long m = min(m0 + kf_x + KFS * u, g.M - 1);
      aoff[u] = m * g.Bn;                                                               // + (k0 + kf_k) / O
      aoff2[LA == A_T ? u : 0] = m * g.O;                                               // + (k0 + kf_k) % O
    }
  }
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    if (LB == B_NFAST) {
      boff[u] = (long long)(bxf_k + 4 * u) * g.ldb + min(n0 + bxf_x, g.N - 1);          // + k0 ldb
    } else if (LB == B_KFAST) {
      boff[u] = (long long)min(n0 + kf_x + KFS * u, g.N - 1) * g.ldb + kf_k;            // + k0
    } else {
      boff[u] = (long long)(bxf_k + 4 * u) * g.Bn + bt_i1;                              // + k0 Bn
      boff2[LB == B_T ? u : 0] = (long long)(bxf_k + 4 * u) * g.O + bt_o;               // + k0 O
    }
  }
  auto fetch = [&](long long k0, auto tailc) {
    constexpr bool tail = decltype(tailc)::value;
    // tail: the chunk crosses kend — clamp k into [0, K) (the staged value is zeroed afterwards)
    if (LA == A_T) {
      const int k = (int)(tail ? min(k0 + kf_k, (long long)g.K - 1) : k0 + kf_k);
      const int i1 = k / g.O, o = k - i1 * g.O;
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        ra[u] = g.p1[aoff[u] + i1];
        ra2[LA == A_T ? u : 0] = g.dy[aoff2[LA == A_T ? u : 0] + o];
      }
    } else {
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        if (LA == A_KFAST) {
          const long long kc = tail ? min(k0, (long long)g.K - 1 - kf_k) : k0;
          ra[u] = Ag[aoff[u] + kc];
        } else {
          const long long kc = tail ? min(k0, (long long)g.K - 1 - (axf_k + AXS * u)) : k0;
          ra[u] = Ag[aoff[u] + kc * g.lda];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (LB == B_NFAST) {
        const long long kc = tail ? min(k0, (long long)g.K - 1 - (bxf_k + 4 * u)) : k0;
        rb[u] = Bg[boff[u] + kc * g.ldb];
      } else if (LB == B_KFAST) {
        const long long kc = tail ? min(k0, (long long)g.K - 1 - kf_k) : k0;
        rb[u] = Bg[boff[u] + kc];
      } else {
        const long long kc = tail ? min(k0, (long long)g.K - 1 - (bxf_k + 4 * u)) : k0;
        rb[u] = g.p1[boff[u] + kc * g.Bn];
        rb2[LB == B_T ? u : 0] = g.dy[boff2[LB == B_T ? u : 0] + kc * g.O];
      }
    }
  };
  auto stage = [&](long long k0, auto tailc) {
    constexpr bool tail = decltype(tailc)::value;
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      T va = LA == A_T ? ra[u] * ra2[LA == A_T ? u : 0] : ra[u];
      if (AKF) {
        if (tail && !(k0 + kf_k < kend)) va = 0.0;
        As[(kf_x + KFS * u) * APk + kf_k] = va;
      } else {
        const int k = axf_k + AXS * u;
        if (tail && !(k0 + k < kend)) va = 0.0;
        As[axf_x * APk + k] = va;
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      T vb = LB == B_T ? rb[u] * rb2[LB == B_T ? u : 0] : rb[u];
      if (BKF) {
        if (tail && !(k0 + kf_k < kend)) vb = 0.0;
        Bs[kf_k * BPn + kf_x + KFS * u] = vb;
      } else {
        const int k = bxf_k + 4 * u;
        if (tail && !(k0 + k < kend)) vb = 0.0;
        Bs[k * BPn + bxf_x] = vb;
      }
    }
  };
  typedef typename Mma<T>::acc_t acc_t;
  acc_t acc[WTM][2];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
  // Full chunks run the mask-free code; only the last, partial chunk of the k range clamps its loads (a clamped
  // k re-reads an element of the same row / column) and zeroes what lies beyond kend.
  auto mma = [&]() {
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      T a[WTM], b[2];
#pragma unroll
      for (int i = 0; i < WTM; ++i) a[i] = As[(16 * WTM * wm + 16 * i + lr) * APk + 4 * kk + lk];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[(4 * kk + lk) * BPn + 32 * wn + 16 * j + lr];
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mma<T>::mfma(a[i], b[j], acc[i][j]);
    }
  };
  const std::integral_constant<bool, false> FULL;
  const std::integral_constant<bool, true> TAIL;
  const long long kfull = kbeg + (kend - kbeg) / KC * KC;   // end of the full chunks
  if (kbeg < kfull)
    fetch(kbeg, FULL);
  else if (kbeg < kend)
    fetch(kbeg, TAIL);
  for (long long k0 = kbeg; k0 < kfull; k0 += KC) {
    __syncthreads();
    stage(k0, FULL);
    __syncthreads();
    if (k0 + KC < kfull)
      fetch(k0 + KC, FULL);
    else if (k0 + KC < kend)
      fetch(k0 + KC, TAIL);
    mma();
  }
  if (kfull < kend) {
    __syncthreads();
    stage(kfull, TAIL);
    __syncthreads();
    mma();
  }
  if (EPI == EPI_FWD) {
    __shared__ T red[2 * BM * 16];
    const int logo = __ffs(g.O) - 1;
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 16 * WTM * wm + 16 * i + Mma<T>::row(v, lk), m = m0 + r;
        T sum = 0.0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + 32 * wn + 16 * j + lr;
          if (m < g.M && n < g.N) sum += acc[i][j][v] * g.p1[(long long)m * g.Bn + (n >> logo)];
        }
        // the 16 lanes of a row hold 16 consecutive columns: o = lr % O; sum the lanes of equal o
        for (int step = g.O; step < 16; step <<= 1) sum += __shfl_xor(sum, step, 64);
        if (lr < g.O) red[(wn * BM + r) * 16 + lr] = sum;
      }
    if (g.zstore) {
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int m = m0 + 16 * WTM * wm + 16 * i + Mma<T>::row(v, lk), n = n0 + 32 * wn + 16 * j + lr;
            if (m < g.M && n < g.N) g.zstore[(long long)m * g.ldc + n] = acc[i][j][v];
          }
    }
    __syncthreads();
    for (int e = tid; e < BM * g.O; e += 256) {
      const int r = e >> logo, o = e & (g.O - 1), m = m0 + r;
      if (m < g.M)
        Cg[((long long)(trem % tiles_n) * g.M + m) * g.O + o] = red[r * 16 + o] + red[(BM + r) * 16 + o];
    }
    return;
  }
  if (EPI == EPI_DP1) {
    const int logo = __ffs(g.O) - 1;
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int m = m0 + 16 * WTM * wm + 16 * i + Mma<T>::row(v, lk), n = n0 + 32 * wn + 16 * j + lr;
          const bool ok = m < g.M && n < g.N;
          T sum = ok ? acc[i][j][v] * g.dy[(long long)m * g.O + (n & (g.O - 1))] : 0.0;
          for (int step = 1; step < g.O; step <<= 1) sum += __shfl_xor(sum, step, 64);
          if (ok && (n & (g.O - 1)) == 0) Cg[(long long)m * g.Bn + (n >> logo)] = sum;
        }
    return;
  }
  T* C = Cg + (long long)bz * g.cslice;
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int m = m0 + 16 * WTM * wm + 16 * i + Mma<T>::row(v, lk), n = n0 + 32 * wn + 16 * j + lr;
        if (m < g.M && n < g.N) C[(long long)m * g.ldc + n] = acc[i][j][v];
      }
}

// 64 x 64 tiles, k chunks of 16: 128-row tiles or 32-deep chunks measured the same or slower (cfg1: 348 / 379 /
// 343 / 405 us forward for <2,16> / <2,32> / <4,16> / <4,32>), so the variant with the most workgroups is kept.
template <int LA, int LB, int EPI = EPI_STORE, typename T>
void gemm_launch(const T* A, const T* B, T* C, GemmD<T> g, int slices, hipStream_t st) {
  constexpr int WTM = 2, KC = 16;
  g.slices = slices;
  const int total = ((g.N + GT - 1) / GT) * ((g.M + 32 * WTM - 1) / (32 * WTM)) * slices;
  hipLaunchKernelGGL((halves_gemm_k<LA, LB, WTM, KC, EPI, T>), dim3((total + 7) / 8 * 8), dim3(256), 0, st, A, B, C, g);
}

// the fused epilogues need a tile's 64 columns to hold whole groups of O
bool fused_epilogue_ok(int O) { return O >= 1 && O <= 16 && (O & (O - 1)) == 0; }

// --------------------------------------------------------------------- contractions around the GEMMs
// out[w, o] = sum_i1 Z[w, i1, o] P1[w, i1]: one wave per window
template <typename T, typename S = T>
__global__ __launch_bounds__(256) void halves_fwd_contract_k(const T* __restrict__ Z, const S* __restrict__ P1,
                                                          S* __restrict__ out, long long nw, long long Bn, int O) {
  const int lane = threadIdx.x & 63;
  const long long wl = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wl >= nw) return;
  const T* z = Z + wl * Bn * O;
  const S* p1 = P1 + wl * Bn;
  for (int o = 0; o < O; ++o) {
    T s = 0.0;
    for (long long i1 = lane; i1 < Bn; i1 += 64) s += z[i1 * O + o] * (T)p1[i1];
    s = wave_reduce_sum(s);
    if (lane == 0) out[wl * O + o] = (S)s;
  }
}

// dP1[w, i1] = sum_o dY[w, o] Z[w, i1, o]
template <typename T, typename S = T>
__global__ __launch_bounds__(256) void halves_dp1_k(const T* __restrict__ Z, const S* __restrict__ dY,
                                                 T* __restrict__ dP1, long long nw, long long Bn, int O) {
  const long long total = nw * Bn;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long wl = idx / Bn;
    const T* z = Z + idx * O;
    const S* dy = dY + wl * O;
    T s = 0.0;
    for (int o = 0; o < O; ++o) s += (T)dy[o] * z[o];
    dP1[idx] = s;
  }
}

// gxw[(n Q + q)][w] = sum over the entries i of the factor's half whose digit of factor n is q of
// dP[w, i] * prod_{other factors d of the half} x_d[w, digit_d(i)].
// One WAVE per window (4 windows per workgroup, no workgroup barriers).  With factor f of the half at digit
// stride S_f = Q^(nd-1-f), i = (u Q + q) S_f + l and the product splits into PRE_f[u] (factors above f) and
// SUF_f[l] (factors below f): both families of Kronecker prefix / suffix products are built level by level in
// the wave's LDS (sum over levels < E entries each), so a term costs three LDS reads and two multiplies.
// Lane -> (pair (f, q), slice of the (u, l) range); slices are summed through LDS.
// floats of one PRE / SUF table of the dX kernel: levels 0 .. nd-1 with Q^f entries each = (E - 1) / (Q - 1) (< E / 2 from
// Q = 3 on: the tables used to be given E entries each, and the LDS they did not need cost resident workgroups)
__host__ __device__ inline int dx_half_table(int E, int Q) { return ((E - 1) / (Q - 1) + 4) & ~3; }
// position of gradient entry e in the wave's LDS row: one float of padding per 32, so that the lanes of a wave - which walk
// the row with the power-of-two strides of their factors - do not all land on one bank (counters: 68 % of the kernel's
// LDS cycles were bank conflicts and the LDS pipe was 78 % busy)
// (float32 rows only: on the float64 rows of cfg1 the extra index arithmetic cost more than the conflicts it removed)
__host__ __device__ inline int dx_half_pad(int e, size_t esz) { return esz == 4 ? e + (e >> 5) : e; }

// A saved forward (dctn_eps_fwd_save) leaves Z: the second half's dP1[w, i] = sum_o dY[w, o] Z[w, (i, o)] is then formed
// HERE, on the way into the wave's LDS row, instead of by a kernel of its own writing dP1 to memory and this one reading
// it back (float64 / float32 path, Z[w][i][o]: cfg1 backward 659 -> 622 us.  The bf16 path keeps a kernel of its own with
// 8-byte loads of its permuted Z': through this loader's 2-byte loads its layer-2 backward was slower).
struct DxSavedZ {
  const void* z;    // NULL: dP comes from memory
  const void* dy;   // dY rows of the chunk, storage dtype
  int O, mode;      // mode 1: Z[w][i][o]
};

template <int LOGQ, typename T, typename S = T>
__global__ __launch_bounds__(256) void halves_dx_half_k(const S* __restrict__ x, const T* __restrict__ dP,
                                                     T* __restrict__ gxw, HalfP h, int second, long long w0,
                                                     long long nw, DxSavedZ sz) {
  extern __shared__ __align__(16) unsigned char sm_raw[];
  T* sm = reinterpret_cast<T*>(sm_raw);
  const EpsP& p = h.p;
  const int Q = p.Q;
  const int base = second ? h.n0 : 0, nd = second ? h.n1 : h.n0;
  const int E = (int)(second ? h.Bn : h.A);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int TS = dx_half_table(E, Q);
  const int EP = (dx_half_pad(E, sizeof(T)) + 3) & ~3;
  const int per_wave = nd * Q + EP + 2 * TS + 64;
  T* xs = sm + (size_t)wv * per_wave;   // [nd][Q]
  T* dps = xs + nd * Q;                 // [E], padded (dx_half_pad)
  T* pre = dps + EP;                    // levels 0 .. nd-1, level f has Q^f entries
  T* suf = pre + TS;                    // levels nd-1 .. 0, level f has Q^(nd-1-f) entries
  T* red = suf + TS;                    // [64]
  const int np = nd * Q;
  const int ppl = np < 64 ? np : 64, nsl = 64 / ppl;
  const int hw = p.Ho * p.Wo;
  const int EQ = E / Q;
  for (long long wl = (long long)blockIdx.x * 4 + wv; wl < nw; wl += (long long)gridDim.x * 4) {
    const long long w = w0 + wl;
    const long long b = w / hw;
    const int rem = (int)(w - b * hw), ho = rem / p.Wo, wo = rem - ho * p.Wo;
    wave_sync_lds();   // the previous window's reads are done
    for (int e = lane; e < np; e += 64) {
      const int n = base + e / Q, q = e % Q;
      const int pos = n / p.C, ch = n - pos * p.C, dh = pos / p.K, dw = pos - dh * p.K;
      xs[e] = (T)x[ch * p.s[0] + b * p.s[1] + (long long)(ho + dh) * p.s[2] + (long long)(wo + dw) * p.s[3] + q * p.s[4]];
    }
    if (sz.mode == 0) {
      for (int e = lane; e < E; e += 64) dps[dx_half_pad(e, sizeof(T))] = dP[wl * E + e];
    } else {
      const S* zrow = reinterpret_cast<const S*>(sz.z) + wl * (long long)E * sz.O;
      const S* dyr = reinterpret_cast<const S*>(sz.dy) + wl * sz.O;
      for (int e = lane; e < E; e += 64) {
        T acc = 0.0;
        for (int o = 0; o < sz.O; ++o) acc += (T)dyr[o] * (T)zrow[(long long)e * sz.O + o];
        dps[dx_half_pad(e, sizeof(T))] = acc;
      }
    }
    if (lane == 0) pre[0] = 1.0, suf[0] = 1.0;   // level 0 of PRE (factor 0), level nd-1 of SUF (last factor)
    wave_sync_lds();
    // PRE level f (offset (Q^f - 1)/(Q - 1)): pre_f[u Q + q'] = pre_{f-1}[u] x_{f-1}[q']
    // SUF level g counted from the last factor (offset likewise): suf_g[q' Q^(g-1) + l] = x_{nd-g}[q'] suf_{g-1}[l]
    {
      int off_prev = 0, size_prev = 1;
      for (int f = 1; f < nd; ++f) {
        const int off = off_prev + size_prev, size = size_prev * Q;
        for (int e = lane; e < size; e += 64) {
          int u, qq;
          if (LOGQ > 0) {
            u = e >> LOGQ;
            qq = e & ((1 << LOGQ) - 1);
          } else {
            u = e / Q;
            qq = e - u * Q;
          }
          pre[off + e] = pre[off_prev + u] * xs[(f - 1) * Q + qq];
          const int q2 = e / size_prev, l = e - q2 * size_prev;
          suf[off + e] = xs[(nd - f) * Q + q2] * suf[off_prev + l];
        }
        off_prev = off;
        size_prev = size;
        wave_sync_lds();
      }
    }
    for (int pbase = 0; pbase < np; pbase += ppl) {
      const int pr = pbase + lane % ppl, sl = lane / ppl;
      T acc = 0.0;
      if (pr < np && sl < nsl) {
        const int fd = pr / Q, fq = pr - fd * Q;
        // stride of factor fd, offsets of its PRE level (fd) and SUF level (nd-1-fd)
        int stride = 1, off_suf = 0;
        for (int d = nd - 1; d > fd; --d) {
          off_suf += stride;
          stride *= Q;
        }
        int off_pre = 0, sz = 1;
        for (int d = 0; d < fd; ++d) {
          off_pre += sz;
          sz *= Q;
        }
        int lstride = 0;
        if (LOGQ > 0) lstride = LOGQ * (nd - 1 - fd);
#pragma unroll 4
        for (int j = sl; j < EQ; j += nsl) {   // (unrolled: the three LDS reads of four terms are in flight together)
          int u, l;
          if (LOGQ > 0) {
            u = j >> lstride;
            l = j & (stride - 1);
          } else {
            u = j / stride;
            l = j - u * stride;
          }
          acc += dps[dx_half_pad((u * Q + fq) * stride + l, sizeof(T))] * pre[off_pre + u] * suf[off_suf + l];
        }
      }
      wave_sync_lds();
      red[lane] = acc;
      wave_sync_lds();
      if (lane < ppl && pbase + lane < np) {
        T s = 0.0;
        for (int k = 0; k < nsl; ++k) s += red[k * ppl + lane];
        const int prr = pbase + lane;
        gxw[(long long)(base * Q + prr) * p.Wn + w] = s;
      }
    }
  }
}

// ---- cross-lane sums for the butterfly below, on the VALU (DPP / permlane swaps) - ds_bpermute-based shuffles put ~200
// LDS-crossbar operations on every window and the LDS unit of a CU became the limit (87 us per half at cfg1, as slow as the
// table form).  A 64-bit value moves as two 32-bit halves.
typedef __attribute__((ext_vector_type(2))) int hv_i2;
template <int CTRL>
__device__ __forceinline__ float hv_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double hv_dpp(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// lanes 0-31: a summed over (lane, lane + 32); lanes 32-63: b likewise
__device__ __forceinline__ float hv_pair32(float a, float b) {
  const hv_i2 r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ double hv_pair32(double a, double b) {
  const long long x = __double_as_longlong(a), y = __double_as_longlong(b);
  const hv_i2 lo = __builtin_amdgcn_permlane32_swap((int)x, (int)y, false, false);
  const hv_i2 hi = __builtin_amdgcn_permlane32_swap((int)(x >> 32), (int)(y >> 32), false, false);
  return __longlong_as_double(((long long)hi[0] << 32) | (unsigned)lo[0]) + __longlong_as_double(((long long)hi[1] << 32) | (unsigned)lo[1]);
}
// even rows of 16 lanes: a summed over the row pair; odd rows: b
__device__ __forceinline__ float hv_pair16(float a, float b) {
  const hv_i2 r = __builtin_amdgcn_permlane16_swap(__float_as_int(a), __float_as_int(b), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ double hv_pair16(double a, double b) {
  const long long x = __double_as_longlong(a), y = __double_as_longlong(b);
  const hv_i2 lo = __builtin_amdgcn_permlane16_swap((int)x, (int)y, false, false);
  const hv_i2 hi = __builtin_amdgcn_permlane16_swap((int)(x >> 32), (int)(y >> 32), false, false);
  return __longlong_as_double(((long long)hi[0] << 32) | (unsigned)lo[0]) + __longlong_as_double(((long long)hi[1] << 32) | (unsigned)lo[1]);
}
// v + (v of the lane across bit BIT), BIT = 0..5
template <int BIT, typename T>
__device__ __forceinline__ T hv_xsum(T v) {
  if constexpr (BIT == 0) return v + hv_dpp<0xB1>(v);                       // quad_perm [1,0,3,2]
  else if constexpr (BIT == 1) return v + hv_dpp<0x4E>(v);                  // quad_perm [2,3,0,1]
  else if constexpr (BIT == 2) return v + hv_dpp<0x1B>(hv_dpp<0x141>(v));   // half-row mirror (xor 7), then quad reverse (xor 3)
  else if constexpr (BIT == 3) return v + hv_dpp<0x141>(hv_dpp<0x140>(v));  // row mirror (xor 15), then half-row mirror (xor 7)
  else if constexpr (BIT == 4) return hv_pair16(v, v);
  else return hv_pair32(v, v);
}

// ---- the same gradient for Q = 2 halves of 6..8 factors, without tables: a wave-level butterfly.
// F(x_0 .. x_(nd-1)) = sum_e dP[e] prod_f x_f[digit_f(e)] is multilinear; reverse mode over the contraction tree
//   u_nd = dP,   u_f[prefix] = u_(f+1)[prefix, 0] x_f[0] + u_(f+1)[prefix, 1] x_f[1]      (up-sweep, last factor first)
//   dF/dx_f[q] = sum_prefix ( prod_(g < f) x_g[prefix_g] ) u_(f+1)[prefix, q]
// costs ~2 E multiply-adds where the table form above spends 2 nd E (and 3 LDS reads per term).  Lane = the six leading
// digits (factor 0 = lane bit 5 .. factor 5 = lane bit 0), the remaining nd - 6 factors are the lane's own 1 / 2 / 4 values;
// the six cross-lane levels exchange one value with the lane across one bit; the 2 nd results are wave sums.
// One wave per window; everything in registers; cfg1 (two 8-factor halves, float64): 2 x 60 us (the table form: 2 x 88;
// without its loads or without its stores the kernel takes 56-59 us: f64 vector arithmetic and address work, not memory).
// element offsets of the half's factors relative to the window's top-left pixel, formed on the host: formed in the
// kernel (two divisions per factor, everything wave-uniform) they were ~600 scalar instructions per window and wave -
// a wave takes one window
struct DxQ2Off { long long off[8]; };
template <int ND, typename T, typename S>
__global__ __launch_bounds__(256) void halves_dx_half_q2_k(const S* __restrict__ x, const T* __restrict__ dP,
                                                        T* __restrict__ gxw, HalfP h, int second, long long w0,
                                                        long long nw, DxSavedZ sz, DxQ2Off fo) {
  constexpr int NI = ND - 6, VPL = 1 << NI, E = 1 << ND;
  const EpsP& p = h.p;
  const int lane = threadIdx.x & 63;
  const int base = second ? h.n0 : 0;
  const int hw = p.Ho * p.Wo;
  const long long wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  for (long long wl = wave0; wl < nw; wl += (long long)gridDim.x * 4) {
    const long long w = w0 + wl;
    const long long b = w / hw;
    const int rem = (int)(w - b * hw), ho = rem / p.Wo, wo = rem - ho * p.Wo;
    T xv[ND][2];
    const S* win = x + b * p.s[1] + (long long)ho * p.s[2] + (long long)wo * p.s[3];
#pragma unroll
    for (int f = 0; f < ND; ++f) {
      const S* px = win + fo.off[f];
      xv[f][0] = (T)px[0];
      xv[f][1] = (T)px[p.s[4]];
    }
    T d[VPL];
    if (sz.mode == 0) {
#pragma unroll
      for (int j = 0; j < VPL; ++j) d[j] = dP[wl * E + lane * VPL + j];
    } else {
      const S* zrow = reinterpret_cast<const S*>(sz.z) + wl * (long long)E * sz.O;
      const S* dyr = reinterpret_cast<const S*>(sz.dy) + wl * sz.O;
#pragma unroll
      for (int j = 0; j < VPL; ++j) {
        T acc = 0.0;
        for (int o = 0; o < sz.O; ++o) acc += (T)dyr[o] * (T)zrow[(long long)(lane * VPL + j) * sz.O + o];
        d[j] = acc;
      }
    }
    // ---- in-lane levels (factors ND-1 .. 6): u7[i6] and u6 (ND = 8), u6 (ND = 7)
    T u7[2] = {0, 0}, v;
    if constexpr (NI == 2) {
      u7[0] = d[0] * xv[7][0] + d[1] * xv[7][1];
      u7[1] = d[2] * xv[7][0] + d[3] * xv[7][1];
      v = u7[0] * xv[6][0] + u7[1] * xv[6][1];
    } else if constexpr (NI == 1) {
      v = d[0] * xv[6][0] + d[1] * xv[6][1];
    } else {
      v = d[0];
    }
    // ---- six cross-lane levels: bit b of the lane is the digit of factor 5 - b
    T vb[6], wgt[6];
    auto level = [&](auto bc) {   // contract factor 5 - b: both lanes of a pair add their own term (t + the partner's t)
      constexpr int bb = decltype(bc)::value;
      vb[bb] = v;
      const T t = v * (((lane >> bb) & 1) ? xv[5 - bb][1] : xv[5 - bb][0]);
      v = hv_xsum<bb>(t);
    };
    level(std::integral_constant<int, 0>{});
    level(std::integral_constant<int, 1>{});
    level(std::integral_constant<int, 2>{});
    level(std::integral_constant<int, 3>{});
    level(std::integral_constant<int, 4>{});
    level(std::integral_constant<int, 5>{});
    // ---- the weights of the prefixes: wgt[b] = prod over the lane bits above b of that factor's value
    T Wp = 1.0;
#pragma unroll
    for (int bb = 5; bb >= 0; --bb) {
      wgt[bb] = Wp;
      Wp *= ((lane >> bb) & 1) ? xv[5 - bb][1] : xv[5 - bb][0];
    }
    // ---- per-lane terms of the 2 ND results, then their wave sums
    T pg[ND][2];
#pragma unroll
    for (int bb = 0; bb < 6; ++bb) {
      const bool rep = (lane & ((1 << bb) - 1)) == 0;   // one lane per (prefix, digit): the lower bits hold copies
      const int mine = (lane >> bb) & 1;
      const T t = rep ? wgt[bb] * vb[bb] : (T)0.0;
      pg[5 - bb][0] = mine ? (T)0.0 : t;
      pg[5 - bb][1] = mine ? t : (T)0.0;
    }
    if constexpr (NI == 2) {
      pg[6][0] = Wp * u7[0];
      pg[6][1] = Wp * u7[1];
      pg[7][0] = Wp * (xv[6][0] * d[0] + xv[6][1] * d[2]);
      pg[7][1] = Wp * (xv[6][0] * d[1] + xv[6][1] * d[3]);
    } else if constexpr (NI == 1) {
      pg[6][0] = Wp * d[0];
      pg[6][1] = Wp * d[1];
    }
    // the wave sums of the (up to) 16 values k = 2 f + q: two register-halving levels (lane halves take values k / k + 8,
    // then row pairs k / k + 4), then the 16 lanes of a row: row rho ends with values 8 (rho >> 1) + 4 (rho & 1) + j
    T vals[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) vals[k] = k < 2 * ND ? pg[k < 2 * ND ? k / 2 : 0][k & 1] : (T)0.0;
    T t8[8], t4[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) t8[j] = hv_pair32(vals[j], vals[j + 8]);
#pragma unroll
    for (int j = 0; j < 4; ++j) t4[j] = hv_pair16(t8[j], t8[j + 4]);
#pragma unroll
    for (int j = 0; j < 4; ++j) t4[j] = hv_xsum<3>(hv_xsum<2>(hv_xsum<1>(hv_xsum<0>(t4[j]))));
    if ((lane & 15) == 0) {
      const int k0 = 8 * (lane >> 5) + 4 * ((lane >> 4) & 1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k0 + j < 2 * ND) gxw[(long long)(base * 2 + k0 + j) * p.Wn + w] = t4[j];
    }
  }
}

// dCore (+)= sum of the split-k partial products
template <typename T, typename S = T>
__global__ __launch_bounds__(256) void halves_sum_partials_k(const T* __restrict__ part, S* __restrict__ dCore,
                                                          long long n, int slices, int accumulate) {
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
    T s = accumulate ? (T)dCore[idx] : (T)0;
    for (int k = 0; k < slices; ++k) s += part[(long long)k * n + idx];
    dCore[idx] = (S)s;
  }
}

int ilog2_pow2(int q) {
  int l = 0;
  while ((1 << l) < q) ++l;
  return (1 << l) == q ? l : 0;
}

unsigned blocks_for(long long n, int per) {
  long long b = (n + per - 1) / per;
  if (b > 262144) b = 262144;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <int LOGQ, typename T, typename S>
int launch_halves_q(const S* x, S* P0, S* P1, const HalfP& h, long long w0, long long nw, hipStream_t st) {
  // per wave: the window's features + the four quarter tables
  size_t ntab = 0;
  for (int s2 = 0; s2 < 2; ++s2) {
    const int nd = s2 ? h.n1 : h.n0, nlo = nd / 2;
    ntab += (size_t)ipow_ll(h.p.Q, nd - nlo) + (size_t)ipow_ll(h.p.Q, nlo);
  }
  const size_t lds = (size_t)4 * ((size_t)h.p.N * h.p.Q + ntab) * sizeof(T);
  if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)halves_k<LOGQ, T, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((halves_k<LOGQ, T, S>), dim3(blocks_for(nw, 4)), dim3(256), lds, st, x, P0, P1, h, w0, nw);
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

template <typename T, typename S = T>
int launch_halves(const S* x, S* P0, S* P1, const HalfP& h, long long w0, long long nw, hipStream_t st) {
  switch (ilog2_pow2(h.p.Q)) {
    case 1: return launch_halves_q<1, T, S>(x, P0, P1, h, w0, nw, st);
    case 2: return launch_halves_q<2, T, S>(x, P0, P1, h, w0, nw, st);
    case 3: return launch_halves_q<3, T, S>(x, P0, P1, h, w0, nw, st);
    default: return launch_halves_q<0, T, S>(x, P0, P1, h, w0, nw, st);
  }
}

size_t dx_half_lds(const HalfP& h, int second, size_t esz) {
  const int nd = second ? h.n1 : h.n0;
  const long long E = second ? h.Bn : h.A;
  return (size_t)4 * ((size_t)nd * h.p.Q + (size_t)((dx_half_pad((int)E, esz) + 3) & ~3) + 2 * (size_t)dx_half_table((int)E, h.p.Q) + 64) * esz;
}

template <int LOGQ, typename T, typename S>
int launch_dx_half_q(const S* x, const T* dP, T* gxw, const HalfP& h, int second, long long w0,
                     long long nw, hipStream_t st, DxSavedZ sz) {
  const size_t lds = dx_half_lds(h, second, sizeof(T));
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)halves_dx_half_k<LOGQ, T, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((halves_dx_half_k<LOGQ, T, S>), dim3(blocks_for(nw, 4)), dim3(256), lds, st, x, dP, gxw, h, second, w0, nw, sz);
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

template <typename T, typename S = T>
int launch_dx_half(const S* x, const T* dP, T* gxw, const HalfP& h, int second, long long w0,
                   long long nw, hipStream_t st, DxSavedZ sz = DxSavedZ{nullptr, nullptr, 0, 0}) {
  const int ndh = second ? h.n1 : h.n0;
  if (h.p.Q == 2 && ndh >= 6 && ndh <= 8) {   // binary halves of 6..8 factors: the register butterfly
    const dim3 g(blocks_for(nw, 4)), b(256);
    DxQ2Off fo;
    for (int f = 0; f < 8; ++f) {
      const int n = (second ? h.n0 : 0) + (f < ndh ? f : 0);
      const int pos = n / h.p.C, ch = n - pos * h.p.C, dh = pos / h.p.K, dw = pos - dh * h.p.K;
      fo.off[f] = (long long)ch * h.p.s[0] + (long long)dh * h.p.s[2] + (long long)dw * h.p.s[3];
    }
    if (ndh == 8) hipLaunchKernelGGL((halves_dx_half_q2_k<8, T, S>), g, b, 0, st, x, dP, gxw, h, second, w0, nw, sz, fo);
    else if (ndh == 7) hipLaunchKernelGGL((halves_dx_half_q2_k<7, T, S>), g, b, 0, st, x, dP, gxw, h, second, w0, nw, sz, fo);
    else hipLaunchKernelGGL((halves_dx_half_q2_k<6, T, S>), g, b, 0, st, x, dP, gxw, h, second, w0, nw, sz, fo);
    DCTN_CHECK_LAUNCH();
    return DCTN_OK;
  }
  switch (ilog2_pow2(h.p.Q)) {
    case 1: return launch_dx_half_q<1, T, S>(x, dP, gxw, h, second, w0, nw, st, sz);
    case 2: return launch_dx_half_q<2, T, S>(x, dP, gxw, h, second, w0, nw, st, sz);
    case 3: return launch_dx_half_q<3, T, S>(x, dP, gxw, h, second, w0, nw, st, sz);
    default: return launch_dx_half_q<0, T, S>(x, dP, gxw, h, second, w0, nw, st, sz);
  }
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- host side, per element type
bool halves_shape_ok(const EpsP& p, size_t esz) {
  if (p.N < 2) return false;
  if (p.R * p.O < 2048 || p.Wn < 64) return false;
  const HalfP h = make_half(p, esz);
  if (h.Bn > 1024 || h.A > 1024 || h.NB > (1ll << 20)) return false;   // per-wave LDS tables of the dX kernel; int-sized GEMM dims
  if (p.Wn >= (1ll << 31)) return false;
  // per-workgroup LDS of the halves kernel (features + quarter tables) and of the dX kernel (3 E per wave)
  if ((size_t)4 * ((size_t)p.N * p.Q + 2 * (size_t)(h.A + h.Bn)) * esz > DCTN_LDS_BUDGET) return false;
  if ((size_t)4 * ((size_t)(h.n1 > h.n0 ? h.n1 : h.n0) * p.Q + 3 * (size_t)(h.Bn > h.A ? h.Bn : h.A) + 64) * esz > DCTN_LDS_BUDGET) return false;
  return true;
}

// ------------------------------------------------------------------------------------------------
// bf16 storage on the bf16 matrix cores: the same products with v_mfma_f32_16x16x32_bf16 (float32 accumulate).
// The instruction wants 8 consecutive k per lane for BOTH operands, so both LDS tiles are kept [row][k] and every
// operand is read from memory 8 elements (16 bytes) at a time.  To make that possible for any O the core is
// re-laid once per call with its output index in front of the half index:
//     coreP[i0][(o, i1)] = core[i0][(i1, o)]        coreQ[i1][(o, i0)] = core[i0][(i1, o)]
//   forward : Z' = P0 x coreP, a 64-column tile then holds ONE o and 64 consecutive i1: the epilogue multiplies by
//             P1[w, i1] and sums the tile's columns -> partial[tile][w]          (Z never stored, any O)
//   dCore   : dCoreP[i0][(o, i1)] = sum_w P0[w, i0] dY[w, o] P1[w, i1]  (k = window, split-k; un-permuted at the end)
//   dP0     : sum_(o, i1) (dY[w, o] P1[w, i1]) coreP[i0][(o, i1)]          dP1 : sum_(o, i0) (dY[w, o] P0[w, i0]) coreQ[i1][(o, i0)]
// A "scaled" operand is one 16-byte load of the half's row times one scalar of dY, formed in float32 and rounded once.
// Sources whose natural layout is contiguous along the row index instead of k get a k-contiguous copy first (coreT =
// coreP transposed for Z'; P0^T, P1^T, dY^T per chunk for dCore, whose operand is then the elementwise product of two
// 16-byte loads) — staging a row-contiguous source through eight 2-byte LDS writes per thread cost 25 % of the forward.
// Accumulator rows as in the float32 instruction.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;

constexpr int BK = 64;        // k per chunk (two MFMA steps per barrier pair)
// workgroup tile BM x BN: 2 x 2 waves of (16 TI) x (16 TJ).  128 x 64 per wave: 24 KB of LDS fragment reads per 64
// MFMAs (64 x 64 per wave read 16 KB per 32, and with 12 resident waves the LDS pipe, not the matrix pipe, was the
// limit: 0.22-0.30 of the bf16 peak)
constexpr int TI = 4, TJ = 4;
constexpr int BM = 32 * TI, BN = 32 * TJ;

enum { BA_KFAST = 0, BA_SCALED = 1 };   // A[m][k] rows / sc[m][o] * V[m][j], k = o KV + j
enum { BB_KFAST = 0, BB_PROD = 1 };     // B stored [n][k] / scT[o][k..] * VT[j][k..] for row n = (o, j), both k-contiguous
enum { BEPI_STORE = 0, BEPI_FWD = 1 };

struct GemmB {
  int M, N, K;
  long long lda, ldb;
  long long kslice, cslice;
  const bf16_t* vec;   // scaled / product operand: rows of P0 / P1 (or their transposes) and dY (or its transpose)
  const bf16_t* sc;
  long long ldv;
  int KV, O;
  const bf16_t* p1;    // forward epilogue: P1 (ld Bn)
  int Bn;
  int slices;
  // BEPI_FWD of a training forward: Z'[m][n] is also stored (bf16, ld N) for the backward, each block of 64 columns in
  // the order the accumulators hold it - column 16 j + lr of the block at position 4 lr + j, one 8-byte store per lane
  bf16_t* zstore;
};

__device__ __forceinline__ bf16x8v zero8() { return bf16x8v{0, 0, 0, 0, 0, 0, 0, 0}; }

__device__ __forceinline__ bf16x8v scale8(bf16x8v v, float s) {
  bf16x8v r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16_t)((float)v[j] * s);
  return r;
}

// C[m, n] = sum_k A(m, k) B(n, k), float32 out; both operands k-contiguous, K (and KV) multiples of 32.
// 128 x 128 tile, 64-deep chunks: per chunk a thread stages 4 vectors of 8 per operand (rows r, r + 64; k steps 0, 32)
// and a wave issues 2 x 16 MFMAs on 8 + 8 operand vectors read from LDS.
template <int LA, int LB, int EPI>
__global__ __launch_bounds__(256) void bf16_gemm_k(const bf16_t* __restrict__ Ag, const bf16_t* __restrict__ Bg,
                                                   float* __restrict__ Cg, GemmB g) {
  // LDS rows of 128 bytes (64 k), the 16-byte slot XORed with bits 1..3 of the row: every 16-lane group of a ds_read_b128
  // (8 rows at slot s, 8 rows at slot s ^ 1: MI355X_MICROARCH.md LDS table) then covers all 64 banks.  Padded rows
  // (136 bytes: 8-byte accesses only; 144 bytes: 2-way conflicts on exactly those groups, 47 % of the LDS cycles) lost.
  constexpr int BROW = BK;
  // (the row's parity enters the XOR too: stores are banked mod 32, so rows 2j and 2j + 1 - 128 bytes apart, written by
  // one 8-lane group - would otherwise share their four slots: 2-way conflicts on every ds_write_b128, a third of the
  // kernel's LDS cycles)
  auto sw = [](int row, int k) { return row * BK + ((((k >> 3) ^ (row >> 1) ^ ((row & 1) << 2)) & 7) << 3) + (k & 7); };   // element offset
  constexpr int UA = BM / 64, UB = BN / 64;   // staged row groups per thread
  __shared__ __align__(16) bf16_t smem[(BM + BN) * BROW];
  bf16_t* As = smem;               // [m][k]
  bf16_t* Bs = smem + BM * BROW;   // [n][k]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int total = tiles_n * tiles_m * g.slices, per = (total + 7) / 8;
  const int t = ((int)blockIdx.x % 8) * per + (int)blockIdx.x / 8;   // XCD-aware tile order (see halves_gemm_k)
  if (t >= total) return;
  const int bz = t / (tiles_n * tiles_m), trem = t - bz * tiles_n * tiles_m;
  const int m0 = (trem / tiles_n) * BM, n0 = (trem % tiles_n) * BN;
  const long long kbeg = (long long)bz * g.kslice;
  const long long kend = kbeg + g.kslice < g.K ? kbeg + g.kslice : g.K;
  const int kf_r = tid >> 2, kf_k8 = (tid & 3) * 8;   // row (+64 u), 8 k's (+32 s)
  bf16x8v ra[UA][2], rb[UB][2];
  float sa[UA][2];
  auto fetch = [&](long long kc0) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const long long k = min(kc0 + 32 * s2 + kf_k8, (long long)g.K - 8);
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        const long long m = min(m0 + kf_r + 64 * u, g.M - 1);
        if (LA == BA_KFAST) {
          ra[u][s2] = *reinterpret_cast<const bf16x8v*>(Ag + m * g.lda + k);
        } else {   // sc[m][o] * V[m][j .. j+7], k = o KV + j (a 32-wide step lies inside one o)
          const int o = (int)(k / g.KV), j = (int)(k - (long long)o * g.KV);
          ra[u][s2] = *reinterpret_cast<const bf16x8v*>(g.vec + m * g.ldv + j);
          sa[u][s2] = (float)g.sc[m * g.O + o];
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (LB == BB_KFAST) {
          const long long n = min(n0 + kf_r + 64 * u, g.N - 1);
          rb[u][s2] = *reinterpret_cast<const bf16x8v*>(Bg + n * g.ldb + k);
        } else {   // row n = (o, j): scT[o][k .. k+7] * VT[j][k .. k+7], both stored with ld ldv
          const int n = min(n0 + kf_r + 64 * u, g.N - 1);
          const int o = n / g.KV, j = n - o * g.KV;
          const bf16x8v v = *reinterpret_cast<const bf16x8v*>(g.vec + (long long)j * g.ldv + k);
          const bf16x8v c = *reinterpret_cast<const bf16x8v*>(g.sc + (long long)o * g.ldv + k);
#pragma unroll
          for (int e = 0; e < 8; ++e) rb[u][s2][e] = (bf16_t)((float)v[e] * (float)c[e]);
        }
      }
    }
  };
  auto put8 = [&](bf16_t* dst, bf16x8v v) { *reinterpret_cast<bf16x8v*>(dst) = v; };   // 16-byte aligned destination
  auto stage = [&](long long kc0) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bool in = kc0 + 32 * s2 + kf_k8 < kend;
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        const bf16x8v va = LA == BA_SCALED ? scale8(ra[u][s2], sa[u][s2]) : ra[u][s2];
        put8(As + sw(kf_r + 64 * u, 32 * s2 + kf_k8), in ? va : zero8());
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) put8(Bs + sw(kf_r + 64 * u, 32 * s2 + kf_k8), in ? rb[u][s2] : zero8());
    }
  };
  auto get8 = [&](const bf16_t* src) { return *reinterpret_cast<const bf16x8v*>(src); };
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (kbeg < kend) fetch(kbeg);
  for (long long k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();
    stage(k0);
    __syncthreads();
    if (k0 + BK < kend) fetch(k0 + BK);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8v b[TJ];
#pragma unroll
      for (int j = 0; j < TJ; ++j) b[j] = get8(Bs + sw(16 * TJ * wn + 16 * j + lr, 32 * s2 + 8 * lk));
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const bf16x8v a = get8(As + sw(16 * TI * wm + 16 * i + lr, 32 * s2 + 8 * lk));
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  if (EPI == BEPI_FWD) {
    // a wave's 16 TJ = 64 columns are (o, i1 .. i1 + 63) for ONE o (Bn is a multiple of 64): partial[n / 64][w] = sum over
    // those columns of Z'[w, n] P1[w, i1(n)] - no other wave shares the (row, column block), so it is written directly
    static_assert(TJ == 4, "the forward epilogue writes one partial per 64 columns");
    const int nb = n0 + 64 * wn;   // first column of this wave
    if (nb < g.N) {
      const int i1_0 = nb % g.Bn;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int mrow = m0 + 16 * TI * wm + 16 * i + 4 * lk + v, m = min(mrow, g.M - 1);
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < TJ; ++j) sum += acc[i][j][v] * (float)g.p1[(long long)m * g.Bn + i1_0 + 16 * j + lr];
#pragma unroll
          for (int step = 1; step < 16; step <<= 1) sum += __shfl_xor(sum, step, 64);
          if (lr == 0 && mrow < g.M) Cg[(long long)(nb / 64) * g.M + mrow] = sum;
          if (g.zstore && mrow < g.M) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
            const bf16x4v zv = {(bf16_t)acc[i][0][v], (bf16_t)acc[i][1][v], (bf16_t)acc[i][2][v], (bf16_t)acc[i][3][v]};
            *reinterpret_cast<bf16x4v*>(g.zstore + (long long)mrow * g.N + nb + 4 * lr) = zv;
          }
        }
    }
    return;
  }
  float* C = Cg + (long long)bz * g.cslice;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int m = m0 + 16 * TI * wm + 16 * i + 4 * lk + v, n = n0 + 16 * TJ * wn + 16 * j + lr;
        if (m < g.M && n < g.N) C[(long long)m * g.N + n] = acc[i][j][v];
      }
}

template <int LA, int LB, int EPI = BEPI_STORE>
void bf16_gemm_launch(const bf16_t* A, const bf16_t* B, float* C, GemmB g, int slices, hipStream_t st) {
  g.slices = slices;
  const int total = ((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM) * slices;
  hipLaunchKernelGGL((bf16_gemm_k<LA, LB, EPI>), dim3((total + 7) / 8 * 8), dim3(256), 0, st, A, B, C, g);
}

// coreP[i0][o Bn + i1] = core[i0][i1 O + o] (and, with which = 1, coreQ[i1][o A + i0])
__global__ __launch_bounds__(256) void bf16_core_permute_k(const bf16_t* __restrict__ core, bf16_t* __restrict__ dst,
                                                           long long A, long long Bn, int O, int which) {
  const long long total = A * Bn * O;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    // idx runs over the DESTINATION (coalesced writes)
    if (which == 0) {
      const long long i0 = idx / (Bn * O), rem = idx - i0 * Bn * O;
      const long long o = rem / Bn, i1 = rem - o * Bn;
      dst[idx] = core[(i0 * Bn + i1) * O + o];
    } else if (which == 1) {
      const long long i1 = idx / (A * O), rem = idx - i1 * A * O;
      const long long o = rem / A, i0 = rem - o * A;
      dst[idx] = core[(i0 * Bn + i1) * O + o];
    } else {   // coreT[(o, i1)][i0]
      const long long n = idx / A, i0 = idx - n * A;
      const long long o = n / Bn, i1 = n - o * Bn;
      dst[idx] = core[(i0 * Bn + i1) * O + o];
    }
  }
}

// dst[c][r] = src[r][c] for r < R, zero for R <= r < ldd (ldd: row length of dst, a multiple of 64)
__global__ __launch_bounds__(256) void bf16_transpose_k(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                        long long R, int C, long long ldd) {
  __shared__ bf16_t tile[64][66];
  const long long r0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int rr = e >> 6, cc = e & 63;
    tile[rr][cc] = (r0 + rr < R && c0 + cc < C) ? src[(r0 + rr) * C + c0 + cc] : (bf16_t)0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int cc = e >> 6, rr = e & 63;
    if (c0 + cc < C && r0 + rr < ldd) dst[(long long)(c0 + cc) * ldd + r0 + rr] = tile[rr][cc];
  }
}

// dCore[i0][i1 O + o] = (bf16) dCoreP32[i0][o Bn + i1]
__global__ __launch_bounds__(256) void bf16_dcore_unpermute_k(const float* __restrict__ src, bf16_t* __restrict__ dCore,
                                                              long long A, long long Bn, int O) {
  const long long total = A * Bn * O;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long i0 = idx / (Bn * O), rem = idx - i0 * Bn * O;
    const long long i1 = rem / O, o = rem - i1 * O;
    dCore[idx] = (bf16_t)src[(i0 * O + o) * Bn + i1];
  }
}

// out[w][o] = sum over the Bn / 64 column tiles of output o of partial[tile][w]
__global__ __launch_bounds__(256) void bf16_out_sum_k(const float* __restrict__ part, bf16_t* __restrict__ out,
                                                      long long nw, int O, int tiles_per_o) {
  const long long total = nw * O;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long w = idx / O;
    const int o = (int)(idx - w * O);
    float s = 0.f;
    for (int tt = 0; tt < tiles_per_o; ++tt) s += part[(long long)(o * tiles_per_o + tt) * nw + w];
    out[idx] = (bf16_t)s;
  }
}

// dP1[w, i1] = sum_o dY[w, o] Z'[w, (o, i1)] from the Z' a training forward stored (GemmB::zstore: blocks of 64 columns
// in accumulator order).  A thread takes 4 stored positions of a block (one 8-byte load per o) = columns 16 j + lr.
__global__ __launch_bounds__(256) void bf16_dp1_saved_k(const bf16_t* __restrict__ Zs, const bf16_t* __restrict__ dY,
                                                        float* __restrict__ dP1, long long nw, int Bn, int O) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
  const int per_w = Bn / 4;
  const long long total = nw * per_w;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long w = idx / per_w;
    const int t = (int)(idx - w * per_w), blk = t >> 4, lr = t & 15;
    const bf16_t* z = Zs + w * (long long)Bn * O + blk * 64 + 4 * lr;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int o = 0; o < O; ++o) {
      const bf16x4v zv = *reinterpret_cast<const bf16x4v*>(z + (long long)o * Bn);
      const float d = (float)dY[w * O + o];
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += d * (float)zv[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) dP1[w * Bn + blk * 64 + 16 * j + lr] = s[j];
  }
}

// chunk of windows of the bf16 path: no Z buffer; per window P0, P1 (bf16), their transposes, dP0, dP1 (float32) and the
// forward partials
HalfP make_half_bf16(const EpsP& p) {
  HalfP h = make_half(p, sizeof(float), true);
  const long long per_win = (h.A + h.Bn) * 8 + (h.NB / 64) * 4 + p.O * 2;
  const size_t budget = chunk_budget(p);
  long long wc = (long long)(budget / (size_t)per_win) / 64 * 64;
  if (wc < 64) wc = 64;
  const long long wn64 = (p.Wn + 63) / 64 * 64;
  h.wc = wc < wn64 ? wc : wn64;
  return h;
}

// k slices of the dCore product (128 x 128 tiles): one full round of the 3 workgroups a CU holds (768; 576 workgroups
// left a quarter of the slots empty: 388 -> 3xx us on the cfg3a layer-2 core), >= 256 windows per slice
int bf16_ksplit(const HalfP& h) {
  const long long tiles = ((h.A + BM - 1) / BM) * ((h.NB + BN - 1) / BN);
  long long ks = 768 / tiles;
  if (ks * tiles < 512) ks = (512 + tiles - 1) / tiles;
  const long long max_ks = (h.wc + 255) / 256;
  if (ks > max_ks) ks = max_ks;
  if (ks < 1) ks = 1;
  if (ks > 64) ks = 64;
  return (int)ks;
}

// bf16 path: every k-contiguous source is read 16 bytes at a time, scaled chunks and forward tiles lie inside one o
bool bf16_shape_ok(const EpsP& p) {
  if (!halves_shape_ok(p, sizeof(float))) return false;
  const HalfP h = make_half(p, sizeof(float), true);
  return h.A % 32 == 0 && h.Bn % 64 == 0;
}

size_t bf16_fwd_workspace(const EpsP& p) {
  const HalfP h = make_half_bf16(p);
  return align_up((size_t)h.wc * h.A * 2) + align_up((size_t)h.wc * h.Bn * 2) + align_up((size_t)h.A * h.NB * 2) +
         align_up((size_t)(h.NB / 64) * h.wc * 4) + 256;
}

// what a training forward keeps: both halves and the GEMM result, for all windows
struct SavedLayout { size_t p0, p1, z, total; };
SavedLayout saved_layout(const HalfP& h, size_t esz) {
  SavedLayout L;
  L.p0 = 0;
  L.p1 = align_up((size_t)h.p.Wn * h.A * esz);
  L.z = L.p1 + align_up((size_t)h.p.Wn * h.Bn * esz);
  L.total = L.z + align_up((size_t)h.p.Wn * h.NB * esz);
  return L;
}
constexpr size_t SAVED_MAX_BYTES = (size_t)16 << 30;   // beyond: nothing is kept, the backward recomputes

int bf16_fwd(const void* xv, const void* corev, void* outv, void* ws, const EpsP& p, hipStream_t st, void* saved) {
  const HalfP h = make_half_bf16(p);
  const SavedLayout SL = saved_layout(h, 2);
  const bf16_t* x = (const bf16_t*)xv;
  const bf16_t* core = (const bf16_t*)corev;
  bf16_t* out = (bf16_t*)outv;
  unsigned char* w8 = (unsigned char*)ws;
  auto take = [&](size_t bytes) {
    void* ptr = w8;
    w8 += align_up(bytes);
    return ptr;
  };
  bf16_t* P0 = (bf16_t*)take((size_t)h.wc * h.A * 2);
  bf16_t* P1 = (bf16_t*)take((size_t)h.wc * h.Bn * 2);
  bf16_t* coreP = (bf16_t*)take((size_t)h.A * h.NB * 2);
  float* part = (float*)take((size_t)(h.NB / 64) * h.wc * 4);
  // coreT[(o, i1)][i0]: the B operand of Z' with k = i0 contiguous
  hipLaunchKernelGGL(bf16_core_permute_k, dim3(blocks_for(h.A * h.NB, 256)), dim3(256), 0, st, core, coreP, h.A, h.Bn,
                     p.O, 2);
  DCTN_CHECK_LAUNCH();
  for (long long w0 = 0; w0 < p.Wn; w0 += h.wc) {
    const long long nw = p.Wn - w0 < h.wc ? p.Wn - w0 : h.wc;
    if (saved) {   // the halves of this chunk go where the backward will read them
      P0 = (bf16_t*)((unsigned char*)saved + SL.p0) + w0 * h.A;
      P1 = (bf16_t*)((unsigned char*)saved + SL.p1) + w0 * h.Bn;
    }
    int rc = launch_halves<float, bf16_t>(x, P0, P1, h, w0, nw, st);
    if (rc != DCTN_OK) return rc;
    GemmB g{(int)nw, (int)h.NB, (int)h.A, h.A, h.A, h.A, 0, nullptr, nullptr, 0, 32, p.O, P1, (int)h.Bn};
    g.zstore = saved ? (bf16_t*)((unsigned char*)saved + SL.z) + w0 * h.NB : nullptr;
    bf16_gemm_launch<BA_KFAST, BB_KFAST, BEPI_FWD>(P0, coreP, part, g, 1, st);
    DCTN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bf16_out_sum_k, dim3(blocks_for(nw * p.O, 256)), dim3(256), 0, st, (const float*)part,
                       out + w0 * p.O, nw, p.O, (int)(h.Bn / 64));
    DCTN_CHECK_LAUNCH();
  }
  return DCTN_OK;
}

size_t bf16_bwd_workspace(const EpsP& p, int need_dx, int need_dcore) {
  const HalfP h = make_half_bf16(p);
  size_t s = align_up((size_t)h.wc * h.A * 2) + align_up((size_t)h.wc * h.Bn * 2);
  if (need_dcore)
    s += align_up((size_t)bf16_ksplit(h) * h.A * h.NB * 4) + align_up((size_t)h.A * h.NB * 4) +
         align_up((size_t)(h.A + h.Bn + p.O) * h.wc * 2);
  if (need_dx)
    s += 2 * align_up((size_t)h.A * h.NB * 2) + align_up((size_t)h.wc * h.A * 4) + align_up((size_t)h.wc * h.Bn * 4) +
         align_up((size_t)p.N * p.Q * p.Wn * 4);
  return s + 256;
}

int bf16_bwd(const void* xv, const void* corev, const void* dYv, void* dXv, void* dCorev, void* ws, const EpsP& p,
             hipStream_t st, const void* saved) {
  const int need_dx = dXv != nullptr, need_dcore = dCorev != nullptr;
  const HalfP h = make_half_bf16(p);
  const SavedLayout SL = saved_layout(h, 2);
  const bf16_t* x = (const bf16_t*)xv;
  const bf16_t* core = (const bf16_t*)corev;
  const bf16_t* dY = (const bf16_t*)dYv;
  unsigned char* w8 = (unsigned char*)ws;
  auto take = [&](size_t bytes) {
    void* ptr = w8;
    w8 += align_up(bytes);
    return ptr;
  };
  bf16_t* P0 = (bf16_t*)take((size_t)h.wc * h.A * 2);
  bf16_t* P1 = (bf16_t*)take((size_t)h.wc * h.Bn * 2);
  float *part = nullptr, *dcore32 = nullptr, *dP0 = nullptr, *dP1 = nullptr, *gxw = nullptr;
  bf16_t *coreP = nullptr, *coreQ = nullptr, *halvesT = nullptr;
  if (need_dcore) {
    part = (float*)take((size_t)bf16_ksplit(h) * h.A * h.NB * 4);
    dcore32 = (float*)take((size_t)h.A * h.NB * 4);   // float32 running sum over the window chunks, (o, i1) order
    halvesT = (bf16_t*)take((size_t)(h.A + h.Bn + p.O) * h.wc * 2);   // P0^T | P1^T | dY^T, rows of wc windows
  }
  if (need_dx) {
    coreP = (bf16_t*)take((size_t)h.A * h.NB * 2);
    coreQ = (bf16_t*)take((size_t)h.A * h.NB * 2);
    dP0 = (float*)take((size_t)h.wc * h.A * 4);
    dP1 = (float*)take((size_t)h.wc * h.Bn * 4);
    gxw = (float*)take((size_t)p.N * p.Q * p.Wn * 4);
    hipLaunchKernelGGL(bf16_core_permute_k, dim3(blocks_for(h.A * h.NB, 256)), dim3(256), 0, st, core, coreP, h.A, h.Bn,
                       p.O, 0);
    DCTN_CHECK_LAUNCH();
    if (!saved) {
      hipLaunchKernelGGL(bf16_core_permute_k, dim3(blocks_for(h.A * h.NB, 256)), dim3(256), 0, st, core, coreQ, h.A, h.Bn,
                         p.O, 1);
      DCTN_CHECK_LAUNCH();
    }
  }
  int chunk = 0;
  for (long long w0 = 0; w0 < p.Wn; w0 += h.wc, ++chunk) {
    const long long nw = p.Wn - w0 < h.wc ? p.Wn - w0 : h.wc;
    const bf16_t* dyc = dY + w0 * p.O;
    int rc;
    if (saved) {   // the training forward left both halves (and Z')
      P0 = (bf16_t*)((unsigned char*)saved + SL.p0) + w0 * h.A;
      P1 = (bf16_t*)((unsigned char*)saved + SL.p1) + w0 * h.Bn;
    } else {
      rc = launch_halves<float, bf16_t>(x, P0, P1, h, w0, nw, st);
      if (rc != DCTN_OK) return rc;
    }
    if (need_dcore) {
      // dCoreP[i0][(o, i1)] = sum_w P0[w, i0] dY[w, o] P1[w, i1]: k = windows, split over grid.z.  Both operands are
      // read along the window index, from transposed copies of the halves and of dY (zero beyond nw)
      bf16_t* P0T = halvesT;
      bf16_t* P1T = P0T + h.A * h.wc;
      bf16_t* dYT = P1T + h.Bn * h.wc;
      const unsigned rb64 = (unsigned)((nw + 63) / 64);
      hipLaunchKernelGGL(bf16_transpose_k, dim3(rb64, (unsigned)((h.A + 63) / 64)), dim3(256), 0, st, (const bf16_t*)P0, P0T,
                         nw, (int)h.A, h.wc);
      hipLaunchKernelGGL(bf16_transpose_k, dim3(rb64, (unsigned)((h.Bn + 63) / 64)), dim3(256), 0, st, (const bf16_t*)P1,
                         P1T, nw, (int)h.Bn, h.wc);
      hipLaunchKernelGGL(bf16_transpose_k, dim3(rb64, 1), dim3(256), 0, st, dyc, dYT, nw, p.O, h.wc);
      DCTN_CHECK_LAUNCH();
      const long long K8 = (nw + 7) / 8 * 8;
      const int ksp = bf16_ksplit(h);
      const long long ksl = ((K8 + ksp - 1) / ksp + BK - 1) / BK * BK;
      const int slices = (int)((K8 + ksl - 1) / ksl);
      GemmB g{(int)h.A, (int)h.NB, (int)K8, h.wc, 0, ksl, h.A * h.NB, P1T, dYT, h.wc, (int)h.Bn, p.O, nullptr, (int)h.Bn};
      bf16_gemm_launch<BA_KFAST, BB_PROD>(P0T, nullptr, part, g, slices, st);
      DCTN_CHECK_LAUNCH();
      hipLaunchKernelGGL((halves_sum_partials_k<float, float>), dim3(blocks_for(h.A * h.NB, 256)), dim3(256), 0, st,
                         (const float*)part, dcore32, h.A * h.NB, slices, chunk > 0);
      DCTN_CHECK_LAUNCH();
    }
    if (need_dx) {
      // dP0[w, i0] = sum_(o, i1) (dY[w, o] P1[w, i1]) coreP[i0][(o, i1)]
      GemmB g0{(int)nw, (int)h.A, (int)h.NB, 0, h.NB, h.NB, 0, P1, dyc, h.Bn, (int)h.Bn, p.O, nullptr, (int)h.Bn};
      bf16_gemm_launch<BA_SCALED, BB_KFAST>(nullptr, coreP, dP0, g0, 1, st);
      DCTN_CHECK_LAUNCH();
      if (saved) {
        // dP1[w, i1] = sum_o dY[w, o] Z'[w, (o, i1)]: one pass over the Z' the forward kept instead of a third GEMM (a
        // kernel of its own with 8-byte loads: formed inside the leave-one-out kernel's loader - 2-byte loads - the layer-2
        // backward was slower, 1 231 against 1 111 us)
        const bf16_t* Zs = (const bf16_t*)((const unsigned char*)saved + SL.z) + w0 * h.NB;
        hipLaunchKernelGGL(bf16_dp1_saved_k, dim3(blocks_for(nw * (h.Bn / 4), 256)), dim3(256), 0, st, Zs, dyc, dP1, nw,
                           (int)h.Bn, p.O);
        DCTN_CHECK_LAUNCH();
      } else {
        // dP1[w, i1] = sum_(o, i0) (dY[w, o] P0[w, i0]) coreQ[i1][(o, i0)]
        const long long KQ = h.A * p.O;
        GemmB g1{(int)nw, (int)h.Bn, (int)KQ, 0, KQ, KQ, 0, P0, dyc, h.A, (int)h.A, p.O, nullptr, (int)h.Bn};
        bf16_gemm_launch<BA_SCALED, BB_KFAST>(nullptr, coreQ, dP1, g1, 1, st);
        DCTN_CHECK_LAUNCH();
      }
      rc = launch_dx_half<float, bf16_t>(x, dP0, gxw, h, 0, w0, nw, st);
      if (rc != DCTN_OK) return rc;
      rc = launch_dx_half<float, bf16_t>(x, dP1, gxw, h, 1, w0, nw, st);
      if (rc != DCTN_OK) return rc;
    }
  }
  if (need_dcore) {
    hipLaunchKernelGGL(bf16_dcore_unpermute_k, dim3(blocks_for(h.A * h.NB, 256)), dim3(256), 0, st, (const float*)dcore32,
                       (bf16_t*)dCorev, h.A, h.Bn, p.O);
    DCTN_CHECK_LAUNCH();
  }
  if (need_dx) {
    const int rc = eps_gather_dx_launch(gxw, dXv, p, DCTN_BF16, st);
    if (rc != DCTN_OK) return rc;
  }
  return DCTN_OK;
}

template <typename T>
size_t fwd_workspace_t(const EpsP& p) {
  const HalfP h = make_half(p, sizeof(T));
  return align_up((size_t)h.wc * h.A * sizeof(T)) + align_up((size_t)h.wc * h.Bn * sizeof(T)) +
         align_up((size_t)h.wc * h.NB * sizeof(T)) + 256;
}

template <typename T>
int fwd_t(const void* xv, const void* corev, void* outv, void* ws, const EpsP& p, hipStream_t st, void* saved) {
  const HalfP h = make_half(p, sizeof(T));
  const SavedLayout SL = saved_layout(h, sizeof(T));
  const T* x = (const T*)xv;
  const T* core = (const T*)corev;
  T* out = (T*)outv;
  unsigned char* w8 = (unsigned char*)ws;
  T* P0 = (T*)w8;
  T* P1 = (T*)(w8 + align_up((size_t)h.wc * h.A * sizeof(T)));
  T* Z = (T*)((unsigned char*)P1 + align_up((size_t)h.wc * h.Bn * sizeof(T)));
  for (long long w0 = 0; w0 < p.Wn; w0 += h.wc) {
    const long long nw = p.Wn - w0 < h.wc ? p.Wn - w0 : h.wc;
    T* Zs = nullptr;
    if (saved) {   // halves and Z of this chunk go where the backward will read them
      P0 = (T*)((unsigned char*)saved + SL.p0) + w0 * h.A;
      P1 = (T*)((unsigned char*)saved + SL.p1) + w0 * h.Bn;
      Zs = (T*)((unsigned char*)saved + SL.z) + w0 * h.NB;
    }
    int rc = launch_halves<T>(x, P0, P1, h, w0, nw, st);
    if (rc != DCTN_OK) return rc;
    GemmD<T> g{(int)nw, (int)h.NB, (int)h.A, h.A, h.NB, h.NB, h.A, 0, P1, nullptr, h.Bn, p.O};
    g.zstore = Zs;
    if (fused_epilogue_ok(p.O)) {
      // Z stays in the accumulators: per column tile partial sums [tile_n][w][o] (in the Z slot), then their sum
      const int tiles_n = (int)((h.NB + GT - 1) / GT);
      gemm_launch<A_KFAST, B_NFAST, EPI_FWD, T>(P0, core, Z, g, 1, st);
      DCTN_CHECK_LAUNCH();
      hipLaunchKernelGGL(halves_sum_partials_k<T>, dim3(blocks_for(nw * p.O, 256)), dim3(256), 0, st, (const T*)Z,
                         out + w0 * p.O, nw * p.O, tiles_n, 0);
      DCTN_CHECK_LAUNCH();
    } else {
      T* Zc = Zs ? Zs : Z;
      gemm_launch<A_KFAST, B_NFAST, EPI_STORE, T>(P0, core, Zc, g, 1, st);
      DCTN_CHECK_LAUNCH();
      hipLaunchKernelGGL(halves_fwd_contract_k<T>, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, st, (const T*)Zc,
                         (const T*)P1, out + w0 * p.O, nw, h.Bn, p.O);
      DCTN_CHECK_LAUNCH();
    }
  }
  return DCTN_OK;
}

template <typename T>
size_t bwd_workspace_t(const EpsP& p, int need_dx, int need_dcore) {
  const HalfP h = make_half(p, sizeof(T));
  const size_t e = sizeof(T);
  size_t s = align_up((size_t)h.wc * h.A * e) + align_up((size_t)h.wc * h.Bn * e);
  if (need_dcore) s += align_up((size_t)h.ksplit * h.A * h.NB * e);
  if (need_dx)
    s += align_up((size_t)h.wc * h.NB * e) + align_up((size_t)h.wc * h.A * e) + align_up((size_t)h.wc * h.Bn * e) +
         align_up((size_t)p.N * p.Q * p.Wn * e);
  return s + 256;
}

template <typename T>
int bwd_t(const void* xv, const void* corev, const void* dYv, void* dXv, void* dCorev, void* ws, const EpsP& p,
          int dtype, hipStream_t st, const void* saved) {
  const int need_dx = dXv != nullptr, need_dcore = dCorev != nullptr;
  const HalfP h = make_half(p, sizeof(T));
  const SavedLayout SL = saved_layout(h, sizeof(T));
  const size_t e = sizeof(T);
  const T* x = (const T*)xv;
  const T* core = (const T*)corev;
  const T* dY = (const T*)dYv;
  unsigned char* w8 = (unsigned char*)ws;
  auto take = [&](size_t bytes) {
    T* ptr = (T*)w8;
    w8 += align_up(bytes);
    return ptr;
  };
  T* P0 = take((size_t)h.wc * h.A * e);
  T* P1 = take((size_t)h.wc * h.Bn * e);
  T* part = need_dcore ? take((size_t)h.ksplit * h.A * h.NB * e) : nullptr;
  T *Z = nullptr, *dP0 = nullptr, *dP1 = nullptr, *gxw = nullptr;
  if (need_dx) {
    Z = take((size_t)h.wc * h.NB * e);
    dP0 = take((size_t)h.wc * h.A * e);
    dP1 = take((size_t)h.wc * h.Bn * e);
    gxw = take((size_t)p.N * p.Q * p.Wn * e);
  }
  int chunk = 0;
  for (long long w0 = 0; w0 < p.Wn; w0 += h.wc, ++chunk) {
    const long long nw = p.Wn - w0 < h.wc ? p.Wn - w0 : h.wc;
    const T* dyc = dY + w0 * p.O;
    int rc;
    const T* Zs = nullptr;
    if (saved) {   // the training forward left both halves and Z
      P0 = (T*)((unsigned char*)saved + SL.p0) + w0 * h.A;
      P1 = (T*)((unsigned char*)saved + SL.p1) + w0 * h.Bn;
      Zs = (const T*)((const unsigned char*)saved + SL.z) + w0 * h.NB;
    } else {
      rc = launch_halves<T>(x, P0, P1, h, w0, nw, st);
      if (rc != DCTN_OK) return rc;
    }
    if (need_dcore) {
      // dCore[(i0), (i1 o)] = sum_w P0[w, i0] T[w, (i1 o)]: K = windows, split over grid.z
      const long long ksl = ((nw + h.ksplit - 1) / h.ksplit + GK - 1) / GK * GK;
      const int slices = (int)((nw + ksl - 1) / ksl);
      GemmD<T> g{(int)h.A, (int)h.NB, (int)nw, h.A, 0, h.NB, ksl, h.A * h.NB, P1, dyc, h.Bn, p.O};
      gemm_launch<A_MFAST, B_T, EPI_STORE, T>(P0, nullptr, part, g, slices, st);
      DCTN_CHECK_LAUNCH();
      hipLaunchKernelGGL(halves_sum_partials_k<T>, dim3(blocks_for(h.A * h.NB, 256)), dim3(256), 0, st, (const T*)part,
                         (T*)dCorev, h.A * h.NB, slices, chunk > 0);
      DCTN_CHECK_LAUNCH();
    }
    if (need_dx) {
      // dP0[w, i0] = sum_(i1 o) T[w, (i1 o)] Core[i0, (i1 o)]
      GemmD<T> g0{(int)nw, (int)h.A, (int)h.NB, 0, h.NB, h.A, h.NB, 0, P1, dyc, h.Bn, p.O};
      gemm_launch<A_T, B_KFAST, EPI_STORE, T>(nullptr, core, dP0, g0, 1, st);
      DCTN_CHECK_LAUNCH();
      // Z[w, (i1 o)] = sum_i0 P0[w, i0] Core[i0, (i1 o)], dP1[w, i1] = sum_o dY[w, o] Z[w, i1, o]
      GemmD<T> g1{(int)nw, (int)h.NB, (int)h.A, h.A, h.NB, h.NB, h.A, 0, nullptr, dyc, h.Bn, p.O};
      if (Zs) {
        // the forward kept Z (the reference's autograd keeps the result of path step (0,1) too: dctn/eps.py:25-30): dP1 is
        // formed from it inside the leave-one-out kernel's loader below - no GEMM, no dP1 round trip through memory
      } else if (fused_epilogue_ok(p.O)) {
        gemm_launch<A_KFAST, B_NFAST, EPI_DP1, T>(P0, core, dP1, g1, 1, st);
        DCTN_CHECK_LAUNCH();
      } else {
        gemm_launch<A_KFAST, B_NFAST, EPI_STORE, T>(P0, core, Z, g1, 1, st);
        DCTN_CHECK_LAUNCH();
        hipLaunchKernelGGL(halves_dp1_k<T>, dim3(blocks_for(nw * h.Bn, 256)), dim3(256), 0, st, (const T*)Z, dyc, dP1,
                           nw, h.Bn, p.O);
        DCTN_CHECK_LAUNCH();
      }
      rc = launch_dx_half<T>(x, dP0, gxw, h, 0, w0, nw, st);
      if (rc != DCTN_OK) return rc;
      rc = launch_dx_half<T>(x, dP1, gxw, h, 1, w0, nw, st, Zs ? DxSavedZ{Zs, dyc, p.O, 1} : DxSavedZ{nullptr, nullptr, 0, 0});
      if (rc != DCTN_OK) return rc;
    }
  }
  if (need_dx) {
    const int rc = eps_gather_dx_launch(gxw, dXv, p, dtype, st);
    if (rc != DCTN_OK) return rc;
  }
  return DCTN_OK;
}

}  // namespace

// float64 (the dtype of the reference's tests and of BASELINE cfg1) and float32 shapes the other families leave
// (odd Q, ...; the dispatcher asks the bf16-register and bigcore families first): at least two factors, a core
// worth a GEMM, halves that fit the per-wave LDS tables.
bool eps_halves_wanted(const EpsP& p, int dtype) {
  if (dtype == DCTN_F64) return halves_shape_ok(p, sizeof(double));
  if (dtype == DCTN_F32) return halves_shape_ok(p, sizeof(float));
  if (dtype == DCTN_BF16) return bf16_shape_ok(p);   // bf16 MFMA; the dispatcher asks the bf16 register family first
  return false;
}

size_t eps_fwd_halves_workspace(const EpsP& p, int dtype) {
  if (!eps_halves_wanted(p, dtype)) return 0;
  if (dtype == DCTN_BF16) return bf16_fwd_workspace(p);
  return dtype == DCTN_F64 ? fwd_workspace_t<double>(p) : fwd_workspace_t<float>(p);
}

size_t eps_halves_saved_bytes(const EpsP& p, int dtype) {
  if (!eps_halves_wanted(p, dtype)) return 0;
  const size_t total = dtype == DCTN_BF16 ? saved_layout(make_half_bf16(p), 2).total
                       : dtype == DCTN_F64 ? saved_layout(make_half(p, 8), 8).total
                                           : saved_layout(make_half(p, 4), 4).total;
  return total <= SAVED_MAX_BYTES ? total : 0;
}

int eps_fwd_halves(const void* x, const void* core, void* out, void* ws, size_t ws_bytes, const EpsP& p, int dtype,
                   hipStream_t st, void* saved) {
  if (!eps_halves_wanted(p, dtype)) return DCTN_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < eps_fwd_halves_workspace(p, dtype)) return DCTN_ERR_WORKSPACE;
  if (saved && (uintptr_t)saved % 256) return DCTN_ERR_BAD_SHAPE;
  if (dtype == DCTN_BF16) {
    if ((uintptr_t)core % 16) return DCTN_ERR_UNSUPPORTED;   // 16-byte vector loads of core rows
    const int rc = bf16_fwd(x, core, out, ws, p, st, saved);
    if (rc == DCTN_OK) dctn_set_last_kernel(saved ? "eps_fwd_mfma_bf16_halves_saving" : "eps_fwd_mfma_bf16_halves");
    return rc;
  }
  const int rc = dtype == DCTN_F64 ? fwd_t<double>(x, core, out, ws, p, st, saved) : fwd_t<float>(x, core, out, ws, p, st, saved);
  if (rc == DCTN_OK)
    dctn_set_last_kernel(dtype == DCTN_F64 ? (saved ? "eps_fwd_mfma_f64_halves_saving" : "eps_fwd_mfma_f64_halves")
                                           : (saved ? "eps_fwd_mfma_f32_halves_saving" : "eps_fwd_mfma_f32_halves"));
  return rc;
}

size_t eps_bwd_halves_workspace(const EpsP& p, int dtype, int need_dx, int need_dcore) {
  if (!eps_halves_wanted(p, dtype)) return 0;
  if (dtype == DCTN_BF16) return bf16_bwd_workspace(p, need_dx, need_dcore);
  return dtype == DCTN_F64 ? bwd_workspace_t<double>(p, need_dx, need_dcore) : bwd_workspace_t<float>(p, need_dx, need_dcore);
}

int eps_bwd_halves(const void* x, const void* core, const void* dY, void* dX, void* dCore, void* ws, size_t ws_bytes,
                   const EpsP& p, int dtype, hipStream_t st, const void* saved, size_t saved_bytes) {
  if (!eps_halves_wanted(p, dtype)) return DCTN_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < eps_bwd_halves_workspace(p, dtype, dX != nullptr, dCore != nullptr)) return DCTN_ERR_WORKSPACE;
  if (saved) {   // only what the training forward of this very shape can have written
    const size_t need = eps_halves_saved_bytes(p, dtype);
    if (need == 0 || saved_bytes < need || (uintptr_t)saved % 256) saved = nullptr;
  }
  if (dtype == DCTN_BF16) {
    if ((uintptr_t)core % 16) return DCTN_ERR_UNSUPPORTED;
    const int rc = bf16_bwd(x, core, dY, dX, dCore, ws, p, st, saved);
    if (rc == DCTN_OK) dctn_set_last_kernel(saved ? "eps_bwd_mfma_bf16_halves_savedz" : "eps_bwd_mfma_bf16_halves");
    return rc;
  }
  const int rc = dtype == DCTN_F64 ? bwd_t<double>(x, core, dY, dX, dCore, ws, p, dtype, st, saved)
                                   : bwd_t<float>(x, core, dY, dX, dCore, ws, p, dtype, st, saved);
  if (rc == DCTN_OK)
    dctn_set_last_kernel(dtype == DCTN_F64 ? (saved ? "eps_bwd_mfma_f64_halves_savedz" : "eps_bwd_mfma_f64_halves")
                                           : (saved ? "eps_bwd_mfma_f32_halves_savedz" : "eps_bwd_mfma_f32_halves"));
  return rc;
}
