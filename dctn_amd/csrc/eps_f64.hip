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

typedef __attribute__((ext_vector_type(4))) double f64x4;

namespace {

constexpr int GT = 64, GK = 16;                 // tile, k chunk
constexpr int AP = GK + 1, BP = GT + 1;         // padded LDS rows
constexpr size_t CHUNK_BYTES = (size_t)1 << 30; // bound of the per-chunk buffers

struct HalfP {
  EpsP p;
  int n0, n1;
  long long A, Bn, NB;    // Q^n0, Q^n1, Bn * O
  long long wc;           // windows per chunk (multiple of 64)
  int ksplit;             // grid.z of the dCore product
};

HalfP make_half(const EpsP& p) {
  HalfP h;
  h.p = p;
  h.n0 = p.N / 2;
  h.n1 = p.N - h.n0;
  h.A = ipow_ll(p.Q, h.n0);
  h.Bn = ipow_ll(p.Q, h.n1);
  h.NB = h.Bn * p.O;
  const long long per_win = (2 * h.A + 2 * h.Bn + h.NB) * (long long)sizeof(double);
  long long wc = (long long)(CHUNK_BYTES / (size_t)per_win);
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

// ------------------------------------------------------------------ P0 / P1 of a chunk of windows
template <int LOGQ>
__global__ __launch_bounds__(256) void f64_halves_k(const double* __restrict__ x, double* __restrict__ P0,
                                                    double* __restrict__ P1, HalfP h, long long w0, long long nw) {
  extern __shared__ double xs[];   // [N][Q]
  const EpsP& p = h.p;
  const int hw = p.Ho * p.Wo;
  for (long long wl = blockIdx.x; wl < nw; wl += gridDim.x) {
    const long long w = w0 + wl;
    const long long b = w / hw;
    const int rem = (int)(w - b * hw), ho = rem / p.Wo, wo = rem - ho * p.Wo;
    __syncthreads();
    for (int e = threadIdx.x; e < p.N * p.Q; e += 256) {
      const int n = e / p.Q, q = e - n * p.Q;
      const int pos = n / p.C, ch = n - pos * p.C, dh = pos / p.K, dw = pos - dh * p.K;
      xs[e] = x[ch * p.s[0] + b * p.s[1] + (long long)(ho + dh) * p.s[2] + (long long)(wo + dw) * p.s[3] + q * p.s[4]];
    }
    __syncthreads();
    for (long long e = threadIdx.x; e < h.A + h.Bn; e += 256) {
      const bool second = e >= h.A;
      long long t = second ? e - h.A : e;
      const int base = second ? h.n0 : 0, nd = second ? h.n1 : h.n0;
      double pr = 1.0;
      for (int d = nd - 1; d >= 0; --d) {
        int digit;
        if (LOGQ > 0) {
          digit = (int)(t & ((1 << LOGQ) - 1));
          t >>= LOGQ;
        } else {
          digit = (int)(t % p.Q);
          t /= p.Q;
        }
        pr *= xs[(base + d) * p.Q + digit];
      }
      if (second)
        P1[wl * h.Bn + (e - h.A)] = pr;
      else
        P0[wl * h.A + e] = pr;
    }
  }
}

// ------------------------------------------------------------------------------- the GEMM kernel
enum { A_KFAST = 0, A_MFAST = 1, A_T = 2 };   // A[m][k] k-contiguous / stored [k][m] / T[w, (i1 o)] formed from P1, dY
enum { B_NFAST = 0, B_KFAST = 1, B_T = 2 };   // B[k][n] n-contiguous / stored [n][k] / T (k = window)

struct GemmD {
  int M, N, K;
  long long lda, ldb, ldc;
  long long kslice;       // k range per grid.z slice
  long long cslice;       // elements between the C of two slices
  const double* p1;       // T operand: P1 (ld Bn) and dY (ld O) of the chunk
  const double* dy;
  long long Bn;
  int O;
};

template <int LA, int LB>
__global__ __launch_bounds__(256) void f64_gemm_k(const double* __restrict__ Ag, const double* __restrict__ Bg,
                                                  double* __restrict__ Cg, GemmD g) {
  __shared__ double As[GT * AP];
  __shared__ double Bs[GK * BP];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const long long kbeg = (long long)blockIdx.z * g.kslice;
  const long long kend = kbeg + g.kslice < g.K ? kbeg + g.kslice : g.K;
  constexpr bool AKF = LA != A_MFAST, BKF = LB == B_KFAST;
  // element e = tid + 256 u of a 64 x 16 operand tile: k fastest: k = e & 15, x = e >> 4 (+16 u); x fastest: x = e & 63, k = e >> 6 (+4 u)
  const int kf_k = tid & 15, kf_x = tid >> 4, xf_x = tid & 63, xf_k = tid >> 6;
  double ra[4], rb[4], ra2[4], rb2[4];
  // T operand on the B side: this thread's column n = (i1, o) is fixed
  int bt_i1 = 0, bt_o = 0;
  if (LB == B_T) {
    const int n = min(n0 + xf_x, g.N - 1);
    bt_i1 = n / g.O;
    bt_o = n - bt_i1 * g.O;
  }
  auto fetch = [&](long long k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (LA == A_KFAST) {
        const long long m = min(m0 + kf_x + 16 * u, g.M - 1), k = min(k0 + kf_k, (long long)g.K - 1);
        ra[u] = Ag[m * g.lda + k];
      } else if (LA == A_MFAST) {
        const long long m = min(m0 + xf_x, g.M - 1), k = min(k0 + xf_k + 4 * u, (long long)g.K - 1);
        ra[u] = Ag[k * g.lda + m];
      } else {   // T[m = w][k = (i1, o)]
        const long long m = min(m0 + kf_x + 16 * u, g.M - 1);
        const int k = (int)min(k0 + kf_k, (long long)g.K - 1);
        const int i1 = k / g.O, o = k - i1 * g.O;
        ra[u] = g.p1[m * g.Bn + i1];
        ra2[u] = g.dy[m * g.O + o];
      }
      if (LB == B_NFAST) {
        const long long k = min(k0 + xf_k + 4 * u, (long long)g.K - 1), n = min(n0 + xf_x, g.N - 1);
        rb[u] = Bg[k * g.ldb + n];
      } else if (LB == B_KFAST) {
        const long long n = min(n0 + kf_x + 16 * u, g.N - 1), k = min(k0 + kf_k, (long long)g.K - 1);
        rb[u] = Bg[n * g.ldb + k];
      } else {   // T[k = w][n = (i1, o)]
        const long long k = min(k0 + xf_k + 4 * u, (long long)g.K - 1);
        rb[u] = g.p1[k * g.Bn + bt_i1];
        rb2[u] = g.dy[k * g.O + bt_o];
      }
    }
  };
  auto stage = [&](long long k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      double va = LA == A_T ? ra[u] * ra2[u] : ra[u];
      double vb = LB == B_T ? rb[u] * rb2[u] : rb[u];
      if (AKF) {
        const int m = kf_x + 16 * u;
        if (!((m0 + m < g.M) & (k0 + kf_k < kend))) va = 0.0;
        As[m * AP + kf_k] = va;
      } else {
        const int k = xf_k + 4 * u;
        if (!((m0 + xf_x < g.M) & (k0 + k < kend))) va = 0.0;
        As[xf_x * AP + k] = va;
      }
      if (BKF) {
        const int n = kf_x + 16 * u;
        if (!((n0 + n < g.N) & (k0 + kf_k < kend))) vb = 0.0;
        Bs[kf_k * BP + n] = vb;
      } else {
        const int k = xf_k + 4 * u;
        if (!((n0 + xf_x < g.N) & (k0 + k < kend))) vb = 0.0;
        Bs[k * BP + xf_x] = vb;
      }
    }
  };
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  if (kbeg < kend) fetch(kbeg);
  for (long long k0 = kbeg; k0 < kend; k0 += GK) {
    __syncthreads();
    stage(k0);
    __syncthreads();
    if (k0 + GK < kend) fetch(k0 + GK);
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[(32 * wm + 16 * i + lr) * AP + 4 * kk + lk];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[(4 * kk + lk) * BP + 32 * wn + 16 * j + lr];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  double* C = Cg + (long long)blockIdx.z * g.cslice;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int m = m0 + 32 * wm + 16 * i + 4 * v + lk, n = n0 + 32 * wn + 16 * j + lr;
        if (m < g.M && n < g.N) C[(long long)m * g.ldc + n] = acc[i][j][v];
      }
}

template <int LA, int LB>
void gemm_launch(const double* A, const double* B, double* C, const GemmD& g, int slices, hipStream_t st) {
  const dim3 grid((g.N + GT - 1) / GT, (g.M + GT - 1) / GT, slices);
  hipLaunchKernelGGL((f64_gemm_k<LA, LB>), grid, dim3(256), 0, st, A, B, C, g);
}

// --------------------------------------------------------------------- contractions around the GEMMs
// out[w, o] = sum_i1 Z[w, i1, o] P1[w, i1]: one wave per window
__global__ __launch_bounds__(256) void f64_fwd_contract_k(const double* __restrict__ Z, const double* __restrict__ P1,
                                                          double* __restrict__ out, long long nw, long long Bn, int O) {
  const int lane = threadIdx.x & 63;
  const long long wl = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wl >= nw) return;
  const double* z = Z + wl * Bn * O;
  const double* p1 = P1 + wl * Bn;
  for (int o = 0; o < O; ++o) {
    double s = 0.0;
    for (long long i1 = lane; i1 < Bn; i1 += 64) s += z[i1 * O + o] * p1[i1];
    s = wave_reduce_sum(s);
    if (lane == 0) out[wl * O + o] = s;
  }
}

// dP1[w, i1] = sum_o dY[w, o] Z[w, i1, o]
__global__ __launch_bounds__(256) void f64_dp1_k(const double* __restrict__ Z, const double* __restrict__ dY,
                                                 double* __restrict__ dP1, long long nw, long long Bn, int O) {
  const long long total = nw * Bn;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long wl = idx / Bn;
    const double* z = Z + idx * O;
    const double* dy = dY + wl * O;
    double s = 0.0;
    for (int o = 0; o < O; ++o) s += dy[o] * z[o];
    dP1[idx] = s;
  }
}

// gxw[(n Q + q)][w] = sum over the entries i of the factor's half whose digit of factor n is q of
// dP[w, i] * prod_{other factors d of the half} x_d[w, digit_d(i)].  One workgroup per window; thread
// t owns the pair (factor, q) = t % (nd Q) and the slice t / (nd Q) of the half's entries.
template <int LOGQ>
__global__ __launch_bounds__(256) void f64_dx_half_k(const double* __restrict__ x, const double* __restrict__ dP,
                                                     double* __restrict__ gxw, HalfP h, int second, long long w0,
                                                     long long nw) {
  extern __shared__ double sm[];
  const EpsP& p = h.p;
  const int base = second ? h.n0 : 0, nd = second ? h.n1 : h.n0;
  const long long E = second ? h.Bn : h.A;
  double* xs = sm;                    // [nd][Q]
  double* dps = xs + nd * p.Q;        // [E]
  double* red = dps + E;              // [256]
  const int np = nd * p.Q;            // (factor, q) pairs
  const int nsl = 256 / np;           // slices
  const int pr = threadIdx.x % np, sl = threadIdx.x / np;
  const int fd = pr / p.Q, fq = pr - fd * p.Q;
  const long long EQ = E / p.Q;       // entries with a given digit at factor fd
  // stride of factor fd's digit inside the half's index (factor 0 of the half is the most significant)
  long long stride = 1;
  for (int d = nd - 1; d > fd; --d) stride *= p.Q;
  const int hw = p.Ho * p.Wo;
  for (long long wl = blockIdx.x; wl < nw; wl += gridDim.x) {
    const long long w = w0 + wl;
    const long long b = w / hw;
    const int rem = (int)(w - b * hw), ho = rem / p.Wo, wo = rem - ho * p.Wo;
    __syncthreads();
    for (int e = threadIdx.x; e < nd * p.Q; e += 256) {
      const int n = base + e / p.Q, q = e % p.Q;
      const int pos = n / p.C, ch = n - pos * p.C, dh = pos / p.K, dw = pos - dh * p.K;
      xs[e] = x[ch * p.s[0] + b * p.s[1] + (long long)(ho + dh) * p.s[2] + (long long)(wo + dw) * p.s[3] + q * p.s[4]];
    }
    for (long long e = threadIdx.x; e < E; e += 256) dps[e] = dP[wl * E + e];
    __syncthreads();
    double acc = 0.0;
    if (sl < nsl) {
      for (long long j = sl; j < EQ; j += nsl) {
        // i = j with the digit fq inserted at factor fd: i = (j / stride) * stride * Q + fq * stride + j % stride
        const long long jh = j / stride, jl = j - jh * stride;
        const long long i = (jh * p.Q + fq) * stride + jl;
        long long t = i;
        double prd = 1.0;
        for (int d = nd - 1; d >= 0; --d) {
          int digit;
          if (LOGQ > 0) {
            digit = (int)(t & ((1 << LOGQ) - 1));
            t >>= LOGQ;
          } else {
            digit = (int)(t % p.Q);
            t /= p.Q;
          }
          if (d != fd) prd *= xs[d * p.Q + digit];
        }
        acc += dps[i] * prd;
      }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < np) {
      double s = 0.0;
      for (int k = 0; k < nsl; ++k) s += red[k * np + threadIdx.x];
      gxw[(long long)((base + fd) * p.Q + fq) * p.Wn + w] = s;
    }
  }
}

// dCore (+)= sum of the split-k partial products
__global__ __launch_bounds__(256) void f64_sum_partials_k(const double* __restrict__ part, double* __restrict__ dCore,
                                                          long long n, int slices, int accumulate) {
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
    double s = accumulate ? dCore[idx] : 0.0;
    for (int k = 0; k < slices; ++k) s += part[(long long)k * n + idx];
    dCore[idx] = s;
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

int launch_halves(const double* x, double* P0, double* P1, const HalfP& h, long long w0, long long nw, hipStream_t st) {
  const size_t lds = (size_t)h.p.N * h.p.Q * sizeof(double);
  const unsigned grid = blocks_for(nw, 1);
  switch (ilog2_pow2(h.p.Q)) {
    case 1: hipLaunchKernelGGL(f64_halves_k<1>, dim3(grid), dim3(256), lds, st, x, P0, P1, h, w0, nw); break;
    case 2: hipLaunchKernelGGL(f64_halves_k<2>, dim3(grid), dim3(256), lds, st, x, P0, P1, h, w0, nw); break;
    case 3: hipLaunchKernelGGL(f64_halves_k<3>, dim3(grid), dim3(256), lds, st, x, P0, P1, h, w0, nw); break;
    default: hipLaunchKernelGGL(f64_halves_k<0>, dim3(grid), dim3(256), lds, st, x, P0, P1, h, w0, nw); break;
  }
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

int launch_dx_half(const double* x, const double* dP, double* gxw, const HalfP& h, int second, long long w0,
                   long long nw, hipStream_t st) {
  const int nd = second ? h.n1 : h.n0;
  const long long E = second ? h.Bn : h.A;
  const size_t lds = ((size_t)nd * h.p.Q + (size_t)E + 256) * sizeof(double);
  const unsigned grid = blocks_for(nw, 1);
  switch (ilog2_pow2(h.p.Q)) {
    case 1: hipLaunchKernelGGL(f64_dx_half_k<1>, dim3(grid), dim3(256), lds, st, x, dP, gxw, h, second, w0, nw); break;
    case 2: hipLaunchKernelGGL(f64_dx_half_k<2>, dim3(grid), dim3(256), lds, st, x, dP, gxw, h, second, w0, nw); break;
    case 3: hipLaunchKernelGGL(f64_dx_half_k<3>, dim3(grid), dim3(256), lds, st, x, dP, gxw, h, second, w0, nw); break;
    default: hipLaunchKernelGGL(f64_dx_half_k<0>, dim3(grid), dim3(256), lds, st, x, dP, gxw, h, second, w0, nw); break;
  }
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// float64, at least two factors, a core worth a GEMM, halves that fit the per-window LDS of the dX kernel
bool eps_f64_wanted(const EpsP& p, int dtype) {
  if (dtype != DCTN_F64 || p.N < 2) return false;
  if (p.R * p.O < 2048 || p.Wn < 64) return false;
  const HalfP h = make_half(p);
  if (h.Bn > 4096 || h.A > 4096 || h.NB > (1ll << 24)) return false;   // dps[] in LDS; int-sized GEMM dims
  if (h.n1 * p.Q > 256 || p.Wn >= (1ll << 31)) return false;
  return true;
}

size_t eps_fwd_f64_workspace(const EpsP& p, int dtype) {
  if (!eps_f64_wanted(p, dtype)) return 0;
  const HalfP h = make_half(p);
  return align_up((size_t)h.wc * h.A * 8) + align_up((size_t)h.wc * h.Bn * 8) + align_up((size_t)h.wc * h.NB * 8) + 256;
}

int eps_fwd_f64(const void* xv, const void* corev, void* outv, void* ws, size_t ws_bytes, const EpsP& p, int dtype,
                hipStream_t st) {
  if (!eps_f64_wanted(p, dtype)) return DCTN_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < eps_fwd_f64_workspace(p, dtype)) return DCTN_ERR_WORKSPACE;
  const HalfP h = make_half(p);
  const double* x = (const double*)xv;
  const double* core = (const double*)corev;
  double* out = (double*)outv;
  unsigned char* w8 = (unsigned char*)ws;
  double* P0 = (double*)w8;
  double* P1 = (double*)(w8 + align_up((size_t)h.wc * h.A * 8));
  double* Z = (double*)((unsigned char*)P1 + align_up((size_t)h.wc * h.Bn * 8));
  for (long long w0 = 0; w0 < p.Wn; w0 += h.wc) {
    const long long nw = p.Wn - w0 < h.wc ? p.Wn - w0 : h.wc;
    int rc = launch_halves(x, P0, P1, h, w0, nw, st);
    if (rc != DCTN_OK) return rc;
    GemmD g{(int)nw, (int)h.NB, (int)h.A, h.A, h.NB, h.NB, h.A, 0, nullptr, nullptr, h.Bn, p.O};
    gemm_launch<A_KFAST, B_NFAST>(P0, core, Z, g, 1, st);
    DCTN_CHECK_LAUNCH();
    hipLaunchKernelGGL(f64_fwd_contract_k, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, st, Z, P1, out + w0 * p.O, nw,
                       h.Bn, p.O);
    DCTN_CHECK_LAUNCH();
  }
  dctn_set_last_kernel("eps_fwd_mfma_f64_halves");
  return DCTN_OK;
}

size_t eps_bwd_f64_workspace(const EpsP& p, int dtype, int need_dx, int need_dcore) {
  if (!eps_f64_wanted(p, dtype)) return 0;
  const HalfP h = make_half(p);
  size_t s = align_up((size_t)h.wc * h.A * 8) + align_up((size_t)h.wc * h.Bn * 8);
  if (need_dcore) s += align_up((size_t)h.ksplit * h.A * h.NB * 8);
  if (need_dx)
    s += align_up((size_t)h.wc * h.NB * 8) + align_up((size_t)h.wc * h.A * 8) + align_up((size_t)h.wc * h.Bn * 8) +
         align_up((size_t)p.N * p.Q * p.Wn * 8);
  return s + 256;
}

int eps_bwd_f64(const void* xv, const void* corev, const void* dYv, void* dXv, void* dCorev, void* ws, size_t ws_bytes,
                const EpsP& p, int dtype, hipStream_t st) {
  if (!eps_f64_wanted(p, dtype)) return DCTN_ERR_UNSUPPORTED;
  const int need_dx = dXv != nullptr, need_dcore = dCorev != nullptr;
  if (!ws || ws_bytes < eps_bwd_f64_workspace(p, dtype, need_dx, need_dcore)) return DCTN_ERR_WORKSPACE;
  const HalfP h = make_half(p);
  const double* x = (const double*)xv;
  const double* core = (const double*)corev;
  const double* dY = (const double*)dYv;
  unsigned char* w8 = (unsigned char*)ws;
  auto take = [&](size_t bytes) {
    double* ptr = (double*)w8;
    w8 += align_up(bytes);
    return ptr;
  };
  double* P0 = take((size_t)h.wc * h.A * 8);
  double* P1 = take((size_t)h.wc * h.Bn * 8);
  double* part = need_dcore ? take((size_t)h.ksplit * h.A * h.NB * 8) : nullptr;
  double *Z = nullptr, *dP0 = nullptr, *dP1 = nullptr, *gxw = nullptr;
  if (need_dx) {
    Z = take((size_t)h.wc * h.NB * 8);
    dP0 = take((size_t)h.wc * h.A * 8);
    dP1 = take((size_t)h.wc * h.Bn * 8);
    gxw = take((size_t)p.N * p.Q * p.Wn * 8);
  }
  int chunk = 0;
  for (long long w0 = 0; w0 < p.Wn; w0 += h.wc, ++chunk) {
    const long long nw = p.Wn - w0 < h.wc ? p.Wn - w0 : h.wc;
    const double* dyc = dY + w0 * p.O;
    int rc = launch_halves(x, P0, P1, h, w0, nw, st);
    if (rc != DCTN_OK) return rc;
    if (need_dcore) {
      // dCore[(i0), (i1 o)] = sum_w P0[w, i0] T[w, (i1 o)]: K = windows, split over grid.z
      const long long ksl = ((nw + h.ksplit - 1) / h.ksplit + GK - 1) / GK * GK;
      const int slices = (int)((nw + ksl - 1) / ksl);
      GemmD g{(int)h.A, (int)h.NB, (int)nw, h.A, 0, h.NB, ksl, h.A * h.NB, P1, dyc, h.Bn, p.O};
      gemm_launch<A_MFAST, B_T>(P0, nullptr, part, g, slices, st);
      DCTN_CHECK_LAUNCH();
      hipLaunchKernelGGL(f64_sum_partials_k, dim3(blocks_for(h.A * h.NB, 256)), dim3(256), 0, st, part, (double*)dCorev,
                         h.A * h.NB, slices, chunk > 0);
      DCTN_CHECK_LAUNCH();
    }
    if (need_dx) {
      // dP0[w, i0] = sum_(i1 o) T[w, (i1 o)] Core[i0, (i1 o)]
      GemmD g0{(int)nw, (int)h.A, (int)h.NB, 0, h.NB, h.A, h.NB, 0, P1, dyc, h.Bn, p.O};
      gemm_launch<A_T, B_KFAST>(nullptr, core, dP0, g0, 1, st);
      DCTN_CHECK_LAUNCH();
      // Z[w, (i1 o)] = sum_i0 P0[w, i0] Core[i0, (i1 o)], dP1[w, i1] = sum_o dY[w, o] Z[w, i1, o]
      GemmD g1{(int)nw, (int)h.NB, (int)h.A, h.A, h.NB, h.NB, h.A, 0, nullptr, nullptr, h.Bn, p.O};
      gemm_launch<A_KFAST, B_NFAST>(P0, core, Z, g1, 1, st);
      DCTN_CHECK_LAUNCH();
      hipLaunchKernelGGL(f64_dp1_k, dim3(blocks_for(nw * h.Bn, 256)), dim3(256), 0, st, Z, dyc, dP1, nw, h.Bn, p.O);
      DCTN_CHECK_LAUNCH();
      rc = launch_dx_half(x, dP0, gxw, h, 0, w0, nw, st);
      if (rc != DCTN_OK) return rc;
      rc = launch_dx_half(x, dP1, gxw, h, 1, w0, nw, st);
      if (rc != DCTN_OK) return rc;
    }
  }
  if (need_dx) {
    const int rc = eps_gather_dx_launch(gxw, dXv, p, DCTN_F64, st);
    if (rc != DCTN_OK) return rc;
  }
  dctn_set_last_kernel("eps_bwd_mfma_f64_halves");
  return DCTN_OK;
}
