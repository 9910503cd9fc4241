// Generic EPS kernels: any (C, K, Q, O), float32 / float64 / bf16 storage (f32 accumulate).
//
// Replaces dctn/eps.py:19-40 (forward) and torch autograd through it (backward).
// One lane owns one window.  The core index i = (i_0 .. i_{N-1}) (row-major, factor
// n = pos*C + ch) is split into a HIGH part (first NH factors) and a LOW part (last m factors,
// LO = Q^m <= 64): P[w,i] = Phi[w,hi] * Plo[w,lo].  Plo lives in a per-lane LDS column
// ([entry][lane] layout: conflict-free), the core is read through wave-uniform (scalar) loads.
// Nothing of size Wn x Q^n is ever written to HBM.
//
// These kernels are the exact reference-grade path (f64 for the reference's own tests, odd Q);
// the MFMA kernels in eps_mfma.hip take over for the power-of-two-Q shape families.
#include "common.h"

namespace {

template <typename S>
__device__ __forceinline__ typename AccOf<S>::type ldv(const S* p) {
  return (typename AccOf<S>::type)(*p);
}

struct WinCoord {
  long long b;
  int ho, wo;
};

__device__ __forceinline__ WinCoord win_coord(const EpsP& p, long long w) {
  WinCoord c;
  const int hw = p.Ho * p.Wo;
  c.b = w / hw;
  const int rem = (int)(w - c.b * hw);
  c.ho = rem / p.Wo;
  c.wo = rem - c.ho * p.Wo;
  return c;
}

// stage the N*Q window features of this lane's window into its LDS column
template <typename S, typename A>
__device__ __forceinline__ void stage_window(const S* __restrict__ x, const EpsP& p, bool valid,
                                             const WinCoord& c, A* xs, int tid) {
  for (int n = 0; n < p.N; ++n) {
    const int pos = n / p.C, ch = n - pos * p.C;
    const int dh = pos / p.K, dw = pos - dh * p.K;
    const S* px = x + ch * p.s[0] + c.b * p.s[1] + (long long)(c.ho + dh) * p.s[2] +
                  (long long)(c.wo + dw) * p.s[3];
    for (int q = 0; q < p.Q; ++q)
      xs[(n * p.Q + q) * DCTN_WAVE + tid] = valid ? (A)px[q * p.s[4]] : A(0);
  }
}

template <typename A>
__device__ __forceinline__ void build_plo(const EpsP& p, const A* xs, A* plo, int tid) {
  for (int lo = 0; lo < p.LO; ++lo) {
    int t = lo;
    A pr = A(1);
    for (int d = p.m - 1; d >= 0; --d) {
      const int digit = t % p.Q;
      t /= p.Q;
      pr *= xs[((p.NH + d) * p.Q + digit) * DCTN_WAVE + tid];
    }
    plo[lo * DCTN_WAVE + tid] = pr;
  }
}

// ------------------------------------------------------------------------------------ forward
template <typename S, typename A, int OT>
__global__ __launch_bounds__(DCTN_WAVE) void eps_fwd_generic_k(const S* __restrict__ x,
                                                               const S* __restrict__ core,
                                                               S* __restrict__ out, EpsP p,
                                                               long long hi_per_slice) {
  extern __shared__ __align__(16) unsigned char smem[];
  A* xs = reinterpret_cast<A*>(smem);               // [N*Q][64]
  A* plo = xs + (size_t)p.N * p.Q * DCTN_WAVE;      // [LO][64]
  const int tid = threadIdx.x;
  const long long w = (long long)blockIdx.x * DCTN_WAVE + tid;
  const bool valid = w < p.Wn;
  WinCoord c = {0, 0, 0};
  if (valid) c = win_coord(p, w);
  stage_window<S, A>(x, p, valid, c, xs, tid);
  build_plo<A>(p, xs, plo, tid);

  // gridDim.y > 1: the high half of the core rows is split over grid.y (few windows would leave
  // most SIMDs idle behind one serial loop per wave); slices meet by atomic adds on a zeroed `out`
  const long long hi_begin = (long long)blockIdx.y * hi_per_slice;
  const long long hi_end = hi_begin + hi_per_slice < p.HI ? hi_begin + hi_per_slice : p.HI;
  for (int o0 = 0; o0 < p.O; o0 += OT) {
    A acc[OT];
#pragma unroll
    for (int j = 0; j < OT; ++j) acc[j] = A(0);
    for (long long hi = hi_begin; hi < hi_end; ++hi) {
      long long t = hi;
      A phi = A(1);
      for (int d = p.NH - 1; d >= 0; --d) {
        const int digit = (int)(t % p.Q);
        t /= p.Q;
        phi *= xs[(d * p.Q + digit) * DCTN_WAVE + tid];
      }
      const S* crow = core + (hi * p.LO) * p.O + o0;
      for (int lo = 0; lo < p.LO; ++lo) {
        const A pp = phi * plo[lo * DCTN_WAVE + tid];
#pragma unroll
        for (int j = 0; j < OT; ++j)
          if (o0 + j < p.O) acc[j] += pp * (A)crow[lo * p.O + j];
      }
    }
    if (valid) {
#pragma unroll
      for (int j = 0; j < OT; ++j)
        if (o0 + j < p.O) {
          if constexpr (sizeof(S) == sizeof(A)) {
            if (gridDim.y > 1) {
              atomicAdd(reinterpret_cast<A*>(out) + w * p.O + o0 + j, acc[j]);
              continue;
            }
          }
          out[w * p.O + o0 + j] = (S)acc[j];
        }
    }
  }
}

// ------------------------------------------------------------- backward: per-window d/d(factor)
// Writes gx[(n*Q+q)][w] = d(sum_o out[w,o] dY[w,o]) / d x_n[w,q] to the workspace (coalesced);
// eps_gather_dx_k then sums, for every input pixel, the K*K windows that cover it (no atomics,
// deterministic).
template <typename S, typename A>
__global__ __launch_bounds__(DCTN_WAVE) void eps_bwd_dfactor_generic_k(
    const S* __restrict__ x, const S* __restrict__ core, const S* __restrict__ dY,
    A* __restrict__ gxw, EpsP p, long long hi_per_slice) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int NQ = p.N * p.Q;
  const long long hi_begin = (long long)blockIdx.y * hi_per_slice;   // split as in the forward kernel
  const long long hi_end = hi_begin + hi_per_slice < p.HI ? hi_begin + hi_per_slice : p.HI;
  A* xs = reinterpret_cast<A*>(smem);      // [NQ][64]
  A* gx = xs + (size_t)NQ * DCTN_WAVE;     // [NQ][64]
  A* plo = gx + (size_t)NQ * DCTN_WAVE;    // [LO][64]
  A* U = plo + (size_t)p.LO * DCTN_WAVE;   // [LO][64]
  A* suf = U + (size_t)p.LO * DCTN_WAVE;   // [NH+1][64]
  A* dys = suf + (size_t)(p.NH + 1) * DCTN_WAVE;  // [O][64]
  const int tid = threadIdx.x;
  const long long w = (long long)blockIdx.x * DCTN_WAVE + tid;
  const bool valid = w < p.Wn;
  WinCoord c = {0, 0, 0};
  if (valid) c = win_coord(p, w);
  stage_window<S, A>(x, p, valid, c, xs, tid);
  build_plo<A>(p, xs, plo, tid);
  for (int e = 0; e < NQ; ++e) gx[e * DCTN_WAVE + tid] = A(0);
  for (int lo = 0; lo < p.LO; ++lo) U[lo * DCTN_WAVE + tid] = A(0);
  for (int o = 0; o < p.O; ++o) dys[o * DCTN_WAVE + tid] = valid ? (A)dY[w * p.O + o] : A(0);

  long long pw_top = 1;
  for (int d = 0; d + 1 < p.NH; ++d) pw_top *= p.Q;

  for (long long hi = hi_begin; hi < hi_end; ++hi) {
    // suffix products over the high factors
    {
      long long t = hi;
      A sfx = A(1);
      suf[p.NH * DCTN_WAVE + tid] = sfx;
      for (int d = p.NH - 1; d >= 0; --d) {
        const int digit = (int)(t % p.Q);
        t /= p.Q;
        sfx *= xs[(d * p.Q + digit) * DCTN_WAVE + tid];
        suf[d * DCTN_WAVE + tid] = sfx;
      }
    }
    const A phi = suf[tid];
    const S* crow = core + (hi * p.LO) * p.O;
    A V = A(0);
    for (int lo = 0; lo < p.LO; ++lo) {
      A sv = A(0);
      for (int o = 0; o < p.O; ++o) sv += (A)crow[lo * p.O + o] * dys[o * DCTN_WAVE + tid];
      V += plo[lo * DCTN_WAVE + tid] * sv;
      U[lo * DCTN_WAVE + tid] += phi * sv;
    }
    // leave-one-out over the high factors: prefix (running) * suffix (stored)
    A pre = A(1);
    long long pw = pw_top;
    for (int d = 0; d < p.NH; ++d) {
      const int digit = (int)((hi / pw) % p.Q);
      pw /= p.Q;
      const int e = (d * p.Q + digit) * DCTN_WAVE + tid;
      gx[e] += V * pre * suf[(d + 1) * DCTN_WAVE + tid];
      pre *= xs[e];
    }
  }
  // low factors
  for (int lo = 0; lo < p.LO; ++lo) {
    const A u = U[lo * DCTN_WAVE + tid];
    for (int e = 0; e < p.m; ++e) {
      A pr = A(1);
      int t = lo, de = 0;
      for (int d = p.m - 1; d >= 0; --d) {
        const int digit = t % p.Q;
        t /= p.Q;
        if (d == e)
          de = digit;
        else
          pr *= xs[((p.NH + d) * p.Q + digit) * DCTN_WAVE + tid];
      }
      gx[((p.NH + e) * p.Q + de) * DCTN_WAVE + tid] += u * pr;
    }
  }
  if (valid) {
    if (gridDim.y > 1) {
      for (int e = 0; e < NQ; ++e) atomicAdd(&gxw[(long long)e * p.Wn + w], gx[e * DCTN_WAVE + tid]);
    } else {
      for (int e = 0; e < NQ; ++e) gxw[(long long)e * p.Wn + w] = gx[e * DCTN_WAVE + tid];
    }
  }
}

}  // namespace

// dX[ch,b,h,w,q] = sum over positions (dh,dw) whose window (h-dh, w-dw) exists of
// gxw[((dh*K+dw)*C+ch)*Q+q][window].  One thread per dX element; dX contiguous (C,B,H,W,Q).
template <typename S, typename A>
__global__ void eps_gather_dx_k(const A* __restrict__ gxw, S* __restrict__ dX, EpsP p) {
  const long long total = (long long)p.C * p.B * p.H * p.W * p.Q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    long long t = idx;
    const int q = (int)(t % p.Q);
    t /= p.Q;
    const int wi = (int)(t % p.W);
    t /= p.W;
    const int hi = (int)(t % p.H);
    t /= p.H;
    const int b = (int)(t % p.B);
    const int ch = (int)(t / p.B);
    A acc = A(0);
    for (int dh = 0; dh < p.K; ++dh) {
      const int ho = hi - dh;
      if (ho < 0 || ho >= p.Ho) continue;
      for (int dw = 0; dw < p.K; ++dw) {
        const int wo = wi - dw;
        if (wo < 0 || wo >= p.Wo) continue;
        const long long win = ((long long)b * p.Ho + ho) * p.Wo + wo;
        const int n = (dh * p.K + dw) * p.C + ch;
        acc += gxw[(long long)(n * p.Q + q) * p.Wn + win];
      }
    }
    dX[idx] = (S)acc;
  }
}

template __global__ void eps_gather_dx_k<float, float>(const float*, float*, EpsP);
template __global__ void eps_gather_dx_k<double, double>(const double*, double*, EpsP);
template __global__ void eps_gather_dx_k<bf16_t, float>(const float*, bf16_t*, EpsP);

namespace {

// ------------------------------------------------------------------------- backward: dCore
// dCore[i,o] = sum_w P[w,i] dY[w,o].  One lane owns one core row i (digits packed in 128 bits),
// window features and dY are staged through LDS and read as broadcasts.  grid.y splits the
// windows; partial sums are combined with float atomics into an A-typed accumulator.
constexpr int DC_WB = 32;  // windows staged per step

template <typename S, typename A, int OT>
__global__ __launch_bounds__(DCTN_WAVE) void eps_bwd_dcore_generic_k(
    const S* __restrict__ x, const S* __restrict__ dY, A* __restrict__ dCoreAcc, EpsP p,
    long long win_per_block) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int NQ = p.N * p.Q;
  A* xs = reinterpret_cast<A*>(smem);        // [DC_WB][NQ]
  A* dys = xs + (size_t)DC_WB * NQ;          // [DC_WB][O]
  const int tid = threadIdx.x;
  const long long row = (long long)blockIdx.x * DCTN_WAVE + tid;
  const bool rvalid = row < p.R;
  unsigned long long pk0 = 0, pk1 = 0;
  {
    long long t = rvalid ? row : 0;
    const int per = 64 / p.bits;
    for (int n = p.N - 1; n >= 0; --n) {
      const unsigned long long digit = (unsigned long long)(t % p.Q);
      t /= p.Q;
      if (n < per)
        pk0 |= digit << (n * p.bits);
      else
        pk1 |= digit << ((n - per) * p.bits);
    }
  }
  const long long w_begin = (long long)blockIdx.y * win_per_block;
  long long w_end = w_begin + win_per_block;
  if (w_end > p.Wn) w_end = p.Wn;
  const int per = 64 / p.bits;
  const unsigned mask = (1u << p.bits) - 1u;

  for (int o0 = 0; o0 < p.O; o0 += OT) {
    A acc[OT];
#pragma unroll
    for (int j = 0; j < OT; ++j) acc[j] = A(0);
    for (long long w0 = w_begin; w0 < w_end; w0 += DC_WB) {
      const int nw = (int)((w_end - w0) < DC_WB ? (w_end - w0) : DC_WB);
      __syncthreads();
      for (int e = tid; e < nw * NQ; e += DCTN_WAVE) {
        const int wl = e / NQ, f = e - wl * NQ;
        const int n = f / p.Q, q = f - n * p.Q;
        const int pos = n / p.C, ch = n - pos * p.C;
        const int dh = pos / p.K, dw = pos - dh * p.K;
        const WinCoord c = win_coord(p, w0 + wl);
        xs[e] = (A)x[ch * p.s[0] + c.b * p.s[1] + (long long)(c.ho + dh) * p.s[2] +
                     (long long)(c.wo + dw) * p.s[3] + q * p.s[4]];
      }
      for (int e = tid; e < nw * p.O; e += DCTN_WAVE) dys[e] = (A)dY[w0 * p.O + e];
      __syncthreads();
      for (int wl = 0; wl < nw; ++wl) {
        const A* xw = xs + wl * NQ;
        A P = A(1);
        for (int n = 0; n < p.N; ++n) {
          const unsigned dg = (n < per) ? (unsigned)(pk0 >> (n * p.bits)) & mask
                                        : (unsigned)(pk1 >> ((n - per) * p.bits)) & mask;
          P *= xw[n * p.Q + dg];
        }
#pragma unroll
        for (int j = 0; j < OT; ++j)
          if (o0 + j < p.O) acc[j] += P * dys[wl * p.O + o0 + j];
      }
    }
    if (rvalid) {
#pragma unroll
      for (int j = 0; j < OT; ++j)
        if (o0 + j < p.O) atomicAdd(&dCoreAcc[row * p.O + o0 + j], acc[j]);
    }
  }
}

}  // namespace

template <typename S, typename A>
__global__ void convert_k(const A* __restrict__ src, S* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    dst[i] = (S)src[i];
}
template __global__ void convert_k<bf16_t, float>(const float*, bf16_t*, long long);

// ================================================================================== host side
int eps_fill_params(EpsP& p, const int64_t xs[5], int C, int B, int H, int W, int Q, int K, int O, int policy) {
  p.opts = policy & ~DCTN_PREC_MASK;
  if ((p.opts & ~DCTN_OPT_ALL) || (policy & DCTN_PREC_MASK) > DCTN_PREC_BF16 || policy < 0) return DCTN_ERR_UNSUPPORTED;
  if (C < 1 || B < 1 || Q < 1 || K < 1 || O < 1 || H < K || W < K) return DCTN_ERR_BAD_SHAPE;
  p.C = C; p.B = B; p.H = H; p.W = W; p.Q = Q; p.K = K; p.O = O;
  p.N = K * K * C;
  p.Ho = H - K + 1;
  p.Wo = W - K + 1;
  p.Wn = (long long)B * p.Ho * p.Wo;
  // Q^N must stay addressable
  long long R = 1;
  for (int n = 0; n < p.N; ++n) {
    R *= Q;
    if (R > (1LL << 40)) return DCTN_ERR_UNSUPPORTED;
  }
  p.R = R;
  for (int i = 0; i < 5; ++i) p.s[i] = xs[i];
  int m = 0;
  long long lo = 1;
  while (m < p.N && lo * Q <= 32) { lo *= Q; ++m; }
  p.m = m; p.LO = (int)lo; p.NH = p.N - m;
  p.HI = ipow_ll(Q, p.NH);
  p.bits = Q <= 16 ? 4 : 8;
  return DCTN_OK;
}

static size_t fwd_lds(const EpsP& p, size_t asz) {
  return ((size_t)p.N * p.Q + p.LO) * DCTN_WAVE * asz;
}
static size_t dfac_lds(const EpsP& p, size_t asz) {
  return ((size_t)2 * p.N * p.Q + 2 * p.LO + p.NH + 1 + p.O) * DCTN_WAVE * asz;
}
static size_t dcore_lds(const EpsP& p, size_t asz) {
  return ((size_t)DC_WB * (p.N * p.Q + p.O)) * asz;
}

// number of grid.y slices of the high half of the core rows: enough waves for ~4 per SIMD
static long long hi_slices(const EpsP& p, unsigned window_blocks) {
  long long sl = 4096 / (long long)(window_blocks ? window_blocks : 1);
  if (sl > p.HI) sl = p.HI;
  if (sl > 64) sl = 64;
  return sl < 1 ? 1 : sl;
}

template <typename S, typename A>
static int fwd_launch(const void* x, const void* core, void* out, const EpsP& p, hipStream_t st) {
  const size_t lds = fwd_lds(p, sizeof(A));
  if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)((p.Wn + DCTN_WAVE - 1) / DCTN_WAVE);
  const long long slices = sizeof(S) == sizeof(A) ? hi_slices(p, grid) : 1;
  const long long hps = (p.HI + slices - 1) / slices;
  const dim3 g2(grid, (unsigned)((p.HI + hps - 1) / hps));
  if (g2.y > 1 && dctn_zero_async(out, (size_t)p.Wn * p.O * sizeof(S), st) != DCTN_OK) return DCTN_ERR_LAUNCH;
  if (p.O <= 4) {
    (void)hipFuncSetAttribute((const void*)eps_fwd_generic_k<S, A, 4>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((eps_fwd_generic_k<S, A, 4>), g2, dim3(DCTN_WAVE), lds, st,
                       (const S*)x, (const S*)core, (S*)out, p, hps);
  } else {
    (void)hipFuncSetAttribute((const void*)eps_fwd_generic_k<S, A, 8>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((eps_fwd_generic_k<S, A, 8>), g2, dim3(DCTN_WAVE), lds, st,
                       (const S*)x, (const S*)core, (S*)out, p, hps);
  }
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("eps_fwd_generic");
  return DCTN_OK;
}

int eps_fwd_generic(const void* x, const void* core, void* out, EpsP p, int dtype, hipStream_t st) {
  switch (dtype) {
    case DCTN_F32: return fwd_launch<float, float>(x, core, out, p, st);
    case DCTN_F64: return fwd_launch<double, double>(x, core, out, p, st);
    case DCTN_BF16: return fwd_launch<bf16_t, float>(x, core, out, p, st);
  }
  return DCTN_ERR_BAD_DTYPE;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t eps_bwd_generic_workspace(const EpsP& p, int dtype, int need_dx, int need_dcore) {
  const size_t asz = dtype == DCTN_F64 ? 8 : 4;
  size_t total = 0;
  if (need_dx) total += align256((size_t)p.Wn * p.N * p.Q * asz);
  if (need_dcore && dtype == DCTN_BF16) total += align256((size_t)p.R * p.O * asz);
  return total;
}

template <typename S, typename A>
static int bwd_launch(const void* x, const void* core, const void* dY, void* dX, void* dCore,
                      void* ws, size_t ws_bytes, const EpsP& p, int dtype, hipStream_t st) {
  if (eps_bwd_generic_workspace(p, dtype, dX != nullptr, dCore != nullptr) > ws_bytes)
    return DCTN_ERR_WORKSPACE;
  unsigned char* wsp = (unsigned char*)ws;
  if (dX) {
    const size_t lds = dfac_lds(p, sizeof(A));
    if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
    A* gxw = (A*)wsp;
    wsp += align256((size_t)p.Wn * p.N * p.Q * sizeof(A));
    const unsigned grid = (unsigned)((p.Wn + DCTN_WAVE - 1) / DCTN_WAVE);
    const long long slices = hi_slices(p, grid);
    const long long hps = (p.HI + slices - 1) / slices;
    const dim3 g3(grid, (unsigned)((p.HI + hps - 1) / hps));
    if (g3.y > 1 && dctn_zero_async(gxw, (size_t)p.Wn * p.N * p.Q * sizeof(A), st) != DCTN_OK)
      return DCTN_ERR_LAUNCH;
    (void)hipFuncSetAttribute((const void*)eps_bwd_dfactor_generic_k<S, A>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((eps_bwd_dfactor_generic_k<S, A>), g3, dim3(DCTN_WAVE), lds, st,
                       (const S*)x, (const S*)core, (const S*)dY, gxw, p, hps);
    DCTN_CHECK_LAUNCH();
    const long long total = (long long)p.C * p.B * p.H * p.W * p.Q;
    const unsigned g2 = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((eps_gather_dx_k<S, A>), dim3(g2), dim3(256), 0, st, gxw, (S*)dX, p);
    DCTN_CHECK_LAUNCH();
  }
  if (dCore) {
    if (p.N > 2 * (64 / p.bits)) return DCTN_ERR_UNSUPPORTED;
    const size_t lds = dcore_lds(p, sizeof(A));
    if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
    A* acc = sizeof(S) == sizeof(A) ? (A*)dCore : (A*)wsp;
    if (dctn_zero_async(acc, (size_t)p.R * p.O * sizeof(A), st) != DCTN_OK)
      return DCTN_ERR_LAUNCH;
    const long long row_blocks = (p.R + DCTN_WAVE - 1) / DCTN_WAVE;
    long long chunks = 4096 / row_blocks;
    if (chunks < 1) chunks = 1;
    const long long max_chunks = (p.Wn + DC_WB - 1) / DC_WB;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks > 65535) chunks = 65535;
    long long wpb = (p.Wn + chunks - 1) / chunks;
    wpb = (wpb + DC_WB - 1) / DC_WB * DC_WB;
    chunks = (p.Wn + wpb - 1) / wpb;
    dim3 grid((unsigned)row_blocks, (unsigned)chunks);
    if (p.O <= 4) {
      (void)hipFuncSetAttribute((const void*)eps_bwd_dcore_generic_k<S, A, 4>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((eps_bwd_dcore_generic_k<S, A, 4>), grid, dim3(DCTN_WAVE), lds, st,
                         (const S*)x, (const S*)dY, acc, p, wpb);
    } else {
      (void)hipFuncSetAttribute((const void*)eps_bwd_dcore_generic_k<S, A, 8>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((eps_bwd_dcore_generic_k<S, A, 8>), grid, dim3(DCTN_WAVE), lds, st,
                         (const S*)x, (const S*)dY, acc, p, wpb);
    }
    DCTN_CHECK_LAUNCH();
    if constexpr (sizeof(S) != sizeof(A)) {
      const long long n = p.R * p.O;
      const unsigned g = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
      hipLaunchKernelGGL((convert_k<S, A>), dim3(g), dim3(256), 0, st, (const A*)acc, (S*)dCore, n);
      DCTN_CHECK_LAUNCH();
    }
  }
  dctn_set_last_kernel("eps_bwd_generic");
  return DCTN_OK;
}

// deterministic per-pixel sum of the per-window factor gradients gxw[N*Q][Wn] (float32 or float64
// accumulator type of `dtype`) into dX; shared by the generic and the MFMA backward kernels
int eps_gather_dx_launch(const void* gxw, void* dX, const EpsP& p, int dtype, hipStream_t st) {
  const long long total = (long long)p.C * p.B * p.H * p.W * p.Q;
  const unsigned g2 = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  switch (dtype) {
    case DCTN_F32:
      hipLaunchKernelGGL((eps_gather_dx_k<float, float>), dim3(g2), dim3(256), 0, st, (const float*)gxw, (float*)dX, p);
      break;
    case DCTN_F64:
      hipLaunchKernelGGL((eps_gather_dx_k<double, double>), dim3(g2), dim3(256), 0, st, (const double*)gxw, (double*)dX, p);
      break;
    case DCTN_BF16:
      hipLaunchKernelGGL((eps_gather_dx_k<bf16_t, float>), dim3(g2), dim3(256), 0, st, (const float*)gxw, (bf16_t*)dX, p);
      break;
    default: return DCTN_ERR_BAD_DTYPE;
  }
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

int eps_bwd_generic(const void* x, const void* core, const void* dY, void* dX, void* dCore,
                    void* ws, size_t ws_bytes, EpsP p, int dtype, hipStream_t st) {
  switch (dtype) {
    case DCTN_F32:
      return bwd_launch<float, float>(x, core, dY, dX, dCore, ws, ws_bytes, p, dtype, st);
    case DCTN_F64:
      return bwd_launch<double, double>(x, core, dY, dX, dCore, ws, ws_bytes, p, dtype, st);
    case DCTN_BF16:
      return bwd_launch<bf16_t, float>(x, core, dY, dX, dCore, ws, ws_bytes, p, dtype, st);
  }
  return DCTN_ERR_BAD_DTYPE;
}
