// MFMA EPS kernels, family "q2-reg": Q == 2 and a core small enough to live in registers
// (N = K*K*C in {8, 9}: the 3x3 single-channel MNIST layer of BASELINE config 2 and the 2x2
// two-channel layer).  Operands are bf16, accumulation f32 (v_mfma_f32_32x32x16_bf16).
//
// Forward (replaces dctn/eps.py:19-40).  The reference's path is K-R half 0, K-R half 1, one GEMM
// core x half0, a per-window dot with half 1.  Here, with a = index over the first n0 factors
// (A = 2^n0), b over the last n1 (Bn = 2^n1):
//     T[(b,o), w] = sum_a core[a,b,o] * P0[w,a]      <- MFMA, M = (b,o) rows, N = 32 windows, K = a
//     out[w,o]    = sum_b P1[w,b] * T[(b,o), w]      <- lane-local epilogue
// A lane owns ONE window, 64 windows per wave step (the kernels are bounded by VALU issue, so no
// lane may repeat another lane's loads, address arithmetic or products): it builds the whole P0
// row of its window in registers; the MFMA tile has only 32 columns, so the step runs as two
// "sets" (windows of lanes 0-31, then of lanes 32-63) and v_permlane32_swap turns the two k-halves
// a lane built for its own window into its operand share for set 0 and for set 1.  Each lane gets
// 16 rows (b,o) of T per tile back, weights them with P1 of the set's window (also handed over by
// permlane swaps), and one more swap per output joins the two row halves so that every lane ends
// up with the outputs of its own window.  Nothing but x and out touches HBM; no LDS in the main
// loop, no barriers.
//
// Backward dCore[a,b,o] = sum_w P0[w,a] P1[w,b] dY[w,o]  reduces over windows, so windows must
// become the MFMA K index while lanes own windows.  The transpose is done ON the matrix core:
// multiplying the lane-owns-window fragment (as A operand) by an identity B operand returns the
// tile with the feature index on the lane and 16 windows in the accumulator registers — exactly
// the layout the next MFMA consumes as an operand summing over windows (no LDS round trip).
//     Z[w,(b,o)] = P1[w,b] dY[w,o];   dCoreT[(b,o), a] += Zt (A operand) x P0t (B operand)
// Per-wave partial sums stay in registers over all of the wave's windows, are reduced over the
// workgroup's waves in LDS, written to a workspace and summed by a second small kernel
// (deterministic: no float atomics).
#include "common.h"

#include <cstdlib>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) int int2v;
typedef __attribute__((ext_vector_type(4))) int int4v;
typedef __attribute__((ext_vector_type(2))) float f32x2;

namespace {

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

template <typename S> __device__ __forceinline__ float to_f32(S v) { return (float)v; }

// sum over the two lane halves (lane l and l^32), result in every lane
__device__ __forceinline__ float half_sum(float v) {
  const int iv = __float_as_int(v);
  const int2v r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}

// a <- [a.lo, b.lo], b <- [a.hi, b.hi]  (lo / hi = lanes 0-31 / 32-63): with a, b = the values a lane
// computed for the first / second lane half's role, a becomes the operand of set 0 (windows of lanes
// 0-31) and b the operand of set 1 (windows of lanes 32-63).
__device__ __forceinline__ void swap_halves(float& a, float& b) {
  const int2v r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
  a = __int_as_float(r[0]);
  b = __int_as_float(r[1]);
}
__device__ __forceinline__ void swap_halves(bf16x8& a, bf16x8& b) {
  int4v ia = __builtin_bit_cast(int4v, a), ib = __builtin_bit_cast(int4v, b);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int2v r = __builtin_amdgcn_permlane32_swap(ia[i], ib[i], false, false);
    ia[i] = r[0];
    ib[i] = r[1];
  }
  a = __builtin_bit_cast(bf16x8, ia);
  b = __builtin_bit_cast(bf16x8, ib);
}

// unsigned 32-bit division by an invariant (Granlund-Montgomery, round-up variant):
//   q = (t + ((n - t) >> s1)) >> s2,  t = umulhi(M, n)
struct FastDiv {
  unsigned M, s1, s2;
};
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& d) {
  const unsigned t = __umulhi(d.M, n);
  return (t + ((n - t) >> d.s1)) >> d.s2;
}

#define MFMA_MAXN 16
struct MfmaP {
  int C, B, H, W, K, O, Ho, Wo;
  long long Wn, ngroups;  // groups of 64 windows (Wn < 2^31 in this family)
  long long s[5];
  unsigned foffb[MFMA_MAXN];  // BYTE offset of factor n relative to the window's top-left pixel
  unsigned s1b, s2b, s3b, s4b;  // byte strides of x (batch, row, column, feature); x spans < 4 GiB
  FastDiv div_hw, div_wo;
  unsigned rowoffb[MFMA_MAXN];  // row loads: BYTE offset of window row (dh, ch), index dh*C + ch
  unsigned x_bytes;             // extent of x in bytes (row loads are clamped to stay inside)
  int rowvec_ok;                // one 16-byte load per window row usable (bf16, Q=2 contiguous, K <= 4)
  unsigned row_wrap, img_wrap;  // offset corrections when a window walk wraps a row / an image
  int inc_ok;                   // incremental walk usable (Wo >= 16, Ho >= 4)
  long long gpw;                // window groups per wave (contiguous range)
  int vec_ok;             // x: last stride 1, even strides, 4-byte aligned base (bf16 pair loads)
};

// Raw features of one window, as loaded (2 values of type S per factor): the loads are issued
// back to back with no control flow between them and no use of the data, so a whole group's
// loads are in flight together and the next group's can be issued before this one is consumed.
// Addresses are uniform base + 32-bit lane offset (+ uniform per-factor offset).
// Lanes without a window read window 0 (masked later).
template <typename S, int N, bool VEC>
struct RawWindow {
  typedef typename std::conditional<sizeof(S) == 2, unsigned, float2>::type vec_t;
  vec_t v[VEC ? N : 1];
  S e[VEC ? 1 : N][2];
  uint4 row[MFMA_MAXN / 2];  // row-vector mode (bf16): K pixels of one window row per 16-byte load
  unsigned shift;            // bit rw set: that row's load was moved back by one pixel (tensor end)
};

// Byte offset of the top-left pixel of a lane's window, advanced by 64 windows per step without
// divisions or integer multiplies (the wave walks a contiguous range of window groups).
struct WinIter {
  unsigned off;   // byte offset of the window's top-left pixel (feature 0)
  int wo, ho;
};
__device__ __forceinline__ WinIter win_begin(const MfmaP& p, long long w) {
  const unsigned wu = w < p.Wn ? (unsigned)w : 0u;
  const unsigned b = fdiv(wu, p.div_hw);
  const unsigned rem = wu - b * (unsigned)(p.Ho * p.Wo);
  const unsigned ho = fdiv(rem, p.div_wo), wo = rem - ho * (unsigned)p.Wo;
  WinIter it;
  it.off = b * p.s1b + ho * p.s2b + wo * p.s3b;
  it.wo = (int)wo;
  it.ho = (int)ho;
  return it;
}
// requires Wo >= 16 and Ho >= 4 (host-checked: p.inc_ok): at most four row wraps and one image wrap
__device__ __forceinline__ void win_advance64(const MfmaP& p, WinIter& it) {
  it.wo += 64;
  it.off += 64u * p.s3b;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool wrap = it.wo >= p.Wo;
    it.wo -= wrap ? p.Wo : 0;
    it.ho += wrap ? 1 : 0;
    it.off += wrap ? p.row_wrap : 0u;
  }
  const bool iw = it.ho >= p.Ho;
  it.ho -= iw ? p.Ho : 0;
  it.off += iw ? p.img_wrap : 0u;
}

// ROWS > 0: row-vector mode with ROWS = K*C rows (one global_load_dwordx4 per row instead of K
// dword loads: 3x fewer memory requests per window for the 3x3 kernel)
template <typename S, int N, bool VEC, int ROWS>
__device__ __forceinline__ void issue_window(const S* __restrict__ x, const MfmaP& p, unsigned off0,
                                             RawWindow<S, N, VEC>& raw) {
  const char* xb = reinterpret_cast<const char*>(x);
  if constexpr (ROWS > 0) {
    // a 16-byte load at the last pixel of the tensor would run 4 bytes past its end: move it back
    // by one pixel there and remember to select the next dword
    raw.shift = 0;
#pragma unroll
    for (int rw = 0; rw < ROWS; ++rw) {
      unsigned o = off0 + p.rowoffb[rw];
      if (o + 16u > p.x_bytes) { o -= 4u; raw.shift |= 1u << rw; }
      raw.row[rw] = *reinterpret_cast<const uint4*>(xb + (size_t)o);
    }
  } else if constexpr (VEC) {
#pragma unroll
    for (int n = 0; n < N; ++n)
      raw.v[n] = *reinterpret_cast<const typename RawWindow<S, N, VEC>::vec_t*>(xb + (size_t)(off0 + p.foffb[n]));
  } else {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      raw.e[n][0] = *reinterpret_cast<const S*>(xb + (size_t)(off0 + p.foffb[n]));
      raw.e[n][1] = *reinterpret_cast<const S*>(xb + (size_t)(off0 + p.foffb[n] + p.s4b));
    }
  }
}

template <typename S, int N, bool VEC, int ROWS>
__device__ __forceinline__ void unpack_window(const RawWindow<S, N, VEC>& raw, const MfmaP& p,
                                              float (&xv)[N][2]) {
  if constexpr (ROWS > 0) {
    constexpr int KK = N / ROWS;  // pixels per row = K
#pragma unroll
    for (int rw = 0; rw < ROWS; ++rw) {
      const bool sh = (raw.shift >> rw) & 1u;
      const unsigned d[4] = {raw.row[rw].x, raw.row[rw].y, raw.row[rw].z, raw.row[rw].w};
#pragma unroll
      for (int dw = 0; dw < KK; ++dw) {
        // factor n = (dh*K + dw)*C + ch for row rw = dh*C + ch
        const int dh = rw / p.C, ch = rw - dh * p.C;
        (void)dh; (void)ch;
        const unsigned u = sh ? d[dw + 1 < 4 ? dw + 1 : 3] : d[dw];
        // static factor index needs C at compile time: ROWS = K*C and N = K*K*C give C = ROWS*ROWS/N
        constexpr int CC = ROWS * ROWS / N;
        const int n = ((rw / CC) * KK + dw) * CC + (rw % CC);
        xv[n][0] = __uint_as_float(u << 16);
        xv[n][1] = __uint_as_float(u & 0xffff0000u);
      }
    }
    return;
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
    if constexpr (VEC && sizeof(S) == 2) {
      xv[n][0] = __uint_as_float(raw.v[n] << 16);
      xv[n][1] = __uint_as_float(raw.v[n] & 0xffff0000u);
    } else if constexpr (VEC) {
      xv[n][0] = raw.v[n].x;
      xv[n][1] = raw.v[n].y;
    } else {
      xv[n][0] = to_f32(raw.e[n][0]);
      xv[n][1] = to_f32(raw.e[n][1]);
    }
  }
}

// OP consecutive values of one window (dY row / out row), as one vector access when VEC
template <typename S, int OP>
struct alignas(sizeof(S) * OP) RowPack {
  S e[OP];
};
template <typename S, int OP, bool VEC>
__device__ __forceinline__ void issue_row(const S* __restrict__ src, int O, RowPack<S, OP>& pk) {
  if constexpr (VEC) {
    pk = *reinterpret_cast<const RowPack<S, OP>*>(src);
  } else {
#pragma unroll
    for (int o = 0; o < OP; ++o) pk.e[o] = src[o < O ? o : 0];
  }
}
template <typename S, int OP>
__device__ __forceinline__ void unpack_row(const RowPack<S, OP>& pk, int O, float (&v)[OP]) {
#pragma unroll
  for (int o = 0; o < OP; ++o) v[o] = o < O ? to_f32(pk.e[o]) : 0.f;
}
template <typename S, int OP, bool VEC>
__device__ __forceinline__ void store_row(S* __restrict__ dst, int O, const float (&v)[OP]) {
  if constexpr (VEC) {
    struct alignas(sizeof(S) * OP) Pack { S e[OP]; };
    Pack pk;
#pragma unroll
    for (int o = 0; o < OP; ++o) pk.e[o] = (S)v[o];
    *reinterpret_cast<Pack*>(dst) = pk;
  } else {
#pragma unroll
    for (int o = 0; o < OP; ++o)
      if (o < O) dst[o] = (S)v[o];
  }
}

// The whole P0 row of this lane's window, as the two k-halves of every 16-wide k-step:
// X[s][j] = P0[w][a = 16 s + j], Y[s][j] = P0[w][a = 16 s + 8 + j]; a's MSB = factor 0.
template <int N0>
__device__ __forceinline__ void build_p0(const float (*xv)[2], bf16x8 (&X)[(1 << N0) / 16],
                                         bf16x8 (&Y)[(1 << N0) / 16]) {
  constexpr int KS = (1 << N0) / 16;
  float lo8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    lo8[j] = xv[N0 - 3][(j >> 2) & 1] * xv[N0 - 2][(j >> 1) & 1] * xv[N0 - 1][j & 1];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    float hi = 1.f;
#pragma unroll
    for (int t = 0; t < N0 - 4; ++t) hi = t == 0 ? xv[N0 - 5][s & 1] : hi * xv[N0 - 5 - t][(s >> t) & 1];
    const float h0 = N0 > 4 ? hi * xv[N0 - 4][0] : xv[N0 - 4][0];
    const float h1 = N0 > 4 ? hi * xv[N0 - 4][1] : xv[N0 - 4][1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      X[s][j] = (bf16_t)(h0 * lo8[j]);
      Y[s][j] = (bf16_t)(h1 * lo8[j]);
    }
  }
}

// ------------------------------------------------------------------------------------ forward
// Row code of accumulator register v of M-tile t (lane-half bit h excluded):
//   code = (t << 4) | ((v >> 2) << 2) | (v & 3);   o = code & (OP-1);   b = ((code >> LOGO) << 1) | h
template <typename S, int N0, int N1, int OP, bool XVEC, bool OVEC, int ROWS>
__global__ __launch_bounds__(256) void eps_fwd_q2reg_k(const S* __restrict__ x,
                                                       const S* __restrict__ core,
                                                       S* __restrict__ out, MfmaP p) {
  constexpr int N = N0 + N1, A = 1 << N0, BN = 1 << N1, KS = A / 16, MT = BN * OP / 32;
  constexpr int LOGO = ilog2(OP);
  static_assert(MT >= 1 && KS >= 1, "tile too small");
  // A-operand fragments of the core, staged through LDS in fragment order
  // cs[((t*KS + s)*64 + lane)*8 + j] = core[a = 16s + 8h + j][b][o], where lane = 32h + row and
  // row = (g << 3) | (h' << 2) | i, code = (t << 4) | (g << 2) | i = (b >> 1 << LOGO) | o, h' = b & 1
  constexpr int TOT = A * BN * OP, PER = TOT / 256;
  static_assert(TOT % 256 == 0, "core staging assumes a multiple of 256 elements");
  __shared__ __attribute__((aligned(16))) bf16_t cs[TOT];
  const int tid = threadIdx.x, lane = tid & 63;
  {
    S tmp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {  // all loads in flight together: one memory round trip
      const int e = tid + 256 * i, o = e % OP, ab = e / OP;
      tmp[i] = core[(long long)ab * p.O + (o < p.O ? o : 0)];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i, o = e % OP, ab = e / OP, bb = ab % BN, aa = ab / BN;
      const int code = ((bb >> 1) << LOGO) | o;
      const int row = (((code >> 2) & 3) << 3) | ((bb & 1) << 2) | (code & 3);
      const int dst = ((((code >> 4) * KS + (aa >> 4)) * 64 + ((aa >> 3) & 1) * 32 + row) << 3) | (aa & 7);
      cs[dst] = o < p.O ? (bf16_t)to_f32(tmp[i]) : (bf16_t)0.f;
    }
  }
  __syncthreads();
  bf16x8 cf[MT][KS];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int s = 0; s < KS; ++s)
      cf[t][s] = *reinterpret_cast<const bf16x8*>(&cs[((t * KS + s) * 64 + lane) * 8]);

  // XCD-contiguous mapping: workgroups b, b+8, ... share an XCD (round-robin dispatch), so give
  // each XCD one contiguous eighth of the windows: its L2 then fetches only that part of x.
  const long long nb = gridDim.x;
  const long long vb = (nb % 8 == 0) ? (long long)(blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;
  const long long wave = vb * 4 + (tid >> 6);
  const long long g0 = wave * p.gpw;
  const long long g1 = g0 + p.gpw < p.ngroups ? g0 + p.gpw : p.ngroups;
  RawWindow<S, N, XVEC> raw;
  WinIter it = win_begin(p, g0 * 64 + lane);
  if (g0 < g1) issue_window<S, N, XVEC, ROWS>(x, p, g0 * 64 + lane < p.Wn ? it.off : 0u, raw);
  for (long long g = g0; g < g1; ++g) {
    const long long w = g * 64 + lane;
    const bool valid = w < p.Wn;
    float xv[N][2];
    unpack_window<S, N, XVEC, ROWS>(raw, p, xv);
    {  // prefetch the next group of this wave (the last iteration re-reads its own group)
      if (g + 1 < g1) {
        if (p.inc_ok) win_advance64(p, it); else it = win_begin(p, w + 64);
      }
      issue_window<S, N, XVEC, ROWS>(x, p, (g + 1 < g1 ? w + 64 : w) < p.Wn ? it.off : 0u, raw);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this group's arithmetic
    }
    bf16x8 pf0[KS], pf1[KS];   // after the swaps: B operands of set 0 / set 1
    build_p0<N0>(xv, pf0, pf1);
#pragma unroll
    for (int s = 0; s < KS; ++s) swap_halves(pf0[s], pf1[s]);
    // P1 over the last n1 factors: b bit u <-> factor N-1-u; bit 0 (factor N-1) is the lane half of
    // the accumulator row.  m0 / m1: multipliers this lane applies in set 0 / set 1.
    float m0[BN / 2], m1[BN / 2];
#pragma unroll
    for (int bh = 0; bh < BN / 2; ++bh) {
      float v = 1.f;
#pragma unroll
      for (int u = 1; u < N1; ++u) v = u == 1 ? xv[N - 2][bh & 1] : v * xv[N - 1 - u][(bh >> (u - 1)) & 1];
      m0[bh] = v * xv[N - 1][0];
      m1[bh] = v * xv[N - 1][1];
      swap_halves(m0[bh], m1[bh]);
    }
    f32x2 res0[OP / 2], res1[OP / 2];
#pragma unroll
    for (int o = 0; o < OP / 2; ++o) { res0[o] = f32x2{0.f, 0.f}; res1[o] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
      for (int set = 0; set < 2; ++set) {
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf[t][s], set ? pf1[s] : pf0[s], acc, 0, 0, 0);
        // rows v, v+1 (v even) are outputs o, o+1 of the same b: one packed FMA
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          const int code = (t << 4) | ((v >> 2) << 2) | (v & 3);
          const float mm = set ? m1[code >> LOGO] : m0[code >> LOGO];
          f32x2& dst = set ? res1[(code & (OP - 1)) >> 1] : res0[(code & (OP - 1)) >> 1];
          dst = __builtin_elementwise_fma(f32x2{acc[v], acc[v + 1]}, f32x2{mm, mm}, dst);
        }
      }
    }
    // join the two row halves: lanes 0-31 end up with set 0 (their own windows), lanes 32-63 with set 1
    float res[OP];
#pragma unroll
    for (int o = 0; o < OP; ++o) {
      float a0 = res0[o >> 1][o & 1], a1 = res1[o >> 1][o & 1];
      swap_halves(a0, a1);
      res[o] = a0 + a1;
    }
    if (valid) store_row<S, OP, OVEC>(out + w * p.O, p.O, res);
  }
}

// ------------------------------------------------------------------------------ backward: dCore
// feature index m of Z: code = m = (mt << 5) | (s << 4) | (h << 3) | j;  o = m & (OP-1), b = m >> LOGO
constexpr int BWD_WAVES = 8;  // waves per workgroup of the dCore kernel (one LDS reduction per block)

template <typename S, int N0, int N1, int OP, bool XVEC, bool OVEC, int ROWS>
__global__ __launch_bounds__(64 * BWD_WAVES) void eps_bwd_dcore_q2reg_k(const S* __restrict__ x,
                                                             const S* __restrict__ dY,
                                                             float* __restrict__ partial, MfmaP p) {
  constexpr int N = N0 + N1, A = 1 << N0, BN = 1 << N1, KS = A / 16, MT = BN * OP / 32;
  constexpr int AT = A >= 32 ? A / 32 : 1;
  constexpr int LOGO = ilog2(OP);
  __shared__ float red[BWD_WAVES][32 * 32];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5, wv = tid >> 6;

  // identity B operands of the transposing MFMA: element j of k-step parity sp is
  // [16*sp + 8*h + j == r]
  bf16x8 ident[2];
#pragma unroll
  for (int sp = 0; sp < 2; ++sp)
#pragma unroll
    for (int j = 0; j < 8; ++j) ident[sp][j] = (bf16_t)((16 * sp + 8 * h + j == r) ? 1.f : 0.f);

  f32x16 acc[MT][AT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int a = 0; a < AT; ++a)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[t][a][v] = 0.f;

  const long long nb = gridDim.x;
  const long long vb = (nb % 8 == 0) ? (long long)(blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;
  const long long wave = vb * BWD_WAVES + wv;
  const long long g0 = wave * p.gpw;
  const long long g1 = g0 + p.gpw < p.ngroups ? g0 + p.gpw : p.ngroups;
  RawWindow<S, N, XVEC> raw;
  RowPack<S, OP> rawdy;
  WinIter it = win_begin(p, g0 * 64 + lane);
  if (g0 < g1) {
    const long long w0 = g0 * 64 + lane;
    issue_window<S, N, XVEC, ROWS>(x, p, w0 < p.Wn ? it.off : 0u, raw);
    issue_row<S, OP, OVEC>(dY + (w0 < p.Wn ? w0 : 0) * p.O, p.O, rawdy);
  }
  for (long long g = g0; g < g1; ++g) {
    const long long w = g * 64 + lane;
    const bool valid = w < p.Wn;
    float xv[N][2];
    unpack_window<S, N, XVEC, ROWS>(raw, p, xv);
    float dy[OP];
    unpack_row<S, OP>(rawdy, p.O, dy);
    {  // prefetch the next group of this wave
      const long long wn = g + 1 < g1 ? w + 64 : w;
      if (g + 1 < g1) {
        if (p.inc_ok) win_advance64(p, it); else it = win_begin(p, wn);
      }
      issue_window<S, N, XVEC, ROWS>(x, p, wn < p.Wn ? it.off : 0u, raw);
      issue_row<S, OP, OVEC>(dY + (wn < p.Wn ? wn : 0) * p.O, p.O, rawdy);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this group's arithmetic
    }
    if (!valid) {  // lanes past the last window contribute nothing: P0 carries factor 0
      xv[0][0] = 0.f;
      xv[0][1] = 0.f;
    }

    // P0 of the lane's own window -> A operands of set 0 / set 1 -> transposed on the matrix core:
    // features on lanes, windows in registers = B operand fragments summing over windows
    bf16x8 pf[2][KS];
    build_p0<N0>(xv, pf[0], pf[1]);
#pragma unroll
    for (int s = 0; s < KS; ++s) swap_halves(pf[0][s], pf[1][s]);
    bf16x8 p0t[2][AT][2];
#pragma unroll
    for (int set = 0; set < 2; ++set)
#pragma unroll
      for (int a = 0; a < AT; ++a) {
        f32x16 d;
#pragma unroll
        for (int v = 0; v < 16; ++v) d[v] = 0.f;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
          if (2 * a + sp < KS)
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[set][2 * a + sp], ident[sp], d, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; ++v) p0t[set][a][v >> 3][v & 7] = (bf16_t)d[v];
      }
    // full P1 table of this window (b bit u <-> factor N-1-u)
    float p1[BN];
#pragma unroll
    for (int b = 0; b < BN; ++b) {
      float v = 1.f;
#pragma unroll
      for (int u = 0; u < N1; ++u) v = u == 0 ? xv[N - 1][b & 1] : v * xv[N - 1 - u][(b >> u) & 1];
      p1[b] = v;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      // Z of the own window for both k-halves, then the same hand-over as for P0
      bf16x8 zf[2][2];   // [set after the swap][sp]
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ma = (t << 5) | (sp << 4) | j;        // first k-half
          const int mb = ma | 8;                           // second k-half
          zf[0][sp][j] = (bf16_t)(p1[ma >> LOGO] * dy[ma & (OP - 1)]);
          zf[1][sp][j] = (bf16_t)(p1[mb >> LOGO] * dy[mb & (OP - 1)]);
        }
        swap_halves(zf[0][sp], zf[1][sp]);
      }
#pragma unroll
      for (int set = 0; set < 2; ++set) {
        f32x16 d;
#pragma unroll
        for (int v = 0; v < 16; ++v) d[v] = 0.f;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[set][sp], ident[sp], d, 0, 0, 0);
        bf16x8 zt[2];
#pragma unroll
        for (int v = 0; v < 16; ++v) zt[v >> 3][v & 7] = (bf16_t)d[v];
#pragma unroll
        for (int a = 0; a < AT; ++a)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            acc[t][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zt[s2], p0t[set][a][s2], acc[t][a], 0, 0, 0);
      }
    }
  }

  // workgroup reduction of the per-wave partial dCoreT tiles, then one coalesced store per block
  float* dst = partial + (long long)blockIdx.x * (MT * 32) * (AT * 32);
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int a = 0; a < AT; ++a) {
      __syncthreads();
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = (v & 3) + 8 * (v >> 2) + 4 * h;
        red[wv][row * 32 + r] = acc[t][a][v];
      }
      __syncthreads();
      for (int e = tid; e < 1024; e += 64 * BWD_WAVES) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < BWD_WAVES; ++k) sum += red[k][e];
        const int row = e >> 5, col = e & 31;
        dst[(long long)(t * 32 + row) * (AT * 32) + a * 32 + col] = sum;
      }
    }
}

// dCore[a][b][o] = sum_blocks partial[blk][m = b*OP + o][a].  One workgroup per 32 consecutive
// (m, a) entries: lane (k8 = tid / 32, c = tid % 32) streams every 8th block's 128-byte segment
// (coalesced), the 8 partial sums meet in LDS.
template <typename S>
__global__ __launch_bounds__(256) void eps_bwd_dcore_reduce_k(const float* __restrict__ partial,
                                                              S* __restrict__ dCore, int nblk, int A,
                                                              int BN, int O, int OP, int ACOLS) {
  __shared__ float red[8][32];
  const int tid = threadIdx.x, c = tid & 31, k8 = tid >> 5;
  const long long stride = (long long)BN * OP * ACOLS;
  const long long e = (long long)blockIdx.x * 32 + c;  // flat (m, a) index
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  int k = k8;
  for (; k + 56 < nblk; k += 64) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += partial[(k + 8 * i) * stride + e];
  }
  for (; k < nblk; k += 8) acc[0] += partial[k * stride + e];
  red[k8][c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (k8 == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][c];
    const int m = (int)(e / ACOLS), a = (int)(e % ACOLS);
    const int b = m / OP, o = m % OP;
    if (a < A && o < O) dCore[((long long)a * BN + b) * O + o] = (S)t;
  }
}

int next_pow2(int v) {
  int r = 1;
  while (r < v) r <<= 1;
  return r;
}

bool family_ok(const EpsP& p, int dtype, int precision) {
  if (p.Q != 2) return false;
  if (dtype == DCTN_F64) return false;
  if (dtype == DCTN_F32 && precision != DCTN_PREC_BF16) return false;
  if (p.N != 8 && p.N != 9) return false;
  if (p.Wn >= (1LL << 31) - 64) return false;
  {  // 32-bit byte offsets: non-negative strides and an extent below 4 GiB
    long long ext = 0;
    const long long dims[5] = {p.C, p.B, p.H, p.W, p.Q};
    for (int i = 0; i < 5; ++i) {
      if (p.s[i] < 0) return false;
      ext += (dims[i] - 1) * p.s[i];
    }
    if ((ext + 1) * 4 >= (1LL << 32)) return false;
  }
  const int op = next_pow2(p.O);
  return op >= 2 ? op <= 16 : true;
}

FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  if (d <= 1) {
    f.M = 0; f.s1 = 0; f.s2 = 0;
    return f;
  }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.M = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s1 = 1;
  f.s2 = l - 1;
  return f;
}

void fill_mp(MfmaP& m, const EpsP& p, const void* x, int dtype) {
  const long long esz_ = dtype == DCTN_BF16 ? 2 : 4;
  for (int n = 0; n < p.N && n < MFMA_MAXN; ++n) {
    const int pos = n / p.C, ch = n - pos * p.C;
    const int dh = pos / p.K, dw = pos - dh * p.K;
    m.foffb[n] = (unsigned)((ch * p.s[0] + dh * p.s[2] + dw * p.s[3]) * esz_);
  }
  m.s1b = (unsigned)(p.s[1] * esz_); m.s2b = (unsigned)(p.s[2] * esz_);
  m.s3b = (unsigned)(p.s[3] * esz_); m.s4b = (unsigned)(p.s[4] * esz_);
  {
    long long ext = 0;
    const long long dims[5] = {p.C, p.B, p.H, p.W, p.Q};
    for (int i = 0; i < 5; ++i) ext += (dims[i] - 1) * p.s[i];
    m.x_bytes = (unsigned)((ext + 1) * esz_);
    const int rows = p.K * p.C;
    for (int rw = 0; rw < rows && rw < MFMA_MAXN; ++rw) {
      const int dh = rw / p.C, ch = rw - dh * p.C;
      m.rowoffb[rw] = (unsigned)((ch * p.s[0] + dh * p.s[2]) * esz_);
    }
    // K pixels of a row in one 16-byte load: bf16 pairs (4 bytes per pixel), pixels contiguous
    m.rowvec_ok = dtype == DCTN_BF16 && p.s[4] == 1 && p.s[3] == 2 && p.K <= 4 && rows <= MFMA_MAXN / 2 &&
                  ((p.N == 9 && rows == 3) || (p.N == 8 && rows == 4)) && m.x_bytes >= 32;
  }
  m.row_wrap = (unsigned)(p.s[2] * esz_) - (unsigned)p.Wo * (unsigned)(p.s[3] * esz_);
  m.img_wrap = (unsigned)(p.s[1] * esz_) - (unsigned)p.Ho * (unsigned)(p.s[2] * esz_);
  m.inc_ok = p.Wo >= 16 && p.Ho >= 4;
  m.gpw = 1;
  m.div_hw = make_fastdiv((unsigned)(p.Ho * p.Wo));
  m.div_wo = make_fastdiv((unsigned)p.Wo);
  m.C = p.C; m.B = p.B; m.H = p.H; m.W = p.W; m.K = p.K; m.O = p.O; m.Ho = p.Ho; m.Wo = p.Wo;
  m.Wn = p.Wn;
  m.ngroups = (p.Wn + 63) / 64;
  for (int i = 0; i < 5; ++i) m.s[i] = p.s[i];
  const size_t esz = dtype == DCTN_BF16 ? 2 : 4;
  m.vec_ok = p.s[4] == 1 && p.s[0] % 2 == 0 && p.s[1] % 2 == 0 && p.s[2] % 2 == 0 &&
             p.s[3] % 2 == 0 && ((uintptr_t)x % (2 * esz)) == 0;
}

constexpr int FWD_BLOCKS_PER_CU = 4;
constexpr int NUM_CU = 256;

int bwd_grid(MfmaP& m) {
  long long blocks = (m.ngroups + BWD_WAVES - 1) / BWD_WAVES;
  if (blocks > NUM_CU) blocks = NUM_CU;
  if (blocks < 1) blocks = 1;
  m.gpw = (m.ngroups + blocks * BWD_WAVES - 1) / (blocks * BWD_WAVES);
  blocks = (m.ngroups + m.gpw * BWD_WAVES - 1) / (m.gpw * BWD_WAVES);
  if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;
  if (blocks > NUM_CU) blocks = NUM_CU;
  return (int)blocks;
}

template <typename S, int N0, int N1, int OP>
int fwd_launch_t(const void* x, const void* core, void* out, const MfmaP& m_in, hipStream_t st) {
  MfmaP m = m_in;
  long long blocks = (m.ngroups + 3) / 4;
  if (blocks > (long long)FWD_BLOCKS_PER_CU * NUM_CU) blocks = (long long)FWD_BLOCKS_PER_CU * NUM_CU;
  m.gpw = (m.ngroups + blocks * 4 - 1) / (blocks * 4);
  blocks = (m.ngroups + m.gpw * 4 - 1) / (m.gpw * 4);   // no idle workgroups at the end
  if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;        // XCD-contiguous mapping needs a multiple of 8
  const bool ovec = m.O == OP && ((uintptr_t)out % (sizeof(S) * OP)) == 0;
  const dim3 g((unsigned)blocks), b(256);
  constexpr int NN = N0 + N1;
  constexpr int RW = NN == 9 ? 3 : 4;   // K*C window rows: 3x3 single channel, 2x2 two channels
  if (m.rowvec_ok && m.vec_ok && ovec && sizeof(S) == 2)
    hipLaunchKernelGGL((eps_fwd_q2reg_k<S, N0, N1, OP, true, true, RW>), g, b, 0, st, (const S*)x,
                       (const S*)core, (S*)out, m);
  else if (m.vec_ok && ovec)
    hipLaunchKernelGGL((eps_fwd_q2reg_k<S, N0, N1, OP, true, true, 0>), g, b, 0, st, (const S*)x,
                       (const S*)core, (S*)out, m);
  else if (m.vec_ok)
    hipLaunchKernelGGL((eps_fwd_q2reg_k<S, N0, N1, OP, true, false, 0>), g, b, 0, st, (const S*)x,
                       (const S*)core, (S*)out, m);
  else
    hipLaunchKernelGGL((eps_fwd_q2reg_k<S, N0, N1, OP, false, false, 0>), g, b, 0, st, (const S*)x,
                       (const S*)core, (S*)out, m);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("eps_fwd_mfma_q2reg");
  return DCTN_OK;
}

template <typename S, int N0, int N1, int OP>
int bwd_launch_t(const void* x, const void* dY, void* dCore, void* ws, const MfmaP& m_in,
                 hipStream_t st) {
  constexpr int A = 1 << N0, BN = 1 << N1, AT = A >= 32 ? A / 32 : 1;
  MfmaP m = m_in;
  const int grid = bwd_grid(m);
  const bool ovec = m.O == OP && ((uintptr_t)dY % (sizeof(S) * OP)) == 0;
  const dim3 g(grid), b(64 * BWD_WAVES);
  constexpr int NN = N0 + N1;
  constexpr int RW = NN == 9 ? 3 : 4;
  if (m.rowvec_ok && m.vec_ok && ovec && sizeof(S) == 2)
    hipLaunchKernelGGL((eps_bwd_dcore_q2reg_k<S, N0, N1, OP, true, true, RW>), g, b, 0, st,
                       (const S*)x, (const S*)dY, (float*)ws, m);
  else if (m.vec_ok && ovec)
    hipLaunchKernelGGL((eps_bwd_dcore_q2reg_k<S, N0, N1, OP, true, true, 0>), g, b, 0, st,
                       (const S*)x, (const S*)dY, (float*)ws, m);
  else if (m.vec_ok)
    hipLaunchKernelGGL((eps_bwd_dcore_q2reg_k<S, N0, N1, OP, true, false, 0>), g, b, 0, st,
                       (const S*)x, (const S*)dY, (float*)ws, m);
  else
    hipLaunchKernelGGL((eps_bwd_dcore_q2reg_k<S, N0, N1, OP, false, false, 0>), g, b, 0, st,
                       (const S*)x, (const S*)dY, (float*)ws, m);
  DCTN_CHECK_LAUNCH();
  if (dctn_main_kernel_only()) return DCTN_OK;
  hipLaunchKernelGGL((eps_bwd_dcore_reduce_k<S>), dim3(BN * OP * AT), dim3(256), 0, st,
                     (const float*)ws, (S*)dCore, grid, A, BN, m.O, OP, AT * 32);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("eps_bwd_mfma_q2reg");
  return DCTN_OK;
}

#define DISPATCH_OP(FN, S, N0, N1, OPV, ...)                         \
  switch (OPV) {                                                     \
    case 2: return FN<S, N0, N1, 2>(__VA_ARGS__);                    \
    case 4: return FN<S, N0, N1, 4>(__VA_ARGS__);                    \
    case 8: return FN<S, N0, N1, 8>(__VA_ARGS__);                    \
    case 16: return FN<S, N0, N1, 16>(__VA_ARGS__);                  \
  }                                                                  \
  return DCTN_ERR_UNSUPPORTED;

template <typename S>
int fwd_dispatch(const void* x, const void* core, void* out, const MfmaP& m, int N, int op,
                 hipStream_t st) {
  if (N == 9) { DISPATCH_OP(fwd_launch_t, S, 5, 4, op, x, core, out, m, st) }
  DISPATCH_OP(fwd_launch_t, S, 4, 4, op, x, core, out, m, st)
}

template <typename S>
int bwd_dispatch(const void* x, const void* dY, void* dCore, void* ws, const MfmaP& m, int N,
                 int op, hipStream_t st) {
  if (N == 9) { DISPATCH_OP(bwd_launch_t, S, 5, 4, op, x, dY, dCore, ws, m, st) }
  DISPATCH_OP(bwd_launch_t, S, 4, 4, op, x, dY, dCore, ws, m, st)
}

}  // namespace

int eps_fwd_mfma(const void* x, const void* core, void* out, const EpsP& p, int dtype,
                 int precision, hipStream_t st) {
  if (!family_ok(p, dtype, precision)) return DCTN_ERR_UNSUPPORTED;
  MfmaP m;
  fill_mp(m, p, x, dtype);
  const int op = next_pow2(p.O) < 2 ? 2 : next_pow2(p.O);
  if (dtype == DCTN_BF16) return fwd_dispatch<bf16_t>(x, core, out, m, p.N, op, st);
  return fwd_dispatch<float>(x, core, out, m, p.N, op, st);
}

size_t eps_bwd_mfma_workspace(const EpsP& p, int dtype, int precision, int need_dx,
                              int need_dcore) {
  (void)need_dx;
  if (!need_dcore || !family_ok(p, dtype, precision)) return 0;
  const int op = next_pow2(p.O) < 2 ? 2 : next_pow2(p.O);
  const int n0 = (p.N + 1) / 2, n1 = p.N - n0;
  const long long A = 1LL << n0, BN = 1LL << n1, acols = A >= 32 ? A : 32;
  return (size_t)NUM_CU * (size_t)(BN * op) * (size_t)acols * sizeof(float);
}

// dCore only; the caller (capi) sends dX to the generic kernels.
int eps_bwd_mfma(const void* x, const void* core, const void* dY, void* dX, void* dCore, void* ws,
                 size_t ws_bytes, const EpsP& p, int dtype, int precision, hipStream_t st) {
  (void)core;
  if (dX || !dCore) return DCTN_ERR_UNSUPPORTED;
  if (!family_ok(p, dtype, precision)) return DCTN_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < eps_bwd_mfma_workspace(p, dtype, precision, 0, 1)) return DCTN_ERR_WORKSPACE;
  MfmaP m;
  fill_mp(m, p, x, dtype);
  const int op = next_pow2(p.O) < 2 ? 2 : next_pow2(p.O);
  if (dtype == DCTN_BF16) return bwd_dispatch<bf16_t>(x, dY, dCore, ws, m, p.N, op, st);
  return bwd_dispatch<float>(x, dY, dCore, ws, m, p.N, op, st);
}
