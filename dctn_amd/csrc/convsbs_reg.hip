// ConvSBS, register-resident sweep for SMALL bonds (every bond <= 4): float32 open chains with at most one two-valued
// core, q^C <= 4 features per core - the 9-core snake of the reference's classifier (mnist.py:189-252) at its default
// bond 2 and at bond 4, BASELINE cfg4 r = 4 (CIFAR colour layout, q = 3).
//
// Replaces dctn/conv_sbs.py:258-304 (ConvSBS.forward) and torch autograd through it for these strings.
//
// At bond 4 a window's whole chain state is ~70 floats and the string's parameters are 1.6 KB: the problem is pure
// latency / launch overhead, not bandwidth or flops ("HBM / launch-latency bound"; 2.5 MB algorithmic per
// forward).  The matrix-core sweep (convsbs_mfma.hip) round-trips forward states and per-window feature gradients
// through HBM (112 MB per backward at the cfg4 shape, 27.6 x the algorithmic bytes) and needs four helper launches
// (zero, gather, two reduction stages).  Here:
//   * lane = window; the chain state v[2][R] and, in the backward, the input state of EVERY core (9 x 2 x 4 floats)
//     live in VGPRs; the forward chain is recomputed in the backward (0.1 GFLOP per launch - noise);
//   * the cores are packed once per workgroup into LDS (zero-padded to R x R, so bonds below R, the bond-1 ends of
//     the chain and absent second output values need no special cases) and read back as broadcasts;
//   * a backward workgroup owns a band of pixel rows of ONE image and computes every window that touches it (the
//     windows of the max_h rows above the band twice: 6.7 % at the cfg4 shape), keeps the per-window feature gradients
//     in LDS and writes the band's dX itself: no per-window gradient tensor in HBM, no zero-fill, no gather launch, a
//     fixed summation order (bit-reproducible);
//   * dCore_c[o,l,r,qq] = sum_w v_c[l] G_c[o,r] f_c[qq]: per-lane products, summed over the 64 lanes of a wave by two
//     register-halving swap levels (v_permlane32_swap / v_permlane16_swap: one swap + one add per pair of entries) and
//     four DPP row rotations, over the waves of the workgroup in LDS, over the workgroups by one small tail kernel
//     (records written entry-major so that it reads them coalesced).  Fixed order everywhere: bit-reproducible.
// Launches per training step: 1 (forward) + 2 (backward, tail) instead of 2 + 5.
#include "common.h"

#include <type_traits>

typedef __attribute__((ext_vector_type(2))) int sr_int2;

// diagnostic build only (make EXTRA=-DDCTN_STAMPS, tools/stamp_sbs_reg.py): wave 0 of every workgroup leaves the
// s_memtime value of each phase boundary behind the dCore records in the workspace (the public workspace query sizes
// it for the matrix-core family too: tens of MB of room)
#ifdef DCTN_STAMPS
#define SR_STAMP(k)                                                                                         \
  do {                                                                                                      \
    if (threadIdx.x == 0 && p.part)                                                                         \
      reinterpret_cast<long long*>(p.part + (((long long)p.coff[SU_NC] * p.nrec + 63) / 64 * 64))[blockIdx.x * 8 + (k)] = \
          (long long)__builtin_readcyclecounter();                                                          \
  } while (0)
#else
#define SR_STAMP(k) do {} while (0)
#endif

namespace {

constexpr int SR_MAXC = 9;        // cores per string (compile-time unrolled backward: 9 = the 3 x 3 snake; shorter strings run too)
constexpr int SR_BWD_THREADS = 512;
constexpr int SR_FWD_THREADS = 256;

struct SrP {
  const float* x;
  long long xs[5];
  const float* core[SR_MAXC];
  int o[SR_MAXC], bl[SR_MAXC], br[SR_MAXC], ph[SR_MAXC], pw[SR_MAXC];
  int n, C, q, B, H, W, Ho, Wo, Otot, max_h;
  long long Wn;
  // backward
  const float* dY;
  float* dX;          // (C, B, H, W, q) contiguous, or NULL
  float* part;        // [padded entry][record] per-workgroup dCore sums, or NULL (no core gradient wanted)
  int band_rows, bands, nrec;
  int coff[SR_MAXC + 1];   // uniform-bond kernels: element offset of core c in the record (natural core layouts back to back)
  // several strings of one layer in one launch (ManyConvSBS, dctn/conv_sbs.py:367-370): every string has its own block
  float* out;              // forward output of this string
  int slot[SR_MAXC];       // core c's pixel = slot-th pixel of the FIRST string (all strings cover the same window positions):
                           // where its feature gradient goes in the per-window LDS row that the dX writer sums
  int tot_all;             // elements of all strings' cores together (coff is absolute inside that record)
  // a string with ONE many-valued core (the ten-label final string of the reference's classifier, mnist.py:213-223)
  int mv, O;               // index of that core (-1: none) and its output count (3..16); every other core has one value
};

constexpr int SR_MAXS = 2;        // strings per launch
struct SrTailP {
  float* dcore[SR_MAXS * SR_MAXC];
  int o[SR_MAXC], bl[SR_MAXC], br[SR_MAXC];
  int n, nrec;
  int coff[SR_MAXS * SR_MAXC + 1];
};

// padded pack: pk[((c * 2 + o) * R + l) * R * QC + r * QC + qq], zero outside the core's real extents
template <int R, int QC>
__device__ __forceinline__ void sr_fill_pack(float* pack, const SrP& p, int ncores, int tid, int nthreads) {
  constexpr int PKC = 2 * R * R * QC;
  for (int e = tid; e < ncores * PKC; e += nthreads) {
    const int c = e / PKC, rem = e - c * PKC;
    const int o = rem / (R * R * QC), r2 = rem - o * (R * R * QC);
    const int l = r2 / (R * QC), r3 = r2 - l * (R * QC);
    const int r = r3 / QC, qq = r3 - r * QC;
    float v = 0.f;
    if (c < p.n && o < p.o[c] && l < p.bl[c] && r < p.br[c]) v = p.core[c][((o * p.bl[c] + l) * p.br[c] + r) * QC + qq];
    pack[e] = v;
  }
}

// f[qq] of core c for the window at (b, ho, wo): product of the pixel's channel values, channel 0 most significant.
// xr (two channels of two values): the raw values x_ch[qv] at xr[ch * 2 + qv]
typedef float sr_v2 __attribute__((ext_vector_type(2)));
typedef float sr_v3 __attribute__((ext_vector_type(3)));
typedef float sr_v4 __attribute__((ext_vector_type(4)));
typedef sr_v2 sr_v2u __attribute__((aligned(4)));   // dword-aligned multi-dword loads (one instruction per pixel)
typedef sr_v3 sr_v3u __attribute__((aligned(4)));
typedef sr_v4 sr_v4u __attribute__((aligned(4)));

template <int Q>
__device__ __forceinline__ void sr_load_pixel(const float* px, long long s4, float* f) {
  if (s4 == 1) {   // the values of a pixel are contiguous (collate_quantum's layout): one load
    if constexpr (Q == 2) { const sr_v2 t = *reinterpret_cast<const sr_v2u*>(px); f[0] = t[0]; f[1] = t[1]; }
    else if constexpr (Q == 3) { const sr_v3 t = *reinterpret_cast<const sr_v3u*>(px); f[0] = t[0]; f[1] = t[1]; f[2] = t[2]; }
    else { const sr_v4 t = *reinterpret_cast<const sr_v4u*>(px); f[0] = t[0]; f[1] = t[1]; f[2] = t[2]; f[3] = t[3]; }
  } else {
#pragma unroll
    for (int qq = 0; qq < Q; ++qq) f[qq] = px[qq * s4];
  }
}

template <int QC, bool TWOCH>
__device__ __forceinline__ void sr_features(const SrP& p, int c, long long b, int ho, int wo, float* f, float* xr) {
  const float* px = p.x + b * p.xs[1] + (long long)(ho + p.ph[c]) * p.xs[2] + (long long)(wo + p.pw[c]) * p.xs[3];
  if constexpr (TWOCH) {
    sr_load_pixel<2>(px, p.xs[4], xr);
    sr_load_pixel<2>(px + p.xs[0], p.xs[4], xr + 2);
    f[0] = xr[0] * xr[2]; f[1] = xr[0] * xr[3]; f[2] = xr[1] * xr[2]; f[3] = xr[1] * xr[3];
  } else {
    sr_load_pixel<QC>(px, p.xs[4], f);
  }
}

// dX of a band of pixel rows from the per-window feature gradients in LDS: a thread takes a pixel (all its q values),
// sums the windows that cover it in core order (fixed order, no atomics) and writes it once.
template <int NCT, int Q>
__device__ __forceinline__ void sr_write_dx_band(const SrP& p, const float* dfl, int ncores, int img, int r0, int r1, int wr0,
                                                 int tid, int nthreads) {
  const int nrows = r1 - r0, Cq = p.C * Q, NCq = ncores * Cq;
  const int npix = p.C * nrows * p.W;
  for (int e = tid; e < npix; e += nthreads) {
    const int ch = e / (nrows * p.W), r2 = e - ch * nrows * p.W;
    const int hr = r2 / p.W, wc = r2 - hr * p.W;
    const int hp = r0 + hr;
    float acc[Q];
#pragma unroll
    for (int qv = 0; qv < Q; ++qv) acc[qv] = 0.f;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      if (c < ncores) {
        const int ho = hp - p.ph[c], wo = wc - p.pw[c];
        if (ho >= 0 && ho < p.Ho && wo >= 0 && wo < p.Wo) {
          const float* d = dfl + (size_t)((ho - wr0) * p.Wo + wo) * NCq + (c * p.C + ch) * Q;
#pragma unroll
          for (int qv = 0; qv < Q; ++qv) acc[qv] += d[qv];
        }
      }
    }
    float* dst = p.dX + ((((long long)ch * p.B + img) * p.H + hp) * p.W + wc) * Q;
#pragma unroll
    for (int qv = 0; qv < Q; ++qv) dst[qv] = acc[qv];
  }
}

// T[l][r] = sum_qq pk[l][r][qq] f[qq]
template <int R, int QC>
__device__ __forceinline__ void sr_tmat(const float* pk, const float* f, float (*T)[R]) {
#pragma unroll
  for (int l = 0; l < R; ++l)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float t = 0.f;
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) t = fmaf(pk[(l * R + r) * QC + qq], f[qq], t);
      T[l][r] = t;
    }
}

// one chain step: v (rows live rows) -> nv; o = 1 keeps the rows, o = 2 (only with one live row) makes two
template <int R, int QC>
__device__ __forceinline__ void sr_step(const float* pkc, int oc, const float* f, const float (*v)[R], float (*nv)[R], int& rows) {
  float T[R][R];
  sr_tmat<R, QC>(pkc, f, T);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float a = 0.f;
#pragma unroll
    for (int l = 0; l < R; ++l) a = fmaf(v[0][l], T[l][r], a);
    nv[0][r] = a;
  }
  if (oc == 2) {
    sr_tmat<R, QC>(pkc + R * R * QC, f, T);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float a = 0.f;
#pragma unroll
      for (int l = 0; l < R; ++l) a = fmaf(v[0][l], T[l][r], a);
      nv[1][r] = a;
    }
    rows = 2;
  } else if (rows == 2) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float a = 0.f;
#pragma unroll
      for (int l = 0; l < R; ++l) a = fmaf(v[1][l], T[l][r], a);
      nv[1][r] = a;
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) nv[1][r] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------ forward
template <int R, int QC, bool TWOCH>
__global__ __launch_bounds__(SR_FWD_THREADS) void convsbs_fwd_reg_k(SrP p, float* __restrict__ out) {
  constexpr int PKC = 2 * R * R * QC;
  __shared__ __align__(16) float pack[SR_MAXC * PKC];
  const int tid = threadIdx.x;
  sr_fill_pack<R, QC>(pack, p, p.n, tid, SR_FWD_THREADS);
  __syncthreads();
  const int hw = p.Ho * p.Wo;
  for (long long w = (long long)blockIdx.x * SR_FWD_THREADS + tid; w < p.Wn; w += (long long)gridDim.x * SR_FWD_THREADS) {
    const long long b = w / hw;
    const int rem = (int)(w - b * hw);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    float v[2][R], nv[2][R];
#pragma unroll
    for (int r = 0; r < R; ++r) { v[0][r] = r == 0 ? 1.f : 0.f; v[1][r] = 0.f; }
    int rows = 1;
#pragma unroll 1
    for (int c = 0; c < p.n; ++c) {
      float f[QC], xr[4];
      sr_features<QC, TWOCH>(p, c, b, ho, wo, f, xr);
      sr_step<R, QC>(pack + c * PKC, p.o[c], f, v, nv, rows);
#pragma unroll
      for (int r = 0; r < R; ++r) { v[0][r] = nv[0][r]; v[1][r] = nv[1][r]; }
    }
    if (p.Otot == 2) {
      float2 o2 = make_float2(v[0][0], v[1][0]);
      *reinterpret_cast<float2*>(out + w * 2) = o2;
    } else {
      out[w] = v[0][0];
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
__device__ __forceinline__ float sr_pair32(float a, float b) {   // lanes 0-31: a summed over (lane, lane + 32); lanes 32-63: b
  const sr_int2 r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float sr_pair16(float a, float b) {   // even rows of 16: a summed over the row pair; odd rows: b
  const sr_int2 r = __builtin_amdgcn_permlane16_swap(__float_as_int(a), __float_as_int(b), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}
// Sum over the 16 lanes of a row, in every lane, of four values at once: 4 x 4 v_add_f32_dpp (the compiler leaves
// `v += mov_dpp(v)` as two instructions; written out, each rotation step is one).  A DPP operand must not have been
// written by the two preceding VALU instructions: the leading s_nop covers the producers, inside the block a value is
// read four instructions after it was written.
__device__ __forceinline__ void sr_row_sum4(float& a, float& b, float& c, float& d) {
  asm volatile(
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %3, %3, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %3, %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// Sum E per-lane values over the 64 lanes of the wave and add them to wacc[0 .. E).  The values are padded to EP (a
// multiple of 16: four row-sum groups of four): after the two halving levels row rho (= lane / 16) holds the entries
// [rho * EP/4, (rho + 1) * EP/4).
template <int E>
__device__ __forceinline__ void sr_wave_reduce_add(const float* prod, float* wacc, int lane, bool first) {
  constexpr int EP = (E + 15) / 16 * 16;
  float t[EP / 2];
#pragma unroll
  for (int j = 0; j < EP / 2; ++j) {
    const float a = j < E ? prod[j] : 0.f, b = j + EP / 2 < E ? prod[j + EP / 2 < E ? j + EP / 2 : 0] : 0.f;
    t[j] = (j < E || j + EP / 2 < E) ? sr_pair32(a, b) : 0.f;
  }
  float u[EP / 4];
#pragma unroll
  for (int j = 0; j < EP / 4; ++j) u[j] = sr_pair16(t[j], t[j + EP / 4]);
#pragma unroll
  for (int j = 0; j < EP / 4; j += 4) sr_row_sum4(u[j], u[j + 1], u[j + 2], u[j + 3]);
  if ((lane & 15) == 0) {
    // this lane's entries: half h = lane / 32 took [h EP/2, (h+1) EP/2) at the first level, row parity the lower / upper
    // quarter of that at the second
    const int e0 = (lane >> 5) * (EP / 2) + ((lane >> 4) & 1) * (EP / 4);
    if (first) {   // the wave's first window group: a plain store (no LDS read to wait for)
#pragma unroll
      for (int j = 0; j < EP / 4; ++j)
        if (e0 + j < E) wacc[e0 + j] = u[j];
    } else {
#pragma unroll
      for (int j = 0; j < EP / 4; ++j)
        if (e0 + j < E) wacc[e0 + j] += u[j];
    }
  }
}

template <int R, int QC, bool TWOCH, int NC>
__global__ __launch_bounds__(SR_BWD_THREADS) void convsbs_bwd_reg_k(SrP p) {
  constexpr int PKC = 2 * R * R * QC, PK = NC * PKC, E = R * R * QC;
  constexpr int NWAVES = SR_BWD_THREADS / 64;
  extern __shared__ __align__(16) float smem[];
  float* pack = smem;                 // [PK]
  float* wacc = pack + PK;            // [NWAVES][PK]: per-wave dCore sums (wave-private: plain read-modify-write)
  float* dfl = wacc + NWAVES * PK;    // [windows of the band][n * C * q]: d/d(pixel values) per window
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cq = p.C * p.q, NCq = p.n * Cq;
  const int img = blockIdx.x / p.bands, band = blockIdx.x - img * p.bands;
  const int r0 = band * p.band_rows, r1 = min(p.H, r0 + p.band_rows);
  const int wr0 = max(0, r0 - p.max_h), wr1 = min(p.Ho, r1);
  const int nwin = wr1 > wr0 ? (wr1 - wr0) * p.Wo : 0;
  sr_fill_pack<R, QC>(pack, p, NC, tid, SR_BWD_THREADS);
  for (int e = tid; e < NWAVES * PK; e += SR_BWD_THREADS) wacc[e] = 0.f;
  __syncthreads();

  for (int base = wave * 64; base < nwin; base += SR_BWD_THREADS) {   // (uniform per wave: the lane sums need every lane)
    const int i = base + lane;
    const bool valid = i < nwin;
    const int ic = valid ? i : nwin - 1;
    const int hrow = ic / p.Wo;
    const int ho = wr0 + hrow, wo = ic - hrow * p.Wo;
    const long long w = ((long long)img * p.Ho + ho) * p.Wo + wo;
    // a window counts towards dCore in the band that holds its top-left pixel (the rows above are a neighbour's)
    const bool owner = valid && ho >= r0 && p.part != nullptr;

    // ---- forward: input state of every core
    float fs[NC][QC], xr[TWOCH ? NC : 1][4];
    float vs[NC + 1][2][R];
    int rows_in[NC + 1];
#pragma unroll
    for (int r = 0; r < R; ++r) { vs[0][0][r] = r == 0 ? 1.f : 0.f; vs[0][1][r] = 0.f; }
    int rows = 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      rows_in[c] = rows;
      if (c < p.n) {
        sr_features<QC, TWOCH>(p, c, img, ho, wo, fs[c], xr[TWOCH ? c : 0]);
        sr_step<R, QC>(pack + c * PKC, p.o[c], fs[c], vs[c], vs[c + 1], rows);
      } else {
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) fs[c][qq] = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) { vs[c + 1][0][r] = vs[c][0][r]; vs[c + 1][1][r] = vs[c][1][r]; }
      }
    }
    rows_in[NC] = rows;

    // ---- adjoint: G = d/d(state after core c), rows as the state has them
    float G[2][R];
#pragma unroll
    for (int r = 0; r < R; ++r) { G[0][r] = 0.f; G[1][r] = 0.f; }
    if (valid) {
      if (p.Otot == 2) {
        const float2 g2 = *reinterpret_cast<const float2*>(p.dY + w * 2);
        G[0][0] = g2.x; G[1][0] = g2.y;
      } else {
        G[0][0] = p.dY[w];
      }
    }
#pragma unroll
    for (int c = NC - 1; c >= 0; --c) {
      if (c >= p.n) continue;
      const float* pkc = pack + c * PKC;
      const int oc = p.o[c];
      float dF[QC], Gin[2][R];
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) dF[qq] = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) { Gin[0][r] = 0.f; Gin[1][r] = 0.f; }
      float fo[QC];   // the features as the dCore products see them: 0 for the windows another band owns
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) fo[qq] = owner ? fs[c][qq] : 0.f;
      // slices of the core: o = 2 -> output rows 0, 1 from input row 0 (slice = output row); o = 1 -> every live row
      // through slice 0.  The slice index is a compile-time constant (register arrays are indexed statically).
      auto slice = [&](auto slc) {
        constexpr int SL = decltype(slc)::value;
        const float* pks = pkc + SL * E;
        float T[R][R], vg[R][R];
        sr_tmat<R, QC>(pks, fs[c], T);
        // vg[l][r] = sum over the rows that pass through this slice of vin[row][l] G[row'][r]
        if (SL == 1 || oc == 2) {
#pragma unroll
          for (int l = 0; l < R; ++l)
#pragma unroll
            for (int r = 0; r < R; ++r) vg[l][r] = vs[c][0][l] * G[SL][r];
#pragma unroll
          for (int l = 0; l < R; ++l) {
            float a = Gin[0][l];
#pragma unroll
            for (int r = 0; r < R; ++r) a = fmaf(T[l][r], G[SL][r], a);
            Gin[0][l] = a;
          }
        } else {
#pragma unroll
          for (int l = 0; l < R; ++l)
#pragma unroll
            for (int r = 0; r < R; ++r) vg[l][r] = vs[c][0][l] * G[0][r];
#pragma unroll
          for (int l = 0; l < R; ++l) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) a = fmaf(T[l][r], G[0][r], a);
            Gin[0][l] = a;
          }
          if (rows_in[c] == 2) {
#pragma unroll
            for (int l = 0; l < R; ++l)
#pragma unroll
              for (int r = 0; r < R; ++r) vg[l][r] = fmaf(vs[c][1][l], G[1][r], vg[l][r]);
#pragma unroll
            for (int l = 0; l < R; ++l) {
              float a = 0.f;
#pragma unroll
              for (int r = 0; r < R; ++r) a = fmaf(T[l][r], G[1][r], a);
              Gin[1][l] = a;
            }
          }
        }
        // d/d(features): dF[qq] += sum_(l,r) vg[l][r] pk[l][r][qq];  dCore products vg[l][r] f[qq]
        float prod[E];
#pragma unroll
        for (int l = 0; l < R; ++l)
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int qq = 0; qq < QC; ++qq) {
              dF[qq] = fmaf(vg[l][r], pks[(l * R + r) * QC + qq], dF[qq]);
              prod[(l * R + r) * QC + qq] = vg[l][r] * fo[qq];
            }
        if (p.part != nullptr) sr_wave_reduce_add<E>(prod, wacc + wave * PK + c * PKC + SL * E, lane, base == wave * 64);
      };
      slice(std::integral_constant<int, 0>{});
      if (oc == 2) slice(std::integral_constant<int, 1>{});
#pragma unroll
      for (int r = 0; r < R; ++r) { G[0][r] = Gin[0][r]; G[1][r] = Gin[1][r]; }
      // d/d(pixel values) of this window and core
      if (valid && p.dX != nullptr) {
        float* d = dfl + (size_t)i * NCq + c * Cq;
        if constexpr (TWOCH) {
          const float* xv = xr[TWOCH ? c : 0];
          d[0] = dF[0] * xv[2] + dF[1] * xv[3];   // d/dx0[0] = sum_q1 dF[0*2+q1] x1[q1]
          d[1] = dF[2] * xv[2] + dF[3] * xv[3];
          d[2] = dF[0] * xv[0] + dF[2] * xv[1];   // d/dx1[0] = sum_q0 dF[q0*2+0] x0[q0]
          d[3] = dF[1] * xv[0] + dF[3] * xv[1];
        } else {
#pragma unroll
          for (int qq = 0; qq < QC; ++qq) d[qq] = dF[qq];
        }
      }
    }
  }
  __syncthreads();

  // ---- dX of the band: every pixel value sums the windows that cover it, in core order (fixed order, no atomics)
  if (p.dX != nullptr) sr_write_dx_band<NC, (TWOCH ? 2 : QC)>(p, dfl, p.n, img, r0, r1, wr0, tid, SR_BWD_THREADS);
  // ---- this workgroup's dCore record (entry-major: the tail kernel reads a row of records coalesced)
  if (p.part != nullptr) {
    for (int e = tid; e < PK; e += SR_BWD_THREADS) {
      float s = 0.f;
#pragma unroll
      for (int wv = 0; wv < NWAVES; ++wv) s += wacc[wv * PK + e];
      p.part[(long long)e * p.nrec + blockIdx.x] = s;
    }
  }
}

// dCore[c][o][l][r][qq] = sum over the records: one wave per padded entry, lanes stride the records
template <int R, int QC>
__global__ __launch_bounds__(256) void convsbs_reg_tail_k(const float* __restrict__ part, SrTailP t) {
  constexpr int PKC = 2 * R * R * QC;
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= t.n * PKC) return;
  const int c = e / PKC, rem = e - c * PKC;
  const int o = rem / (R * R * QC), r2 = rem - o * (R * R * QC);
  const int l = r2 / (R * QC), r3 = r2 - l * (R * QC);
  const int r = r3 / QC, qq = r3 - r * QC;
  if (o >= t.o[c] || l >= t.bl[c] || r >= t.br[c]) return;   // padding (wave-uniform)
  float s = 0.f;
  for (int k = lane; k < t.nrec; k += 64) s += part[(long long)e * t.nrec + k];
  s = wave_reduce_sum(s);
  if (lane == 0) t.dcore[c][((o * t.bl[c] + l) * t.br[c] + r) * QC + qq] = s;
}


// ================================================================================================ uniform strings
// The common case - nine cores, bonds (1, R, ..., R), i.e. every string `ManyConvSBS` builds without trace_edge - gets
// kernels of its own.  The padded LDS pack above costs ~20 broadcast reads per core and window group, each followed by
// a wait that two waves per SIMD cannot hide (rocprofv3 at the cfg4 shape: backward 28.9 us for ~17 us of vector work).
// With all shapes known at compile time the coefficients are read where they lie, in the cores' natural [o][l][r][qq]
// layout, by SCALAR loads (uniform addresses, compile-time offsets: s_load_dwordx16) and enter the multiply-adds as
// SGPR operands: no LDS traffic for them at all, and the first / last core (one bond leg of size 1) do a quarter of a
// middle core's work.  The per-wave dCore sums live in natural layout too, so the record IS the flat gradient.
constexpr int SU_NC = 9;
// a core seen through the CONSTANT address space: uniform loads from it are scalar loads whatever the compiler can prove
// about aliasing stores (through a plain global pointer it issued one vector load per coefficient quad and lane)
typedef const __attribute__((address_space(4))) float* su_kptr;
// (the empty asm makes the pointer opaque at its point of use: otherwise the loads - invariant in the window loop - are
// hoisted out of it, all 400+ coefficients at once, and live in spilled SGPRs: ~300 v_readlane per window in the forward)
__device__ __forceinline__ su_kptr su_k(const float* g) {
  uintptr_t u = (uintptr_t)g;
  asm volatile("" : "+s"(u));
  return (su_kptr)u;
}

template <int R, int QC>
struct SuShape {
  static constexpr int E_END = R * QC;        // entries of one output slice of the first / last core
  static constexpr int E_MID = R * R * QC;    // ... of a middle core
};

// state after the first core: v[row][r] = sum_qq core0[row][0][r][qq] f[qq]
template <int R, int QC>
__device__ __forceinline__ void su_first(su_kptr k0, int o0, const float* f, float (*v)[R]) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float a = 0.f;
#pragma unroll
    for (int qq = 0; qq < QC; ++qq) a = fmaf(k0[r * QC + qq], f[qq], a);
    v[0][r] = a;
    v[1][r] = 0.f;
  }
  if (o0 == 2) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float b = 0.f;
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) b = fmaf(k0[(R + r) * QC + qq], f[qq], b);
      v[1][r] = b;
    }
  }
}

// a middle core: kc = its [o][R][R][QC] coefficients (scalar loads)
template <int R, int QC>
__device__ __forceinline__ void su_tmat(su_kptr ks, const float* f, float (*T)[R]) {
#pragma unroll
  for (int l = 0; l < R; ++l)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float t = 0.f;
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) t = fmaf(ks[(l * R + r) * QC + qq], f[qq], t);
      T[l][r] = t;
    }
}

template <int R, int QC>
__device__ __forceinline__ void su_mid(su_kptr kc, int oc, const float* f, const float (*v)[R], float (*nv)[R],
                                       int& rows) {
  float T[R][R];
  su_tmat<R, QC>(kc, f, T);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float a = 0.f;
#pragma unroll
    for (int l = 0; l < R; ++l) a = fmaf(v[0][l], T[l][r], a);
    nv[0][r] = a;
  }
  if (oc == 2) {
    su_tmat<R, QC>(kc + R * R * QC, f, T);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float a = 0.f;
#pragma unroll
      for (int l = 0; l < R; ++l) a = fmaf(v[0][l], T[l][r], a);
      nv[1][r] = a;
    }
    rows = 2;
  } else if (rows == 2) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float a = 0.f;
#pragma unroll
      for (int l = 0; l < R; ++l) a = fmaf(v[1][l], T[l][r], a);
      nv[1][r] = a;
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) nv[1][r] = 0.f;
  }
}

// the last core: t[sl][l] = sum_qq coreN[sl][l][0][qq] f[qq]; out[row] = sum_l v[row or 0][l] t[.][l]
template <int R, int QC>
__device__ __forceinline__ void su_last(su_kptr kl, int ol, const float* f, const float (*v)[R], int rows, float* out2) {
  float t0[R];
#pragma unroll
  for (int l = 0; l < R; ++l) {
    float a = 0.f;
#pragma unroll
    for (int qq = 0; qq < QC; ++qq) a = fmaf(kl[l * QC + qq], f[qq], a);
    t0[l] = a;
  }
  float o0 = 0.f, o1 = 0.f;
#pragma unroll
  for (int l = 0; l < R; ++l) o0 = fmaf(v[0][l], t0[l], o0);
  if (ol == 2) {
#pragma unroll
    for (int l = 0; l < R; ++l) {
      float b = 0.f;
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) b = fmaf(kl[(R + l) * QC + qq], f[qq], b);
      o1 = fmaf(v[0][l], b, o1);
    }
  } else if (rows == 2) {
#pragma unroll
    for (int l = 0; l < R; ++l) o1 = fmaf(v[1][l], t0[l], o1);
  }
  out2[0] = o0;
  out2[1] = o1;
}

template <int R, int QC, bool TWOCH, bool MULTI>
__global__ __launch_bounds__(SR_FWD_THREADS) void convsbs_fwd_regu_k(SrP pa, SrP pb) {
  const SrP& p = (MULTI && blockIdx.y) ? pb : pa;   // one string per grid.y slice (independent outputs)
  float* __restrict__ out = p.out;
  const int tid = threadIdx.x;
  const int hw = p.Ho * p.Wo;
  for (long long w = (long long)blockIdx.x * SR_FWD_THREADS + tid; w < p.Wn; w += (long long)gridDim.x * SR_FWD_THREADS) {
    const long long b = w / hw;
    const int rem = (int)(w - b * hw);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    float fs[SU_NC][QC], xr[4];
#pragma unroll
    for (int c = 0; c < SU_NC; ++c) sr_features<QC, TWOCH>(p, c, b, ho, wo, fs[c], xr);   // every load in flight before the chain
    float v[2][R], nv[2][R];
    su_first<R, QC>(su_k(p.core[0]), p.o[0], fs[0], v);
    int rows = p.o[0];
#pragma unroll
    for (int c = 1; c < SU_NC - 1; ++c) {
      __builtin_amdgcn_sched_barrier(0);   // a core's coefficients are loaded when its turn comes (48+ SGPRs each: all nine do not fit)
      su_mid<R, QC>(su_k(p.core[c]), p.o[c], fs[c], v, nv, rows);
#pragma unroll
      for (int r = 0; r < R; ++r) { v[0][r] = nv[0][r]; v[1][r] = nv[1][r]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    float o2[2];
    su_last<R, QC>(su_k(p.core[SU_NC - 1]), p.o[SU_NC - 1], fs[SU_NC - 1], v, rows, o2);
    if (p.Otot == 2) *reinterpret_cast<float2*>(out + w * 2) = make_float2(o2[0], o2[1]);
    else out[w] = o2[0];
  }
}

// MULTI: several strings per launch (the loop over strings reads each string's block through a run-time pointer; the
// single-string instantiation keeps its block in SGPRs as before: 24.4 against 28.9 us at the cfg4 shape)
template <int R, int QC, bool TWOCH, bool MULTI>
__global__ __launch_bounds__(SR_BWD_THREADS) void convsbs_bwd_regu_k(SrP pa, SrP pb, int nstr) {
  constexpr int NC = SU_NC, EE = R * QC, EM = R * R * QC;
  constexpr int NWAVES = SR_BWD_THREADS / 64;
  extern __shared__ __align__(16) float smem[];
  const SrP& p = pa;                  // geometry, x, dX, records: common to the strings
  const int tot = pa.tot_all;
  float* wacc = smem;                 // [NWAVES][tot]: per-wave dCore sums, natural layout
  float* dfl = wacc + NWAVES * tot;   // [windows of the band][NC * C * q]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cq = p.C * p.q, NCq = NC * Cq;
  const int img = blockIdx.x / p.bands, band = blockIdx.x - img * p.bands;
  const int r0 = band * p.band_rows, r1 = min(p.H, r0 + p.band_rows);
  const int wr0 = max(0, r0 - p.max_h), wr1 = min(p.Ho, r1);
  const int nwin = wr1 > wr0 ? (wr1 - wr0) * p.Wo : 0;
  const bool want_dcore = p.part != nullptr;
  SR_STAMP(0);
  for (int e = tid; e < NWAVES * tot; e += SR_BWD_THREADS) wacc[e] = 0.f;
  __syncthreads();
  SR_STAMP(1);
  float* wa = wacc + wave * tot;

  for (int base = wave * 64; base < nwin; base += SR_BWD_THREADS) {
    const int i = base + lane;
    const bool valid = i < nwin;
    const int ic = valid ? i : nwin - 1;
    const int hrow = ic / p.Wo;
    const int ho = wr0 + hrow, wo = ic - hrow * p.Wo;
    const long long w = ((long long)img * p.Ho + ho) * p.Wo + wo;
    const bool owner = valid && ho >= r0 && want_dcore;
    const bool first = base == wave * 64;
#pragma unroll 1
    for (int si = 0; si < (MULTI ? nstr : 1); ++si) {
    const SrP& p = (MULTI && si) ? pb : pa;      // this string's cores, positions, dY (shadows the common block)

    // ---- features of every core (all loads in flight together), dY
    float fs[NC][QC], xr[TWOCH ? NC : 1][4];
#pragma unroll
    for (int c = 0; c < NC; ++c) sr_features<QC, TWOCH>(p, c, img, ho, wo, fs[c], xr[TWOCH ? c : 0]);
    float gy[2] = {0.f, 0.f};
    if (valid) {
      if (p.Otot == 2) {
        const float2 g2 = *reinterpret_cast<const float2*>(p.dY + w * 2);
        gy[0] = g2.x; gy[1] = g2.y;
      } else {
        gy[0] = p.dY[w];
      }
    }
    // ---- forward: vs[c] = input state of core c (c = 1 .. NC-1)
    float vs[NC][2][R];
    int rows_in[NC];
    su_first<R, QC>(su_k(p.core[0]), p.o[0], fs[0], vs[1]);
    int rows = p.o[0];
#pragma unroll
    for (int c = 1; c < NC - 1; ++c) {
      rows_in[c] = rows;
      __builtin_amdgcn_sched_barrier(0);   // (coefficients of one core at a time in SGPRs)
      su_mid<R, QC>(su_k(p.core[c]), p.o[c], fs[c], vs[c], vs[c + 1], rows);
    }
    __builtin_amdgcn_sched_barrier(0);
    rows_in[NC - 1] = rows;
    SR_STAMP(2);

    auto store_df = [&](int c, const float* dF) {
      if (valid && p.dX != nullptr) {
        float* d = dfl + (size_t)i * NCq + (MULTI ? p.slot[c] : c) * Cq;   // the first string stores, the others add (same lane: no race)
        float g[TWOCH ? 4 : QC];
        if constexpr (TWOCH) {
          const float* xv = xr[TWOCH ? c : 0];
          g[0] = dF[0] * xv[2] + dF[1] * xv[3];
          g[1] = dF[2] * xv[2] + dF[3] * xv[3];
          g[2] = dF[0] * xv[0] + dF[2] * xv[1];
          g[3] = dF[1] * xv[0] + dF[3] * xv[1];
        } else {
#pragma unroll
          for (int qq = 0; qq < QC; ++qq) g[qq] = dF[qq];
        }
        if (!MULTI || si == 0) {
#pragma unroll
          for (int qq = 0; qq < (TWOCH ? 4 : QC); ++qq) d[qq] = g[qq];
        } else {
#pragma unroll
          for (int qq = 0; qq < (TWOCH ? 4 : QC); ++qq) d[qq] += g[qq];
        }
      }
    };

    // ---- last core: out[row] = sum_l vin[row or 0][l] t[sl][l]
    float G[2][R];
    {
      constexpr int c = NC - 1;
      const su_kptr kl = su_k(p.core[c]);
      const int ol = p.o[c];
      float fo[QC], dF[QC];
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) { fo[qq] = owner ? fs[c][qq] : 0.f; dF[qq] = 0.f; }
#pragma unroll
      for (int l = 0; l < R; ++l) { G[0][l] = 0.f; G[1][l] = 0.f; }
      auto slice = [&](auto slc) {
        constexpr int SL = decltype(slc)::value;
        const su_kptr ks = kl + SL * EE;
        float vg[R];
        // o = 2: slice SL reads input row 0 and makes output SL; o = 1: slice 0 carries every live row
#pragma unroll
        for (int l = 0; l < R; ++l) {
          float tl = 0.f;
#pragma unroll
          for (int qq = 0; qq < QC; ++qq) tl = fmaf(ks[l * QC + qq], fs[c][qq], tl);
          if (SL == 1 || ol == 2) {
            vg[l] = vs[c][0][l] * gy[SL];
            G[0][l] = fmaf(tl, gy[SL], G[0][l]);
          } else {
            vg[l] = vs[c][0][l] * gy[0];
            G[0][l] = tl * gy[0];
            if (rows_in[c] == 2) {
              vg[l] = fmaf(vs[c][1][l], gy[1], vg[l]);
              G[1][l] = tl * gy[1];
            }
          }
        }
        float prod[EE];
#pragma unroll
        for (int l = 0; l < R; ++l)
#pragma unroll
          for (int qq = 0; qq < QC; ++qq) {
            dF[qq] = fmaf(vg[l], ks[l * QC + qq], dF[qq]);
            prod[l * QC + qq] = vg[l] * fo[qq];
          }
        if (want_dcore) sr_wave_reduce_add<EE>(prod, wa + p.coff[c] + SL * EE, lane, first);
      };
      slice(std::integral_constant<int, 0>{});
      if (ol == 2) slice(std::integral_constant<int, 1>{});
      store_df(c, dF);
    }
    // ---- middle cores, right to left
#pragma unroll
    for (int c = NC - 2; c >= 1; --c) {
      __builtin_amdgcn_sched_barrier(0);
      const su_kptr kc = su_k(p.core[c]);
      const int oc = p.o[c];
      float fo[QC], dF[QC], Gin[2][R];
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) { fo[qq] = owner ? fs[c][qq] : 0.f; dF[qq] = 0.f; }
#pragma unroll
      for (int r = 0; r < R; ++r) { Gin[0][r] = 0.f; Gin[1][r] = 0.f; }
      auto slice = [&](auto slc) {
        constexpr int SL = decltype(slc)::value;
        const su_kptr ks = kc + SL * EM;
        float T[R][R], vg[R][R];
        su_tmat<R, QC>(ks, fs[c], T);
        if (SL == 1 || oc == 2) {
#pragma unroll
          for (int l = 0; l < R; ++l) {
            float a = Gin[0][l];
#pragma unroll
            for (int r = 0; r < R; ++r) {
              vg[l][r] = vs[c][0][l] * G[SL][r];
              a = fmaf(T[l][r], G[SL][r], a);
            }
            Gin[0][l] = a;
          }
        } else {
#pragma unroll
          for (int l = 0; l < R; ++l) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
              vg[l][r] = vs[c][0][l] * G[0][r];
              a = fmaf(T[l][r], G[0][r], a);
            }
            Gin[0][l] = a;
          }
          if (rows_in[c] == 2) {
#pragma unroll
            for (int l = 0; l < R; ++l) {
              float a = 0.f;
#pragma unroll
              for (int r = 0; r < R; ++r) {
                vg[l][r] = fmaf(vs[c][1][l], G[1][r], vg[l][r]);
                a = fmaf(T[l][r], G[1][r], a);
              }
              Gin[1][l] = a;
            }
          }
        }
        float prod[EM];
#pragma unroll
        for (int l = 0; l < R; ++l)
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int qq = 0; qq < QC; ++qq) {
              dF[qq] = fmaf(vg[l][r], ks[(l * R + r) * QC + qq], dF[qq]);
              prod[(l * R + r) * QC + qq] = vg[l][r] * fo[qq];
            }
        if (want_dcore) sr_wave_reduce_add<EM>(prod, wa + p.coff[c] + SL * EM, lane, first);
      };
      slice(std::integral_constant<int, 0>{});
      if (oc == 2) slice(std::integral_constant<int, 1>{});
#pragma unroll
      for (int r = 0; r < R; ++r) { G[0][r] = Gin[0][r]; G[1][r] = Gin[1][r]; }
      store_df(c, dF);
    }
    // ---- first core: its input state is the constant (1, 0, ..): vg[r] = G[slice or row 0][r]
    __builtin_amdgcn_sched_barrier(0);
    {
      const su_kptr k0 = su_k(p.core[0]);
      const int o0 = p.o[0];
      float fo[QC], dF[QC];
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) { fo[qq] = owner ? fs[0][qq] : 0.f; dF[qq] = 0.f; }
      auto slice = [&](auto slc) {
        constexpr int SL = decltype(slc)::value;
        const su_kptr ks = k0 + SL * EE;
        float prod[EE];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int qq = 0; qq < QC; ++qq) {
            dF[qq] = fmaf(G[SL][r], ks[r * QC + qq], dF[qq]);
            prod[r * QC + qq] = G[SL][r] * fo[qq];
          }
        if (want_dcore) sr_wave_reduce_add<EE>(prod, wa + p.coff[0] + SL * EE, lane, first);
      };
      slice(std::integral_constant<int, 0>{});
      if (o0 == 2) slice(std::integral_constant<int, 1>{});
      store_df(0, dF);
    }
    }   // strings
  }
  SR_STAMP(3);
  __syncthreads();
  SR_STAMP(4);

  // ---- dX of the band (fixed summation order)
  if (p.dX != nullptr) sr_write_dx_band<NC, (TWOCH ? 2 : QC)>(p, dfl, NC, img, r0, r1, wr0, tid, SR_BWD_THREADS);
  SR_STAMP(5);
  // ---- this workgroup's record: the flat dCore, entry-major
  if (want_dcore) {
    for (int e = tid; e < tot; e += SR_BWD_THREADS) {
      float s2 = 0.f;
#pragma unroll
      for (int wv = 0; wv < NWAVES; ++wv) s2 += wacc[wv * tot + e];
      p.part[(long long)e * p.nrec + blockIdx.x] = s2;
    }
  }
  SR_STAMP(6);
}

// uniform strings: dCore[c][local] = sum over the records of entry coff[c] + local
__global__ __launch_bounds__(256) void convsbs_regu_tail_k(const float* __restrict__ part, SrTailP t, int tot) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= tot) return;
  float s = 0.f;
  for (int k = lane; k < t.nrec; k += 64) s += part[(long long)e * t.nrec + k];
  s = wave_reduce_sum(s);
  if (lane == 0) {
    int c = 0;
    while (c + 1 < t.n && e >= t.coff[c + 1]) ++c;
    t.dcore[c][e - t.coff[c]] = s;
  }
}

// ================================================================================================ one many-valued core
// The final string of the reference's ConvSBS classifier (mnist.py:213-223): nine cores, the centre one with NUM_LABELS = 10
// outputs.  Every output shares the prefix state v (cores before m) and the SUFFIX vector s (cores behind m, applied right
// to left: s_c = T_c s_(c+1)), so
//     out[o] = sum_(l,r,qq) core_m[o][l][r][qq] v[l] s[r] f_m[qq]
// is one short dot product per output and window, and the way back is
//     dz[l][r][qq] = sum_o dY[o] core_m[o][l][r][qq],   dCore_m[o] += dY[o] (v x s x f)   (lane sums as for the other cores),
// then dv, ds, df_m from dz and ONE adjoint pass through the prefix and through the suffix - the work of a one-valued
// string plus O dot products, where the matrix-core sweep ran the whole chain once per pair of outputs (five launches
// each way, 0.39 ms of the classifier's 0.57 ms step).
// LDS pack: cores c != m as one [R][R][QC] slice at c * E, the many-valued core's O slices behind them at (NC + o) * E.
template <int R, int QC>
__device__ __forceinline__ void sm_fill_pack(float* pack, const SrP& p, int ncores, int tid, int nthreads) {
  constexpr int E = R * R * QC;
  for (int e = tid; e < (ncores + p.O) * E; e += nthreads) {
    const int sl = e / E, rem = e - sl * E;
    const int c = sl < ncores ? sl : p.mv, o = sl < ncores ? 0 : sl - ncores;
    const int l = rem / (R * QC), r3 = rem - l * (R * QC);
    const int r = r3 / QC, qq = r3 - r * QC;
    float v = 0.f;
    if (c < p.n && (sl >= ncores || c != p.mv) && l < p.bl[c] && r < p.br[c]) v = p.core[c][((o * p.bl[c] + l) * p.br[c] + r) * QC + qq];
    pack[e] = v;
  }
}

// prefix state and suffix vector of a window: v = e0 T_0 .. T_(m-1),  s = T_(m+1) .. T_(n-1) e0
template <int R, int QC, int NC>
__device__ __forceinline__ void sm_prefix_suffix(const float* pack, const SrP& p, const float (*f)[QC], float (*vs)[R], float (*ss)[R]) {
  constexpr int E = R * R * QC;
  // vs[c] = the state entering core c (c <= m); ss[c] = the vector leaving core c - 1 to the right, i.e. s_c (c > m)
#pragma unroll
  for (int r = 0; r < R; ++r) { vs[0][r] = r == 0 ? 1.f : 0.f; ss[NC][r] = r == 0 ? 1.f : 0.f; }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c < p.mv) {
      float T[R][R];
      sr_tmat<R, QC>(pack + c * E, f[c], T);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float a = 0.f;
#pragma unroll
        for (int l = 0; l < R; ++l) a = fmaf(vs[c][l], T[l][r], a);
        vs[c + 1][r] = a;
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) vs[c + 1][r] = vs[c][r];
    }
  }
#pragma unroll
  for (int c = NC - 1; c >= 0; --c) {
    if (c > p.mv && c < p.n) {
      float T[R][R];
      sr_tmat<R, QC>(pack + c * E, f[c], T);
#pragma unroll
      for (int l = 0; l < R; ++l) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) a = fmaf(T[l][r], ss[c + 1][r], a);
        ss[c][l] = a;
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) ss[c][r] = ss[c + 1][r];
    }
  }
}

// The features of every core of a window, all loads issued before the first use (round 5): core by core through
// sr_features, the two-channel products made every core's loads a round trip of their own - a branch on the pixel layout
// around each load, the product right behind it - nine dependent round trips per window in front of the chain.  Cores
// beyond n re-read the last one; `raw` keeps the pixel values of the two-channel mode (the way back needs them).
template <int QC, bool TWOCH, int NC>
__device__ __forceinline__ void sm_features_all(const SrP& p, long long b, int ho, int wo, float (*f)[QC], float (*raw)[4]) {
  const float* win = p.x + b * p.xs[1] + (long long)ho * p.xs[2] + (long long)wo * p.xs[3];
  if constexpr (TWOCH) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int cc = c < p.n ? c : p.n - 1;
      const float* px = win + (long long)p.ph[cc] * p.xs[2] + (long long)p.pw[cc] * p.xs[3];
      raw[c][0] = px[0];
      raw[c][1] = px[p.xs[4]];
      raw[c][2] = px[p.xs[0]];
      raw[c][3] = px[p.xs[0] + p.xs[4]];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bool in = c < p.n;
      f[c][0] = in ? raw[c][0] * raw[c][2] : 0.f;
      f[c][1] = in ? raw[c][0] * raw[c][3] : 0.f;
      f[c][2] = in ? raw[c][1] * raw[c][2] : 0.f;
      f[c][3] = in ? raw[c][1] * raw[c][3] : 0.f;
    }
  } else {
    float ld[NC][QC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int cc = c < p.n ? c : p.n - 1;
      const float* px = win + (long long)p.ph[cc] * p.xs[2] + (long long)p.pw[cc] * p.xs[3];
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) ld[c][qq] = px[qq * p.xs[4]];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) f[c][qq] = c < p.n ? ld[c][qq] : 0.f;
  }
}

template <int R, int QC, bool TWOCH>
__global__ __launch_bounds__(SR_FWD_THREADS) void convsbs_fwd_regmv_k(SrP p, float* __restrict__ out) {
  constexpr int NC = SR_MAXC, E = R * R * QC;
  extern __shared__ __align__(16) float smem[];
  float* pack = smem;
  const int tid = threadIdx.x;
  sm_fill_pack<R, QC>(pack, p, NC, tid, SR_FWD_THREADS);
  __syncthreads();
  const int hw = p.Ho * p.Wo;
  for (long long w = (long long)blockIdx.x * SR_FWD_THREADS + tid; w < p.Wn; w += (long long)gridDim.x * SR_FWD_THREADS) {
    const long long b = w / hw;
    const int rem = (int)(w - b * hw);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    float f[NC][QC];
    {
      float raw[TWOCH ? NC : 1][4];
      sm_features_all<QC, TWOCH, NC>(p, b, ho, wo, f, raw);
    }
    float vs[NC + 1][R], ss[NC + 1][R];
    sm_prefix_suffix<R, QC, NC>(pack, p, f, vs, ss);
    float fm[QC];
#pragma unroll
    for (int qq = 0; qq < QC; ++qq) fm[qq] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (c == p.mv)
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) fm[qq] = f[c][qq];
    float z[E];   // v[l] s[r] f_m[qq]
#pragma unroll
    for (int l = 0; l < R; ++l)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float vsr = vs[NC][l] * ss[0][r];   // (vs is constant from m on, ss from m + 1 down: the ends hold v_m and s_(m+1))
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) z[(l * R + r) * QC + qq] = vsr * fm[qq];
      }
    float* op = out + w * p.O;
#pragma unroll 1
    for (int o = 0; o < p.O; ++o) {
      const float* pk = pack + (NC + o) * E;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int e = 0; e < E; e += 2) { a0 = fmaf(pk[e], z[e], a0); a1 = fmaf(pk[e + 1], z[e + 1], a1); }
      op[o] = a0 + a1;
    }
  }
}

template <int R, int QC, bool TWOCH, int NC>
__global__ __launch_bounds__(SR_BWD_THREADS) void convsbs_bwd_regmv_k(SrP p) {
  constexpr int E = R * R * QC;
  constexpr int NWAVES = SR_BWD_THREADS / 64;
  extern __shared__ __align__(16) float smem[];
  const int PK = (NC + p.O) * E;
  float* pack = smem;                 // [PK]
  float* wacc = pack + PK;            // [NWAVES][PK]: per-wave dCore sums
  float* dfl = wacc + NWAVES * PK;    // [windows of the band][n * C * q]: d/d(pixel values) per window
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cq = p.C * p.q, NCq = p.n * Cq;
  const int img = blockIdx.x / p.bands, band = blockIdx.x - img * p.bands;
  const int r0 = band * p.band_rows, r1 = min(p.H, r0 + p.band_rows);
  const int wr0 = max(0, r0 - p.max_h), wr1 = min(p.Ho, r1);
  const int nwin = wr1 > wr0 ? (wr1 - wr0) * p.Wo : 0;
  sm_fill_pack<R, QC>(pack, p, NC, tid, SR_BWD_THREADS);
  for (int e = tid; e < NWAVES * PK; e += SR_BWD_THREADS) wacc[e] = 0.f;
  __syncthreads();
  float* wmine = wacc + wave * PK;

  for (int base = wave * 64; base < nwin; base += SR_BWD_THREADS) {   // (uniform per wave: the lane sums need every lane)
    const int i = base + lane;
    const bool valid = i < nwin;
    const int ic = valid ? i : nwin - 1;
    const int hrow = ic / p.Wo;
    const int ho = wr0 + hrow, wo = ic - hrow * p.Wo;
    const long long w = ((long long)img * p.Ho + ho) * p.Wo + wo;
    const bool owner = valid && ho >= r0 && p.part != nullptr;   // dCore counts a window in the band of its top-left pixel
    const bool first = base == wave * 64;

    float f[NC][QC];
    {
      float raw[TWOCH ? NC : 1][4];
      sm_features_all<QC, TWOCH, NC>(p, img, ho, wo, f, raw);
      // two channels: the way back turns d/d(products) into d/d(pixel values) and needs the pixel values again - they wait
      // in the window's own slots of `dfl`, which the way back overwrites with the gradients (re-read from memory core by
      // core they were nine more round trips per window)
      if constexpr (TWOCH) {
        if (valid && p.dX != nullptr) {
          float* d = dfl + (size_t)i * NCq;
#pragma unroll
          for (int c = 0; c < NC; ++c)
            if (c < p.n)
#pragma unroll
              for (int k = 0; k < 4; ++k) d[c * 4 + k] = raw[c][k];
        }
      }
    }
    float vs[NC + 1][R], ss[NC + 1][R];
    sm_prefix_suffix<R, QC, NC>(pack, p, f, vs, ss);
    float fm[QC];
#pragma unroll
    for (int qq = 0; qq < QC; ++qq) fm[qq] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (c == p.mv)
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) fm[qq] = f[c][qq];
    float vsr[R][R];
#pragma unroll
    for (int l = 0; l < R; ++l)
#pragma unroll
      for (int r = 0; r < R; ++r) vsr[l][r] = vs[NC][l] * ss[0][r];

    // d/d(pixel values) of this window and core c from d/d(feature products)
    auto put_row = [&](int c, const float (&dF)[QC]) {
      if (!valid || p.dX == nullptr) return;
      float* d = dfl + (size_t)i * NCq + c * Cq;
      if constexpr (TWOCH) {
        const float xv[4] = {d[0], d[1], d[2], d[3]};   // the pixel values left here at the window's start
        d[0] = dF[0] * xv[2] + dF[1] * xv[3];
        d[1] = dF[2] * xv[2] + dF[3] * xv[3];
        d[2] = dF[0] * xv[0] + dF[2] * xv[1];
        d[3] = dF[1] * xv[0] + dF[3] * xv[1];
      } else {
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) d[qq] = dF[qq];
      }
    };

    // ---- the many-valued core, pass A: dCore_m[o] += dY[o] (v x s x f) - O lane sums of E products
    const float* dyp = p.dY + w * p.O;
    if (p.part != nullptr) {
      float gnext = dyp[0];   // (the next label's dY is in flight while this one's 64 lane sums run: a load per turn was a round trip per label)
#pragma unroll 1
      for (int o = 0; o < p.O; ++o) {
        const float g = owner ? gnext : 0.f;
        gnext = dyp[o + 1 < p.O ? o + 1 : o];
        float prod[E];
#pragma unroll
        for (int l = 0; l < R; ++l)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float gv = g * vsr[l][r];
#pragma unroll
            for (int qq = 0; qq < QC; ++qq) prod[(l * R + r) * QC + qq] = gv * fm[qq];
          }
        sr_wave_reduce_add<E>(prod, wmine + (NC + o) * E, lane, first);
      }
    }
    // ---- pass B: dz = sum_o dY[o] core_m[o], then its three contractions
    float dv[R], ds[R], dFm[QC];
    {
      float dz[E];
#pragma unroll
      for (int e = 0; e < E; ++e) dz[e] = 0.f;
      float gnext = dyp[0];
#pragma unroll 1
      for (int o = 0; o < p.O; ++o) {
        const float g = valid ? gnext : 0.f;
        gnext = dyp[o + 1 < p.O ? o + 1 : o];
        const float* pk = pack + (NC + o) * E;
#pragma unroll
        for (int e = 0; e < E; ++e) dz[e] = fmaf(g, pk[e], dz[e]);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) { dv[r] = 0.f; ds[r] = 0.f; }
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) dFm[qq] = 0.f;
#pragma unroll
      for (int l = 0; l < R; ++l)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float zf = 0.f;   // sum_qq dz f_m[qq]
#pragma unroll
          for (int qq = 0; qq < QC; ++qq) {
            zf = fmaf(dz[(l * R + r) * QC + qq], fm[qq], zf);
            dFm[qq] = fmaf(dz[(l * R + r) * QC + qq], vsr[l][r], dFm[qq]);
          }
          dv[l] = fmaf(zf, ss[0][r], dv[l]);
          ds[r] = fmaf(zf, vs[NC][l], ds[r]);
        }
    }
    // ---- the cores behind m, left to right: s_c = T_c s_(c+1);  d/ds_c is known -> dT_c = ds_c x s_(c+1), ds_(c+1) = T_c^T ds_c
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c == p.mv) put_row(c, dFm);
      if (c > p.mv && c < p.n) {
        const float* pkc = pack + c * E;
        float T[R][R], dF[QC], nds[R], prod[E];
        sr_tmat<R, QC>(pkc, f[c], T);
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) dF[qq] = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) nds[r] = 0.f;
#pragma unroll
        for (int l = 0; l < R; ++l)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float dT = ds[l] * ss[c + 1][r];
            nds[r] = fmaf(T[l][r], ds[l], nds[r]);
#pragma unroll
            for (int qq = 0; qq < QC; ++qq) {
              dF[qq] = fmaf(dT, pkc[(l * R + r) * QC + qq], dF[qq]);
              prod[(l * R + r) * QC + qq] = owner ? dT * f[c][qq] : 0.f;
            }
          }
        if (p.part != nullptr) sr_wave_reduce_add<E>(prod, wmine + c * E, lane, first);
#pragma unroll
        for (int r = 0; r < R; ++r) ds[r] = nds[r];
        put_row(c, dF);
      }
    }
    // ---- the cores in front of m, right to left: G = d/d(state leaving core c)
#pragma unroll
    for (int c = NC - 1; c >= 0; --c) {
      if (c < p.mv) {
        const float* pkc = pack + c * E;
        float T[R][R], dF[QC], Gin[R], prod[E];
        sr_tmat<R, QC>(pkc, f[c], T);
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) dF[qq] = 0.f;
#pragma unroll
        for (int l = 0; l < R; ++l) {
          float a = 0.f;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float vg = vs[c][l] * dv[r];
            a = fmaf(T[l][r], dv[r], a);
#pragma unroll
            for (int qq = 0; qq < QC; ++qq) {
              dF[qq] = fmaf(vg, pkc[(l * R + r) * QC + qq], dF[qq]);
              prod[(l * R + r) * QC + qq] = owner ? vg * f[c][qq] : 0.f;
            }
          }
          Gin[l] = a;
        }
        if (p.part != nullptr) sr_wave_reduce_add<E>(prod, wmine + c * E, lane, first);
#pragma unroll
        for (int r = 0; r < R; ++r) dv[r] = Gin[r];
        put_row(c, dF);
      }
    }
  }
  __syncthreads();
  if (p.dX != nullptr) sr_write_dx_band<NC, (TWOCH ? 2 : QC)>(p, dfl, p.n, img, r0, r1, wr0, tid, SR_BWD_THREADS);
  if (p.part != nullptr) {
    for (int e = tid; e < PK; e += SR_BWD_THREADS) {
      float s = 0.f;
#pragma unroll
      for (int wv = 0; wv < NWAVES; ++wv) s += wacc[wv * PK + e];
      p.part[(long long)e * p.nrec + blockIdx.x] = s;
    }
  }
}

// dCore of a string with a many-valued core: padded entry (slice, l, r, qq) -> its place
template <int R, int QC>
__global__ __launch_bounds__(256) void convsbs_regmv_tail_k(const float* __restrict__ part, SrTailP t, int mv, int O) {
  constexpr int E = R * R * QC;
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= (SR_MAXC + O) * E) return;
  const int sl = e / E, rem = e - sl * E;
  const int c = sl < SR_MAXC ? sl : mv, o = sl < SR_MAXC ? 0 : sl - SR_MAXC;
  const int l = rem / (R * QC), r3 = rem - l * (R * QC);
  const int r = r3 / QC, qq = r3 - r * QC;
  if (c >= t.n || (sl < SR_MAXC && c == mv) || l >= t.bl[c] || r >= t.br[c]) return;   // padding (wave-uniform)
  float s = 0.f;
  for (int k = lane; k < t.nrec; k += 64) s += part[(long long)e * t.nrec + k];
  s = wave_reduce_sum(s);
  if (lane == 0) t.dcore[c][((o * t.bl[c] + l) * t.br[c] + r) * QC + qq] = s;
}

// ------------------------------------------------------------------------------------------------ host
struct SrPlan {
  int R, QC, twoch;
  int uniform;   // nine cores, bonds (1, R, ..., R): the scalar-load kernels (R is then the bond itself: 2, 3 or 4)
  int tot;       // elements of all cores together
  int band_rows, bands, nrec, max_w_in_band;
  size_t lds_bwd;
};

bool sr_fill(SrP& p, SrPlan& pl, const void* x, const int64_t xs[5], const void* const* cores, int n, const int* out_sizes,
             const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q, int dtype) {
  if (dtype != DCTN_F32 || n < 2 || n > SR_MAXC) return false;
  if (bond_sizes[0] != 1) return false;            // open chains (rings are walked as slices by the matrix-core family)
  int qc = 1;
  for (int c = 0; c < C; ++c) qc *= q;
  if (!((C == 1 && q >= 2 && q <= 4) || (C == 2 && q == 2))) return false;
  int maxb = 1, otot = 1, max_h = 0, max_w = 0;
  for (int c = 0; c < n; ++c) {
    const int bl = bond_sizes[c], br = bond_sizes[(c + 1) % n];
    if (bl < 1 || br < 1 || bl > 4 || br > 4) return false;
    if (out_sizes[c] != 1 && out_sizes[c] != 2) return false;
    otot *= out_sizes[c];
    maxb = bl > maxb ? bl : maxb;
    maxb = br > maxb ? br : maxb;
    max_h = pos_h[c] > max_h ? pos_h[c] : max_h;
    max_w = pos_w[c] > max_w ? pos_w[c] : max_w;
    p.core[c] = cores ? (const float*)cores[c] : nullptr;
    p.o[c] = out_sizes[c]; p.bl[c] = bl; p.br[c] = br; p.ph[c] = pos_h[c]; p.pw[c] = pos_w[c];
  }
  if (otot > 2 || maxb < 2) return false;
  for (int c = n; c < SR_MAXC; ++c) { p.core[c] = nullptr; p.o[c] = 1; p.bl[c] = 1; p.br[c] = 1; p.ph[c] = 0; p.pw[c] = 0; }
  if (H <= max_h || W <= max_w) return false;
  p.x = (const float*)x;
  for (int i = 0; i < 5; ++i) p.xs[i] = xs ? xs[i] : 0;
  p.n = n; p.C = C; p.q = q; p.B = B; p.H = H; p.W = W; p.Ho = H - max_h; p.Wo = W - max_w; p.Otot = otot; p.max_h = max_h;
  p.Wn = (long long)B * p.Ho * p.Wo;
  p.dY = nullptr; p.dX = nullptr; p.part = nullptr;
  p.mv = -1; p.O = 0;
  pl.R = maxb <= 2 ? 2 : 4;
  pl.QC = qc;
  pl.twoch = C == 2;
  pl.uniform = n == SU_NC;
  for (int c = 0; c < n && pl.uniform; ++c)
    if (bond_sizes[c] != (c == 0 ? 1 : maxb)) pl.uniform = 0;
  if (pl.uniform) pl.R = maxb;
  p.coff[0] = 0;
  for (int c = 0; c < SR_MAXC; ++c) p.coff[c + 1] = p.coff[c] + (c < n ? p.o[c] * p.bl[c] * p.br[c] * qc : 0);
  pl.tot = p.coff[SR_MAXC];
  p.out = nullptr;
  p.tot_all = pl.tot;
  for (int c = 0; c < SR_MAXC; ++c) p.slot[c] = c;
  // bands of pixel rows: the fewest per image whose windows fit one pass of the workgroup's lanes, more (down to
  // ~2 workgroups per CU) when the batch is small
  const int NCq = n * C * q;
  const size_t PK = (size_t)SR_MAXC * 2 * pl.R * pl.R * pl.QC;
  const size_t fixed = pl.uniform ? (size_t)pl.tot * (SR_BWD_THREADS / 64) : PK * (1 + SR_BWD_THREADS / 64);
  int best = -1;
  for (int nb = 1; nb <= H; ++nb) {
    const int br = (H + nb - 1) / nb;
    const int bands = (H + br - 1) / br;
    int maxwin = 0;
    for (int b2 = 0; b2 < bands; ++b2) {
      const int r0 = b2 * br, r1 = r0 + br < H ? r0 + br : H;
      const int wr0 = r0 - max_h > 0 ? r0 - max_h : 0, wr1 = r1 < p.Ho ? r1 : p.Ho;
      const int nw = wr1 > wr0 ? (wr1 - wr0) * p.Wo : 0;
      maxwin = nw > maxwin ? nw : maxwin;
    }
    const size_t lds = (fixed + (size_t)maxwin * NCq) * sizeof(float);
    if (lds > DCTN_LDS_BUDGET) continue;
    best = br; pl.max_w_in_band = maxwin; pl.lds_bwd = lds;
    if (maxwin <= SR_BWD_THREADS && (long long)B * bands >= 256) break;
    if (maxwin <= SR_BWD_THREADS / 2) break;   // finer bands only add redundant halo windows
  }
  if (best < 0) return false;
  pl.band_rows = best;
  pl.bands = (H + best - 1) / best;
  pl.nrec = B * pl.bands;
  p.band_rows = pl.band_rows; p.bands = pl.bands; p.nrec = pl.nrec;
  return true;
}


// the same for a string with ONE many-valued core (3..16 values; every other core one value)
bool sm_fill(SrP& p, SrPlan& pl, const void* x, const int64_t xs[5], const void* const* cores, int n, const int* out_sizes,
             const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q, int dtype) {
  if (dtype != DCTN_F32 || n < 2 || n > SR_MAXC || bond_sizes[0] != 1) return false;
  int qc = 1;
  for (int c = 0; c < C; ++c) qc *= q;
  if (!((C == 1 && q >= 2 && q <= 4) || (C == 2 && q == 2))) return false;
  int maxb = 1, max_h = 0, max_w = 0, mv = -1;
  for (int c = 0; c < n; ++c) {
    const int bl = bond_sizes[c], br = bond_sizes[(c + 1) % n];
    if (bl < 1 || br < 1 || bl > 4 || br > 4) return false;
    if (out_sizes[c] != 1) {
      if (mv >= 0 || out_sizes[c] < 3 || out_sizes[c] > 16) return false;
      mv = c;
    }
    maxb = bl > maxb ? bl : maxb;
    maxb = br > maxb ? br : maxb;
    max_h = pos_h[c] > max_h ? pos_h[c] : max_h;
    max_w = pos_w[c] > max_w ? pos_w[c] : max_w;
    p.core[c] = cores ? (const float*)cores[c] : nullptr;
    p.o[c] = out_sizes[c]; p.bl[c] = bl; p.br[c] = br; p.ph[c] = pos_h[c]; p.pw[c] = pos_w[c];
  }
  if (mv < 0 || maxb < 2) return false;
  for (int c = n; c < SR_MAXC; ++c) { p.core[c] = nullptr; p.o[c] = 1; p.bl[c] = 1; p.br[c] = 1; p.ph[c] = 0; p.pw[c] = 0; }
  if (H <= max_h || W <= max_w) return false;
  p.x = (const float*)x;
  for (int i = 0; i < 5; ++i) p.xs[i] = xs ? xs[i] : 0;
  p.n = n; p.C = C; p.q = q; p.B = B; p.H = H; p.W = W; p.Ho = H - max_h; p.Wo = W - max_w; p.max_h = max_h;
  p.mv = mv; p.O = out_sizes[mv]; p.Otot = p.O;
  p.Wn = (long long)B * p.Ho * p.Wo;
  p.dY = nullptr; p.dX = nullptr; p.part = nullptr; p.out = nullptr;
  pl.R = maxb <= 2 ? 2 : 4;
  pl.QC = qc;
  pl.twoch = C == 2;
  pl.uniform = 0;
  p.coff[0] = 0;
  for (int c = 0; c < SR_MAXC; ++c) p.coff[c + 1] = p.coff[c] + (c < n ? p.o[c] * p.bl[c] * p.br[c] * qc : 0);
  pl.tot = p.coff[SR_MAXC];
  p.tot_all = pl.tot;
  for (int c = 0; c < SR_MAXC; ++c) p.slot[c] = c;
  const int NCq = n * C * q;
  const size_t PK = (size_t)(SR_MAXC + p.O) * pl.R * pl.R * pl.QC;
  const size_t fixed = PK * (1 + SR_BWD_THREADS / 64);
  int best = -1;
  for (int nb = 1; nb <= H; ++nb) {
    const int br = (H + nb - 1) / nb;
    const int bands = (H + br - 1) / br;
    int maxwin = 0;
    for (int b2 = 0; b2 < bands; ++b2) {
      const int r0 = b2 * br, r1 = r0 + br < H ? r0 + br : H;
      const int wr0 = r0 - max_h > 0 ? r0 - max_h : 0, wr1 = r1 < p.Ho ? r1 : p.Ho;
      const int nw = wr1 > wr0 ? (wr1 - wr0) * p.Wo : 0;
      maxwin = nw > maxwin ? nw : maxwin;
    }
    const size_t lds = (fixed + (size_t)maxwin * NCq) * sizeof(float);
    if (lds > DCTN_LDS_BUDGET) continue;
    best = br; pl.max_w_in_band = maxwin; pl.lds_bwd = lds;
    if (maxwin <= SR_BWD_THREADS && (long long)B * bands >= 256) break;
    if (maxwin <= SR_BWD_THREADS / 2) break;
  }
  if (best < 0) return false;
  pl.band_rows = best;
  pl.bands = (H + best - 1) / best;
  pl.nrec = B * pl.bands;
  p.band_rows = pl.band_rows; p.bands = pl.bands; p.nrec = pl.nrec;
  return true;
}

}  // namespace

// room for the per-workgroup dCore records of the register-resident backward (0: the string is not in the family)
size_t convsbs_reg_bwd_workspace(int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                                 int C, int B, int H, int W, int q, int dtype) {
  SrP p;
  SrPlan pl;
  if (!sr_fill(p, pl, nullptr, nullptr, nullptr, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype)) {
    if (!sm_fill(p, pl, nullptr, nullptr, nullptr, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype)) return 0;
    return (size_t)(SR_MAXC + p.O) * pl.R * pl.R * pl.QC * pl.nrec * sizeof(float) + 256;
  }
  const size_t ent = pl.uniform ? (size_t)pl.tot : (size_t)SR_MAXC * 2 * pl.R * pl.R * pl.QC;
  return ent * pl.nrec * sizeof(float) + 256;
}

int convsbs_fwd_reg(const void* x, const int64_t xs[5], const void* const* cores, void* out, int n, const int* out_sizes,
                    const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q, int dtype,
                    hipStream_t st) {
  SrP p;
  SrPlan pl;
  if (!sr_fill(p, pl, x, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype)) {
    // one many-valued core (the classifier's ten-label string): prefix state, suffix vector, one dot product per output
    if (!sm_fill(p, pl, x, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype)) return DCTN_ERR_UNSUPPORTED;
    long long blocks = (p.Wn + SR_FWD_THREADS - 1) / SR_FWD_THREADS;
    if (blocks > 2048) blocks = 2048;
    const size_t lds = (size_t)(SR_MAXC + p.O) * pl.R * pl.R * pl.QC * sizeof(float);
#define SM_FWD(RR, QQ, TC)                                                                                          \
  hipLaunchKernelGGL((convsbs_fwd_regmv_k<RR, QQ, TC>), dim3((unsigned)blocks), dim3(SR_FWD_THREADS), lds, st, p, (float*)out)
    if (pl.R == 2) {
      if (pl.twoch) SM_FWD(2, 4, true);
      else if (pl.QC == 2) SM_FWD(2, 2, false);
      else if (pl.QC == 3) SM_FWD(2, 3, false);
      else SM_FWD(2, 4, false);
    } else {
      if (pl.twoch) SM_FWD(4, 4, true);
      else if (pl.QC == 2) SM_FWD(4, 2, false);
      else if (pl.QC == 3) SM_FWD(4, 3, false);
      else SM_FWD(4, 4, false);
    }
#undef SM_FWD
    DCTN_CHECK_LAUNCH();
    dctn_set_last_kernel("convsbs_fwd_reg_f32");
    return DCTN_OK;
  }
  long long blocks = (p.Wn + SR_FWD_THREADS - 1) / SR_FWD_THREADS;
  if (blocks > 2048) blocks = 2048;
  p.out = (float*)out;
#define SR_FWD(RR, QQ, TC)                                                                                         \
  hipLaunchKernelGGL((convsbs_fwd_reg_k<RR, QQ, TC>), dim3((unsigned)blocks), dim3(SR_FWD_THREADS), 0, st, p, (float*)out)
#define SU_FWD(RR, QQ, TC)                                                                                         \
  hipLaunchKernelGGL((convsbs_fwd_regu_k<RR, QQ, TC, false>), dim3((unsigned)blocks, 1), dim3(SR_FWD_THREADS), 0, st, p, p)
#define SU_FWD_R(RR)                                                                                               \
  do {                                                                                                             \
    if (pl.twoch) SU_FWD(RR, 4, true);                                                                             \
    else if (pl.QC == 2) SU_FWD(RR, 2, false);                                                                     \
    else if (pl.QC == 3) SU_FWD(RR, 3, false);                                                                     \
    else SU_FWD(RR, 4, false);                                                                                     \
  } while (0)
  if (pl.uniform) {
    if (pl.R == 2) SU_FWD_R(2); else if (pl.R == 3) SU_FWD_R(3); else SU_FWD_R(4);
  } else if (pl.R == 2) {
    if (pl.twoch) SR_FWD(2, 4, true);
    else if (pl.QC == 2) SR_FWD(2, 2, false);
    else if (pl.QC == 3) SR_FWD(2, 3, false);
    else SR_FWD(2, 4, false);
  } else {
    if (pl.twoch) SR_FWD(4, 4, true);
    else if (pl.QC == 2) SR_FWD(4, 2, false);
    else if (pl.QC == 3) SR_FWD(4, 3, false);
    else SR_FWD(4, 4, false);
  }
#undef SR_FWD
#undef SU_FWD
#undef SU_FWD_R
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("convsbs_fwd_reg_f32");
  return DCTN_OK;
}

int convsbs_bwd_reg(const void* x, const int64_t xs[5], const void* const* cores, const void* dY, void* dX,
                    float* const* dcores, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h,
                    const int* pos_w, int C, int B, int H, int W, int q, int dtype, hipStream_t st, void* ws, size_t ws_bytes) {
  SrP p;
  SrPlan pl;
  if (!sr_fill(p, pl, x, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype)) {
    if (!sm_fill(p, pl, x, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype)) return DCTN_ERR_UNSUPPORTED;
    const int ent = (SR_MAXC + p.O) * pl.R * pl.R * pl.QC;
    if (dcores && (!ws || ws_bytes < (size_t)ent * pl.nrec * sizeof(float))) return DCTN_ERR_WORKSPACE;
    if ((uintptr_t)ws % 16) return DCTN_ERR_WORKSPACE;
    p.dY = (const float*)dY;
    p.dX = (float*)dX;
    p.part = dcores ? (float*)ws : nullptr;
    SrTailP t;
    for (int c = 0; c < SR_MAXS * SR_MAXC; ++c) t.dcore[c] = (dcores && c < n) ? dcores[c] : nullptr;
    for (int c = 0; c < SR_MAXC; ++c) { t.o[c] = p.o[c]; t.bl[c] = p.bl[c]; t.br[c] = p.br[c]; }
    t.n = n; t.nrec = pl.nrec;
    for (int c = 0; c <= SR_MAXS * SR_MAXC; ++c) t.coff[c] = c <= SR_MAXC ? p.coff[c] : p.coff[SR_MAXC];
#define SM_BWD(RR, QQ, TC)                                                                                            \
  do {                                                                                                                \
    (void)hipFuncSetAttribute((const void*)convsbs_bwd_regmv_k<RR, QQ, TC, SR_MAXC>,                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bwd);                           \
    hipLaunchKernelGGL((convsbs_bwd_regmv_k<RR, QQ, TC, SR_MAXC>), dim3((unsigned)pl.nrec), dim3(SR_BWD_THREADS), pl.lds_bwd, st, p); \
    if (dcores)                                                                                                       \
      hipLaunchKernelGGL((convsbs_regmv_tail_k<RR, QQ>), dim3((unsigned)((ent + 3) / 4)), dim3(256), 0, st, (const float*)ws, t, \
                         p.mv, p.O);                                                                                  \
  } while (0)
    if (pl.R == 2) {
      if (pl.twoch) SM_BWD(2, 4, true);
      else if (pl.QC == 2) SM_BWD(2, 2, false);
      else if (pl.QC == 3) SM_BWD(2, 3, false);
      else SM_BWD(2, 4, false);
    } else {
      if (pl.twoch) SM_BWD(4, 4, true);
      else if (pl.QC == 2) SM_BWD(4, 2, false);
      else if (pl.QC == 3) SM_BWD(4, 3, false);
      else SM_BWD(4, 4, false);
    }
#undef SM_BWD
    DCTN_CHECK_LAUNCH();
    dctn_set_last_kernel("convsbs_bwd_reg_f32");
    return DCTN_OK;
  }
  const size_t ent = pl.uniform ? (size_t)pl.tot : (size_t)SR_MAXC * 2 * pl.R * pl.R * pl.QC;
  const size_t need = ent * pl.nrec * sizeof(float);
  if (dcores && (!ws || ws_bytes < need)) return DCTN_ERR_WORKSPACE;
  if ((uintptr_t)ws % 16) return DCTN_ERR_WORKSPACE;
  p.dY = (const float*)dY;
  p.dX = (float*)dX;
  p.part = dcores ? (float*)ws : nullptr;
  const unsigned grid = (unsigned)pl.nrec;
#define SR_BWD(RR, QQ, TC)                                                                                            \
  do {                                                                                                                \
    (void)hipFuncSetAttribute((const void*)convsbs_bwd_reg_k<RR, QQ, TC, SR_MAXC>,                                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bwd);                           \
    hipLaunchKernelGGL((convsbs_bwd_reg_k<RR, QQ, TC, SR_MAXC>), dim3(grid), dim3(SR_BWD_THREADS), pl.lds_bwd, st, p); \
  } while (0)
#define SR_TAIL(RR, QQ)                                                                                               \
  hipLaunchKernelGGL((convsbs_reg_tail_k<RR, QQ>), dim3((unsigned)((n * 2 * RR * RR * QQ + 3) / 4)), dim3(256), 0, st, \
                     (const float*)ws, t)
  SrTailP t;
  for (int c = 0; c < SR_MAXC; ++c) {
    t.dcore[c] = (dcores && c < n) ? dcores[c] : nullptr;
    t.o[c] = p.o[c]; t.bl[c] = p.bl[c]; t.br[c] = p.br[c];
  }
  t.n = n; t.nrec = pl.nrec;
  for (int c = 0; c <= SR_MAXS * SR_MAXC; ++c) t.coff[c] = c <= SR_MAXC ? p.coff[c] : p.coff[SR_MAXC];
  for (int c = SR_MAXC; c < SR_MAXS * SR_MAXC; ++c) t.dcore[c] = nullptr;
#define SU_BWD(RR, QQ, TC)                                                                                            \
  do {                                                                                                                \
    (void)hipFuncSetAttribute((const void*)convsbs_bwd_regu_k<RR, QQ, TC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)pl.lds_bwd);                                                                       \
    hipLaunchKernelGGL((convsbs_bwd_regu_k<RR, QQ, TC, false>), dim3(grid), dim3(SR_BWD_THREADS), pl.lds_bwd, st, p, p, 1);  \
  } while (0)
#define SU_BWD_R(RR)                                                                                                  \
  do {                                                                                                                \
    if (pl.twoch) SU_BWD(RR, 4, true);                                                                                \
    else if (pl.QC == 2) SU_BWD(RR, 2, false);                                                                        \
    else if (pl.QC == 3) SU_BWD(RR, 3, false);                                                                        \
    else SU_BWD(RR, 4, false);                                                                                        \
  } while (0)
  if (pl.uniform) {
    if (pl.R == 2) SU_BWD_R(2); else if (pl.R == 3) SU_BWD_R(3); else SU_BWD_R(4);
    if (dcores)
      hipLaunchKernelGGL(convsbs_regu_tail_k, dim3((unsigned)((pl.tot + 3) / 4)), dim3(256), 0, st, (const float*)ws, t, pl.tot);
  } else if (pl.R == 2) {
    if (pl.twoch) { SR_BWD(2, 4, true); if (dcores) SR_TAIL(2, 4); }
    else if (pl.QC == 2) { SR_BWD(2, 2, false); if (dcores) SR_TAIL(2, 2); }
    else if (pl.QC == 3) { SR_BWD(2, 3, false); if (dcores) SR_TAIL(2, 3); }
    else { SR_BWD(2, 4, false); if (dcores) SR_TAIL(2, 4); }
  } else {
    if (pl.twoch) { SR_BWD(4, 4, true); if (dcores) SR_TAIL(4, 4); }
    else if (pl.QC == 2) { SR_BWD(4, 2, false); if (dcores) SR_TAIL(4, 2); }
    else if (pl.QC == 3) { SR_BWD(4, 3, false); if (dcores) SR_TAIL(4, 3); }
    else { SR_BWD(4, 4, false); if (dcores) SR_TAIL(4, 4); }
  }
#undef SR_BWD
#undef SR_TAIL
#undef SU_BWD
#undef SU_BWD_R
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("convsbs_bwd_reg_f32");
  return DCTN_OK;
}

// ------------------------------------------------------------------------------------------------ several strings per launch
// ManyConvSBS (dctn/conv_sbs.py:314-370): the strings of one layer read the same input.  When all of them are uniform
// nine-core strings of one bond over the same window positions (the reference's layers: two snakes through the same 3 x 3
// window, mnist.py:189-252) they run as ONE forward launch (grid.y = string) and ONE backward launch (each lane walks
// its window through every string; the strings' feature gradients meet in the same per-window LDS row, so dX is written
// once, already summed) plus the one tail kernel.
namespace {
bool su_plan_many(SrP* ps, SrPlan& pl, int ns, const void* x, const int64_t xs[5], const void* const* cores, int n,
                  const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W,
                  int q, int dtype) {
  if (ns < 1 || ns > SR_MAXS) return false;
  int base = 0;
  for (int s2 = 0; s2 < ns; ++s2) {
    SrPlan pls;
    if (!sr_fill(ps[s2], pls, x, xs, cores ? cores + s2 * n : nullptr, n, out_sizes + s2 * n, bond_sizes + s2 * n, pos_h + s2 * n,
                 pos_w + s2 * n, C, B, H, W, q, dtype))
      return false;
    if (!pls.uniform) return false;
    if (s2 == 0) pl = pls;
    else if (pls.R != pl.R || pls.QC != pl.QC || pls.twoch != pl.twoch || ps[s2].max_h != ps[0].max_h || ps[s2].Wo != ps[0].Wo)
      return false;
    for (int c = 0; c < n; ++c) {   // the pixel of core c in the first string's order
      int found = -1;
      for (int c0 = 0; c0 < n; ++c0)
        if (ps[0].ph[c0] == ps[s2].ph[c] && ps[0].pw[c0] == ps[s2].pw[c]) found = c0;
      if (found < 0) return false;
      ps[s2].slot[c] = found;
    }
    for (int c = 0; c <= SR_MAXC; ++c) ps[s2].coff[c] += base;
    base += pls.tot;
  }
  for (int s2 = 0; s2 < ns; ++s2) ps[s2].tot_all = base;
  pl.tot = base;
  pl.lds_bwd = ((size_t)base * (SR_BWD_THREADS / 64) + (size_t)pl.max_w_in_band * n * C * q) * sizeof(float);
  return pl.lds_bwd <= DCTN_LDS_BUDGET;
}
}  // namespace

size_t convsbs_many_reg_bwd_workspace(int ns, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                                      int C, int B, int H, int W, int q, int dtype) {
  SrP ps[SR_MAXS];
  SrPlan pl;
  if (!su_plan_many(ps, pl, ns, nullptr, nullptr, nullptr, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype)) return 0;
  return (size_t)pl.tot * pl.nrec * sizeof(float) + 256;
}

int convsbs_many_fwd_reg(const void* x, const int64_t xs[5], const void* const* cores, void* const* outs, int ns, int n,
                         const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H,
                         int W, int q, int dtype, hipStream_t st) {
  SrP ps[SR_MAXS];
  SrPlan pl;
  if (!su_plan_many(ps, pl, ns, x, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype)) return DCTN_ERR_UNSUPPORTED;
  for (int s2 = 0; s2 < ns; ++s2) {
    if (!outs[s2]) return DCTN_ERR_NULL;
    ps[s2].out = (float*)outs[s2];
  }
  long long blocks = (ps[0].Wn + SR_FWD_THREADS - 1) / SR_FWD_THREADS;
  if (blocks > 2048) blocks = 2048;
  const SrP& pa = ps[0];
  const SrP& pb = ps[ns > 1 ? 1 : 0];
#define SU_FWD(RR, QQ, TC)                                                                                         \
  hipLaunchKernelGGL((convsbs_fwd_regu_k<RR, QQ, TC, true>), dim3((unsigned)blocks, (unsigned)ns), dim3(SR_FWD_THREADS), 0, st, pa, pb)
#define SU_FWD_R(RR)                                                                                               \
  do {                                                                                                             \
    if (pl.twoch) SU_FWD(RR, 4, true);                                                                             \
    else if (pl.QC == 2) SU_FWD(RR, 2, false);                                                                     \
    else if (pl.QC == 3) SU_FWD(RR, 3, false);                                                                     \
    else SU_FWD(RR, 4, false);                                                                                     \
  } while (0)
  if (pl.R == 2) SU_FWD_R(2); else if (pl.R == 3) SU_FWD_R(3); else SU_FWD_R(4);
#undef SU_FWD
#undef SU_FWD_R
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("convsbs_many_fwd_reg_f32");
  return DCTN_OK;
}

int convsbs_many_bwd_reg(const void* x, const int64_t xs[5], const void* const* cores, const void* const* dYs, void* dX,
                         float* const* dcores, int ns, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h,
                         const int* pos_w, int C, int B, int H, int W, int q, int dtype, hipStream_t st, void* ws, size_t ws_bytes) {
  SrP ps[SR_MAXS];
  SrPlan pl;
  if (!su_plan_many(ps, pl, ns, x, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype)) return DCTN_ERR_UNSUPPORTED;
  const size_t need = (size_t)pl.tot * pl.nrec * sizeof(float);
  if (dcores && (!ws || ws_bytes < need || ((uintptr_t)ws % 16))) return DCTN_ERR_WORKSPACE;
  for (int s2 = 0; s2 < ns; ++s2) {
    if (!dYs[s2]) return DCTN_ERR_NULL;
    ps[s2].dY = (const float*)dYs[s2];
    ps[s2].dX = (float*)dX;
    ps[s2].part = dcores ? (float*)ws : nullptr;
  }
  SrTailP t;
  for (int c = 0; c < SR_MAXC; ++c) { t.o[c] = 1; t.bl[c] = 1; t.br[c] = 1; }
  for (int s2 = 0; s2 < SR_MAXS; ++s2)
    for (int c = 0; c < SR_MAXC; ++c) {
      const bool live = s2 < ns && c < n;
      t.dcore[s2 * SR_MAXC + c] = (dcores && live) ? dcores[s2 * n + c] : nullptr;
      t.coff[s2 * SR_MAXC + c] = s2 < ns ? ps[s2].coff[c] : pl.tot;
    }
  t.coff[SR_MAXS * SR_MAXC] = pl.tot;
  t.n = SR_MAXS * SR_MAXC;
  t.nrec = pl.nrec;
  const unsigned grid = (unsigned)pl.nrec;
  const SrP& pa = ps[0];
  const SrP& pb = ps[ns > 1 ? 1 : 0];
#define SU_BWD(RR, QQ, TC)                                                                                            \
  do {                                                                                                                \
    (void)hipFuncSetAttribute((const void*)convsbs_bwd_regu_k<RR, QQ, TC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)pl.lds_bwd);                                                                       \
    hipLaunchKernelGGL((convsbs_bwd_regu_k<RR, QQ, TC, true>), dim3(grid), dim3(SR_BWD_THREADS), pl.lds_bwd, st, pa, pb, ns); \
  } while (0)
#define SU_BWD_R(RR)                                                                                                  \
  do {                                                                                                                \
    if (pl.twoch) SU_BWD(RR, 4, true);                                                                                \
    else if (pl.QC == 2) SU_BWD(RR, 2, false);                                                                        \
    else if (pl.QC == 3) SU_BWD(RR, 3, false);                                                                        \
    else SU_BWD(RR, 4, false);                                                                                        \
  } while (0)
  if (pl.R == 2) SU_BWD_R(2); else if (pl.R == 3) SU_BWD_R(3); else SU_BWD_R(4);
#undef SU_BWD
#undef SU_BWD_R
  if (dcores)
    hipLaunchKernelGGL(convsbs_regu_tail_k, dim3((unsigned)((pl.tot + 3) / 4)), dim3(256), 0, st, (const float*)ws, t, pl.tot);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("convsbs_many_bwd_reg_f32");
  return DCTN_OK;
}
