// MFMA ConvSBS forward sweep: open chain (bond_sizes[0] == 1), internal bonds <= 32 padded to one tile size r in
// {4, 8, 16, 32} (they need not be equal), q^C <= 4, at most 2 outputs in total, float32 (exact: v_mfma_f32_32x32x2_f32);
// rings, cores with many outputs, several cores with outputs: as slices of that family (sbsm_for_slices).
// This is the string family of the reference's models (mnist.py:189-223: 9-core snakes with one
// 2-output core) on MNIST (q=2), its second layer (C=2, q=2) and the CIFAR colour layout (q=3).
//
// Replaces dctn/conv_sbs.py:258-304.  Per middle core c the reference materialises
// T_c[w][o,l,r'] in HBM and multiplies the chain; here, per 32 windows,
//     U[(o, r', qq), w] = sum_l core_c[o, l, r', qq] * v[l, w]          (MFMA: M = (o,r',qq), K = l)
//     v'[o, r', w]      = sum_qq f_c[w, qq] * U[(o, r', qq), w]         (lane-local epilogue)
// A lane owns a window (column); the state v lives in registers in B-operand layout
// (lane half h holds l = 2s + h).  With rows ordered i = 4*r'_local + qq the accumulator gives lane
// half h' exactly the r' = 2s' + h' it needs as the next core's operand, so the whole chain runs
// in registers; the cores (packed once per workgroup into LDS in A-operand order) and x are the
// only memory traffic.  The first (l = 1) and last (r' = 1) cores are tiny and run on the VALU.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) int int2v;

#define SBSM_MAXC 32

struct SbsMP {
  int n, C, B, H, W, q, qc, Ho, Wo, Otot;
  int Ra;                     // the string's largest bond (<= the kernels' R: packs, tables and states are zero beyond a core's bonds)
  int bl[SBSM_MAXC], br[SBSM_MAXC];   // left / right bond of core c (1 at the open ends); they need not be equal
  int ostride, obase, ostep;  // row length of out / dY, the first output of this launch and the distance to its second (many-output strings run in slices)
  int accum;                  // backward: add to gxw and to the core gradients instead of setting them (slices after the first)
  int out_accum;              // forward: add to out (rings: one launch per value of the closing bond)
  float* save;                // forward: store every core's input state here for a following backward (or NULL)
  int last_stride;            // elements between the rows of the last core's table (qc; rings: a column of its matrix)
  long long Wn, ngroups;
  long long s[5];
  int o[SBSM_MAXC], ph[SBSM_MAXC], pw[SBSM_MAXC];
  int apack_off[SBSM_MAXC];   // float offset of core c's packed A operand in LDS (middle cores)
  int dacc_off[SBSM_MAXC + 1];  // backward: float offset of core c's dCore accumulator in LDS
  long long st_off[SBSM_MAXC + 1];  // backward: element offsets of the stored forward states
  float* dcore[SBSM_MAXC];    // backward: global dCore (zero-initialised by the caller)
  int apack2_off[SBSM_MAXC];  // backward (16x16x4 kernel): float offset of core c's adjoint pack
  int core_off[SBSM_MAXC + 1];  // backward: float offset of core c inside one workgroup's partial-gradient record
  float* partials;            // backward: [workgroup][core_off[n]] partial gradients (NULL: global float atomics)
  int zt_off, vt_off;         // backward (16x16x4 kernel): per-wave transposition tiles
  int first_off, last_off;    // float offsets of the first / last core tables in LDS
  int fs_off;                 // float offset of the per-wave feature slices (4 waves x n*4*32)
  unsigned char digit[4][4];  // digit[qq][ch]: feature index of channel ch in the flat index qq (host-filled)
  const float* core[SBSM_MAXC];
};

namespace {

__device__ __forceinline__ float half_sum(float v) {
  const int iv = __float_as_int(v);
  const int2v r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}

constexpr int ROWP = 65;  // padded length of a packed k-step row (64 lanes + 1): the adjoint sweep
                          // gathers the same pack with a per-lane row offset, 2-way conflicts at most

// f[qq] = prod_ch x[ch][pixel of core c][digit_ch(qq)] (channel 0 most significant), qq < 4
__device__ __forceinline__ void features(const float* __restrict__ x, const SbsMP& p, int c,
                                         long long b, int ho, int wo, bool valid, float (&f)[4]) {
  const float* base = x + b * p.s[1] + (long long)(ho + p.ph[c]) * p.s[2] + (long long)(wo + p.pw[c]) * p.s[3];
  // per channel the q feature values of this pixel (q <= 4), then the products by table
  float xv[2][4];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch)
#pragma unroll
    for (int d = 0; d < 4; ++d)
      xv[ch][d] = (valid && ch < p.C && d < p.q) ? base[ch * p.s[0] + d * p.s[4]] : 1.f;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    float pr = (qq < p.qc && valid) ? 1.f : 0.f;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const int dg = p.digit[qq][ch];
      const float xs = dg == 0 ? xv[ch][0] : dg == 1 ? xv[ch][1] : dg == 2 ? xv[ch][2] : xv[ch][3];
      pr *= ch < p.C ? xs : 1.f;
    }
    f[qq] = pr;
  }
}

// All feature products of a window (every core of the string) at once: the pixel loads of up to 8
// cores are issued back to back (one memory round trip per chunk instead of one per core), the
// products go to the wave's LDS slice fs[(c*4 + qq)*32 + window] and are read back per core.
// ONECH: 1 = the caller knows p.C == 1, 2 = it knows C == 2 and q == 2 (the deeper layers of the reference's classifier):
// only that path is compiled; 0 = decided at run time.  Mode 2 also keeps the pixels' raw values, in a second slice
// behind the products (xs[(c*4 + ch*2 + d)*WPG + window]): the way back turns d/d(products) into d/d(pixel values) from
// them instead of re-reading x through runtime channel loops.
template <int WPG = 32, int CB = 8, int ONECH = 0, bool RAW = true>   // CB: cores per batch of loads (one round trip per batch); RAW: mode 2 keeps the raw values
__device__ __forceinline__ void stage_features(const float* __restrict__ x, const SbsMP& p, long long b, int ho,
                                               int wo, bool valid, float* fs, int lane) {
  const float* win = x + b * p.s[1] + (long long)ho * p.s[2] + (long long)wo * p.s[3];
  if constexpr (ONECH == 2) {
    float* xs = fs + p.n * 4 * WPG;
    for (int c0 = 0; c0 < p.n; c0 += CB) {
      float raw2[CB][2][2];
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) {
        const int c = c0 + cc < p.n ? c0 + cc : p.n - 1;
        const float* base = win + (long long)p.ph[c] * p.s[2] + (long long)p.pw[c] * p.s[3];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int d = 0; d < 2; ++d) raw2[cc][ch][d] = base[ch * p.s[0] + d * p.s[4]];
      }
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) {
        if (c0 + cc < p.n) {
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {   // channel 0 is the most significant digit of the product index
            fs[((c0 + cc) * 4 + qq) * WPG + (lane & (WPG - 1))] = valid ? raw2[cc][0][qq >> 1] * raw2[cc][1][qq & 1] : 0.f;
            if (RAW) xs[((c0 + cc) * 4 + qq) * WPG + (lane & (WPG - 1))] = valid ? raw2[cc][qq >> 1][qq & 1] : 0.f;
          }
        }
      }
    }
    return;
  }
  if (ONECH == 1 || p.C == 1) {   // one channel: the feature products ARE the pixel's q values (no digit table, half the loads)
    for (int c0 = 0; c0 < p.n; c0 += CB) {
      float raw1[CB][4];
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) {
        const int c = c0 + cc < p.n ? c0 + cc : p.n - 1;
        const float* base = win + (long long)p.ph[c] * p.s[2] + (long long)p.pw[c] * p.s[3];
#pragma unroll
        for (int d = 0; d < 4; ++d) raw1[cc][d] = base[(d < p.q ? d : 0) * p.s[4]];
      }
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) {
        if (c0 + cc < p.n) {
#pragma unroll
          for (int d = 0; d < 4; ++d)
            fs[((c0 + cc) * 4 + d) * WPG + (lane & (WPG - 1))] = (valid && d < p.q) ? raw1[cc][d] : 0.f;
        }
      }
    }
    return;
  }
  if constexpr (ONECH != 0) return;
  for (int c0 = 0; c0 < p.n; c0 += CB) {
    float raw[CB][2][4];
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) {
      const int c = c0 + cc < p.n ? c0 + cc : p.n - 1;
      const float* base = win + (long long)p.ph[c] * p.s[2] + (long long)p.pw[c] * p.s[3];
      // clamped index + select instead of a predicated load: no control flow between the loads, they all go out back to
      // back (the caller passes window 0 for lanes without a window, so every address is valid)
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float ld = base[(ch < p.C ? ch : 0) * p.s[0] + (d < p.q ? d : 0) * p.s[4]];
          raw[cc][ch][d] = (valid && ch < p.C && d < p.q) ? ld : 1.f;
        }
    }
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) {
      if (c0 + cc < p.n) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          float pr = (qq < p.qc && valid) ? 1.f : 0.f;
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) {
            const int dg = p.digit[qq][ch];
            const float xs = dg == 0 ? raw[cc][ch][0] : dg == 1 ? raw[cc][ch][1] : dg == 2 ? raw[cc][ch][2] : raw[cc][ch][3];
            pr *= ch < p.C ? xs : 1.f;
          }
          fs[((c0 + cc) * 4 + qq) * WPG + (lane & (WPG - 1))] = pr;   // WPG = 32: both lane halves hold the same window
        }
      }
    }
  }
}
__device__ __forceinline__ void load_features(const float* fs, int c, int lane, float (&f)[4]) {
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) f[qq] = fs[(c * 4 + qq) * 32 + (lane & 31)];
}

// Pack the cores into LDS.  Middle core c, output o, tile t, k-step s:
//   A[lane] = core[o][l = 2s + (lane>>5)][r' = 8t + (i>>2)][qq = i&3], rows of ROWP floats.
template <int R>
__device__ __forceinline__ void pack_cores(float* lds, const SbsMP& p, int tid) {
  constexpr int KS = R / 2;
  constexpr int TILES = R >= 8 ? R / 8 : 1;
  for (int c = 1; c + 1 < p.n; ++c) {
    const int oc = p.o[c];
    const int total = oc * TILES * KS * 64;
    for (int e0 = tid; e0 < total; e0 += 256) {
      const int ln = e0 & 63, i = ln & 31, hh = ln >> 5;
      int t2 = e0 >> 6;
      const int e = (e0 >> 6) * ROWP + ln;
      const int s = t2 % KS; t2 /= KS;
      const int t = t2 % TILES;
      const int o = t2 / TILES;
      const int l = 2 * s + hh, rp = 8 * t + (i >> 2), qq = i & 3;
      float v = 0.f;
      if (l < p.bl[c] && rp < p.br[c] && qq < p.qc) v = p.core[c][(long long)((o * p.bl[c] + l) * p.br[c] + rp) * p.qc + qq];
      lds[p.apack_off[c] + e] = v;
    }
  }
  // first core (1, 1, R, qc) as [r'][4]; last core (1, R, 1, qc) as [l][4]
  for (int e = tid; e < R * 4; e += 256) {
    const int rr = e >> 2, qq = e & 3;
    lds[p.first_off + e] = (qq < p.qc && rr < p.br[0]) ? p.core[0][(long long)rr * p.qc + qq] : 0.f;
    lds[p.last_off + e] = (qq < p.qc && rr < p.bl[p.n - 1]) ? p.core[p.n - 1][(long long)rr * p.last_stride + qq] : 0.f;
  }
}

template <int R, int CH>   // CH: compiled channel handling (stage_features' ONECH)
__global__ __launch_bounds__(256) void convsbs_fwd_mfma_k(const float* __restrict__ x,
                                                          float* __restrict__ out, SbsMP p) {
  constexpr int KS = R / 2;                  // MFMA k-steps per core (K = l = R)
  constexpr int TILES = R >= 8 ? R / 8 : 1;  // 32-row tiles per output index o (rows = 4 * R)
  constexpr int SR = R / 2;                  // state registers per lane: l = 2s + h
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wl = lane & 31, h = lane >> 5;

  pack_cores<R>(lds, p, tid);
  __syncthreads();

  const long long wave = (long long)blockIdx.x * 4 + (tid >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  const int hw = p.Ho * p.Wo;
  for (long long grp = wave; grp < p.ngroups; grp += nwaves) {
    const long long w = grp * 32 + wl;
    const bool valid = w < p.Wn;
    const long long ww = valid ? w : 0;
    const long long b = ww / hw;
    const int rem = (int)(ww - b * hw);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    float f[4];
    float* fs = lds + p.fs_off + (tid >> 6) * p.n * 128;
    stage_features<32, 8, CH, false>(x, p, b, ho, wo, valid, fs, lane);
    // ---- first core: v[0][s] = sum_qq core0[r' = 2s + h][qq] * f[qq]
    float v0[SR], v1[SR];
    load_features(fs, 0, lane, f);
#pragma unroll
    for (int s = 0; s < SR; ++s) {
      const float* cp = lds + p.first_off + (2 * s + h) * 4;
      v0[s] = cp[0] * f[0] + cp[1] * f[1] + cp[2] * f[2] + cp[3] * f[3];
      v1[s] = 0.f;
    }
    int oacc = 1;
    // the input state of core c, for the backward that follows a training forward ([state element][window], the layout
    // the backward's own forward sweep writes): element l = 2s + h of this lane's window
    auto save_state = [&](int c) {
      if (!p.save || !valid) return;
#pragma unroll
      for (int s = 0; s < SR; ++s) {
        p.save[(p.st_off[c] + 2 * s + h) * p.Wn + w] = v0[s];
        if (oacc > 1) p.save[(p.st_off[c] + R + 2 * s + h) * p.Wn + w] = v1[s];
      }
    };
    // ---- middle cores on the matrix pipe
    for (int c = 1; c + 1 < p.n; ++c) {
      const int oc = p.o[c];
      save_state(c);
      load_features(fs, c, lane, f);
      float n0[SR], n1[SR];
#pragma unroll
      for (int s = 0; s < SR; ++s) { n0[s] = 0.f; n1[s] = 0.f; }
      // (input state index a, output index o) -> new state index a * oc + o; at most 2 states
      for (int a = 0; a < oacc; ++a)
        for (int o = 0; o < oc; ++o) {
          const float* ap = lds + p.apack_off[c] + o * TILES * KS * ROWP + lane;
#pragma unroll
          for (int t = 0; t < TILES; ++t) {
            f32x16 D;
#pragma unroll
            for (int vv = 0; vv < 16; ++vv) D[vv] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
              const float bv = a == 0 ? v0[s] : v1[s];
              D = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(t * KS + s) * ROWP], bv, D, 0, 0, 0);
            }
            // rows i = (vv&3) + 8*(vv>>2) + 4h  ->  qq = vv&3, r' = 2*((vv>>2) + 4t) + h
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int sp = g + 4 * t;
              if (sp < SR) {
                const float val = f[0] * D[4 * g] + f[1] * D[4 * g + 1] + f[2] * D[4 * g + 2] + f[3] * D[4 * g + 3];
                if (a * oc + o == 0) n0[sp] = val; else n1[sp] = val;
              }
            }
          }
        }
#pragma unroll
      for (int s = 0; s < SR; ++s) { v0[s] = n0[s]; v1[s] = n1[s]; }
      oacc *= oc;
    }
    // ---- last core: out[a] = sum_l v[a][l] * sum_qq coreL[l][qq] f[qq]
    save_state(p.n - 1);
    load_features(fs, p.n - 1, lane, f);
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int s = 0; s < SR; ++s) {
      const float* cp = lds + p.last_off + (2 * s + h) * 4;
      const float tl = cp[0] * f[0] + cp[1] * f[1] + cp[2] * f[2] + cp[3] * f[3];
      r0 += v0[s] * tl;
      r1 += v1[s] * tl;
    }
    r0 = half_sum(r0);
    r1 = half_sum(r1);
    if (valid && h == 0) {
      float* op = out + w * p.ostride + p.obase;
      op[0] = p.out_accum ? op[0] + r0 : r0;
      if (p.Otot > 1) op[p.ostep] = p.out_accum ? op[p.ostep] + r1 : r1;
    }
  }
}

// ------------------------------------------------------------------------------------ backward
// Same family, R <= 16.  Per 32 windows, everything in registers:
//   * forward sweep, input state of every core stored to the workspace (state layout, coalesced);
//   * adjoint sweep  dv[l] = sum_(o,r',qq) core[o,l,r',qq] f[qq] G[(a,o),r']  as an MFMA whose
//     A operand is gathered from the SAME forward pack with a per-lane row offset, rows permuted
//     (row i <-> l = 2((i&3) + 4(i>>3)) + ((i>>2)&1)) so that the accumulator comes out in state
//     layout again; k is ordered ((r'>>1), qq, r'&1) so a lane multiplies its own G with f[qq];
//   * d/d(features): df[qq] = sum_r' G[r'] U[(r',qq)] with U recomputed by the forward MFMA;
//   * dCore_c += (f (x) G)^T v: windows are the k index, so both operands are transposed on the
//     matrix core (identity B operand), then multiplied; the tile is added to the workgroup's LDS
//     accumulator (ds_add_f32) and flushed once per workgroup with global float atomics.
template <int R>
__global__ __launch_bounds__(256) void convsbs_bwd_mfma_k(const float* __restrict__ x,
                                                          const float* __restrict__ dY,
                                                          float* __restrict__ states,
                                                          float* __restrict__ gxw, SbsMP p, int need_dx) {
  constexpr int KS = R / 2;
  constexpr int TILES = R >= 8 ? R / 8 : 1;
  constexpr int SR = R / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wl = lane & 31, h = lane >> 5;
  pack_cores<R>(lds, p, tid);
  float* dacc = lds + p.dacc_off[0];
  const int dacc_total = p.dacc_off[p.n] - p.dacc_off[0];
  for (int e = tid; e <= dacc_total; e += 256) dacc[e] = 0.f;   // one extra slot stays zero
  const int zero_slot = p.dacc_off[p.n];                         // reads of invalid A rows land here
  __syncthreads();

  // adjoint A operand: row wl <-> l2, gathered from the forward pack
  const int sp_row = (wl & 3) + 4 * (wl >> 3);
  const bool row_ok = sp_row < SR;
  const int l2 = 2 * sp_row + ((wl >> 2) & 1);
  const int lane_base = (l2 >> 1) * ROWP + (l2 & 1) * 32 + (h << 2);

  const long long wave = (long long)blockIdx.x * 4 + (tid >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  const int hw = p.Ho * p.Wo;
  for (long long grp = wave; grp < p.ngroups; grp += nwaves) {
    const long long w = grp * 32 + wl;
    const bool valid = w < p.Wn;
    const long long ww = valid ? w : 0;
    const long long b = ww / hw;
    const int rem = (int)(ww - b * hw);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const float dy0 = valid ? dY[w * p.ostride + p.obase] : 0.f;
    const float dy1 = (valid && p.Otot > 1) ? dY[w * p.ostride + p.obase + p.ostep] : 0.f;
    float f[4];
    float* fs = lds + p.fs_off + (tid >> 6) * p.n * 128;
    stage_features(x, p, b, ho, wo, valid, fs, lane);

    auto write_dx = [&](int c, const float (&df)[4]) {
      if (!need_dx || !valid || h != 0) return;
      for (int ch = 0; ch < p.C; ++ch)
        for (int qv = 0; qv < p.q; ++qv) {
          float g = 0.f;
          for (int qq = 0; qq < p.qc; ++qq) {
            int t = qq;
            float pr = 1.f;
            bool hit = false;
            for (int c2 = p.C - 1; c2 >= 0; --c2) {
              const int dg = t % p.q;
              t /= p.q;
              if (c2 == ch) hit = (dg == qv);
              else pr *= x[c2 * p.s[0] + b * p.s[1] + (long long)(ho + p.ph[c]) * p.s[2] +
                           (long long)(wo + p.pw[c]) * p.s[3] + dg * p.s[4]];
            }
            if (hit) g += df[qq] * pr;
          }
          gxw[(long long)((c * p.C + ch) * p.q + qv) * p.Wn + w] = g;
        }
    };
    auto store_state = [&](int c, int oacc, const float (&a0)[SR], const float (&a1)[SR]) {
      if (!valid) return;
#pragma unroll
      for (int s = 0; s < SR; ++s) {
        states[(p.st_off[c] + 2 * s + h) * p.Wn + w] = a0[s];
        if (oacc > 1) states[(p.st_off[c] + R + 2 * s + h) * p.Wn + w] = a1[s];
      }
    };

    // ---------------- forward sweep
    float v0[SR], v1[SR];
    load_features(fs, 0, lane, f);
#pragma unroll
    for (int s = 0; s < SR; ++s) {
      const float* cp = lds + p.first_off + (2 * s + h) * 4;
      v0[s] = cp[0] * f[0] + cp[1] * f[1] + cp[2] * f[2] + cp[3] * f[3];
      v1[s] = 0.f;
    }
    int oacc = 1;
    for (int c = 1; c + 1 < p.n; ++c) {
      store_state(c, oacc, v0, v1);
      const int oc = p.o[c];
      load_features(fs, c, lane, f);
      float n0[SR], n1[SR];
#pragma unroll
      for (int s = 0; s < SR; ++s) { n0[s] = 0.f; n1[s] = 0.f; }
      for (int a = 0; a < oacc; ++a)
        for (int o = 0; o < oc; ++o) {
          const float* ap = lds + p.apack_off[c] + o * TILES * KS * ROWP + lane;
#pragma unroll
          for (int t = 0; t < TILES; ++t) {
            f32x16 D;
#pragma unroll
            for (int vv = 0; vv < 16; ++vv) D[vv] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
              D = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(t * KS + s) * ROWP], a == 0 ? v0[s] : v1[s], D, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int sp = g + 4 * t;
              if (sp < SR) {
                const float val = f[0] * D[4 * g] + f[1] * D[4 * g + 1] + f[2] * D[4 * g + 2] + f[3] * D[4 * g + 3];
                if (a * oc + o == 0) n0[sp] = val; else n1[sp] = val;
              }
            }
          }
        }
#pragma unroll
      for (int s = 0; s < SR; ++s) { v0[s] = n0[s]; v1[s] = n1[s]; }
      oacc *= oc;
    }
    store_state(p.n - 1, oacc, v0, v1);

    // ---------------- last core
    float G0[SR], G1[SR];
    {
      load_features(fs, p.n - 1, lane, f);
      float df[4] = {0.f, 0.f, 0.f, 0.f};
      float* dl = lds + p.dacc_off[p.n - 1];
#pragma unroll
      for (int s = 0; s < SR; ++s) {
        const float* cp = lds + p.last_off + (2 * s + h) * 4;
        const float tl = cp[0] * f[0] + cp[1] * f[1] + cp[2] * f[2] + cp[3] * f[3];
        G0[s] = dy0 * tl;
        G1[s] = dy1 * tl;
        const float u = dy0 * v0[s] + dy1 * v1[s];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          df[qq] += u * cp[qq];
          if (qq < p.qc) atomicAdd(&dl[(2 * s + h) * p.qc + qq], u * f[qq]);
        }
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) df[qq] = half_sum(df[qq]);
      write_dx(p.n - 1, df);
    }

    // ---------------- middle cores, right to left
    int oacc_out = oacc;
    for (int c = p.n - 2; c >= 1; --c) {
      const int oc = p.o[c];
      const int oacc_in = oacc_out / oc;
      // input state of core c
#pragma unroll
      for (int s = 0; s < SR; ++s) {
        // (clamped address + select: a predicated load is a branch and a wait on the spot)
        const float l0 = states[(p.st_off[c] + 2 * s + h) * p.Wn + (valid ? w : 0)];
        const float l1 = states[(p.st_off[c] + (oacc_in > 1 ? R : 0) + 2 * s + h) * p.Wn + (valid ? w : 0)];
        v0[s] = valid ? l0 : 0.f;
        v1[s] = (valid && oacc_in > 1) ? l1 : 0.f;
      }
      load_features(fs, c, lane, f);
      float df[4] = {0.f, 0.f, 0.f, 0.f};
      float d0[SR], d1[SR];
#pragma unroll
      for (int s = 0; s < SR; ++s) { d0[s] = 0.f; d1[s] = 0.f; }
      for (int a = 0; a < oacc_in; ++a)
        for (int o = 0; o < oc; ++o) {
          const bool g_first = (a * oc + o) == 0;
          float Gs[SR], vin[SR];
#pragma unroll
          for (int s = 0; s < SR; ++s) { Gs[s] = g_first ? G0[s] : G1[s]; vin[s] = a == 0 ? v0[s] : v1[s]; }
          const float* ap = lds + p.apack_off[c] + o * TILES * KS * ROWP;
          // (1) df[qq] += sum_r' G[r'] U[(r',qq)]
#pragma unroll
          for (int t = 0; t < TILES; ++t) {
            f32x16 D;
#pragma unroll
            for (int vv = 0; vv < 16; ++vv) D[vv] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
              D = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(t * KS + s) * ROWP + lane], vin[s], D, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int sp = g + 4 * t;
              if (sp < SR) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) df[qq] += Gs[sp] * D[4 * g + qq];
              }
            }
          }
          // (2) adjoint: dv[l] += sum_(x,qq) A2 * (f[qq] G[x])
          {
            f32x16 D;
#pragma unroll
            for (int vv = 0; vv < 16; ++vv) D[vv] = 0.f;
#pragma unroll
            for (int xx = 0; xx < SR; ++xx) {
#pragma unroll
              for (int qq = 0; qq < 4; ++qq) {
                const int uni = ((xx >> 2) * KS) * ROWP + (((2 * xx) & 7) << 2) + qq;
                const float av = row_ok ? ap[lane_base + uni] : lds[zero_slot];
                D = __builtin_amdgcn_mfma_f32_32x32x2f32(av, f[qq] * Gs[xx], D, 0, 0, 0);
              }
            }
#pragma unroll
            for (int vv = 0; vv < 16; ++vv) {
              const int sp = (vv & 3) + 4 * (vv >> 2);
              if (sp < SR) {
                if (a == 0) d0[sp] += D[vv]; else d1[sp] += D[vv];
              }
            }
          }
          // (3) dCore: transpose v (features l = 2s + h) and Z = f (x) G (features ((x&3)<<3)|(qq<<1)|(r'&1))
          f32x16 Vt;
#pragma unroll
          for (int vv = 0; vv < 16; ++vv) Vt[vv] = 0.f;
#pragma unroll
          for (int s = 0; s < SR; ++s)
            Vt = __builtin_amdgcn_mfma_f32_32x32x2f32(vin[s], (2 * s + h == wl) ? 1.f : 0.f, Vt, 0, 0, 0);
          float* dc = lds + p.dacc_off[c];
#pragma unroll
          for (int tz = 0; tz < TILES; ++tz) {
            f32x16 Zt;
#pragma unroll
            for (int vv = 0; vv < 16; ++vv) Zt[vv] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
              const int xx = 4 * tz + (s2 >> 2), qq = s2 & 3;
              if (xx < SR)
                Zt = __builtin_amdgcn_mfma_f32_32x32x2f32(f[qq] * Gs[xx], (2 * s2 + h == wl) ? 1.f : 0.f, Zt, 0, 0, 0);
            }
            f32x16 acc;
#pragma unroll
            for (int vv = 0; vv < 16; ++vv) acc[vv] = 0.f;
#pragma unroll
            for (int vv = 0; vv < 16; ++vv)
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Zt[vv], Vt[vv], acc, 0, 0, 0);
            // acc: column = l (lane & 31), rows = Z feature n = (vv&3) + 8*(vv>>2) + 4h
            if (wl < R) {
#pragma unroll
              for (int vv = 0; vv < 16; ++vv) {
                const int nf = (vv & 3) + 8 * (vv >> 2) + 4 * h;
                const int rp = 8 * tz + 2 * (nf >> 3) + (nf & 1), qq = (nf >> 1) & 3;
                if (rp < R && qq < p.qc) atomicAdd(&dc[((o * R + wl) * R + rp) * p.qc + qq], acc[vv]);
              }
            }
          }
        }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) df[qq] = half_sum(df[qq]);
      write_dx(c, df);
#pragma unroll
      for (int s = 0; s < SR; ++s) { G0[s] = d0[s]; G1[s] = d1[s]; }
      oacc_out = oacc_in;
    }

    // ---------------- first core
    {
      load_features(fs, 0, lane, f);
      float df[4] = {0.f, 0.f, 0.f, 0.f};
      float* d0p = lds + p.dacc_off[0];
#pragma unroll
      for (int s = 0; s < SR; ++s) {
        const float* cp = lds + p.first_off + (2 * s + h) * 4;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          df[qq] += G0[s] * cp[qq];
          if (qq < p.qc) atomicAdd(&d0p[(2 * s + h) * p.qc + qq], G0[s] * f[qq]);
        }
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) df[qq] = half_sum(df[qq]);
      write_dx(0, df);
    }
  }

  // ---- flush the workgroup's dCore accumulator
  __syncthreads();
  for (int c = 0; c < p.n; ++c) {
    const int E = p.dacc_off[c + 1] - p.dacc_off[c];
    const float* src = lds + p.dacc_off[c];
    for (int e = tid; e < E; e += 256) atomicAdd(&p.dcore[c][e], src[e]);
  }
}

// ------------------------------------------------------------------------- backward, second version
// The same adjoint sweep on v_mfma_f32_16x16x4_f32 (exact f32, 32 cycles): with R <= 16 the 32-row tiles of the first
// version were half empty in two of its three products, its two transposes ran ON the matrix core (identity
// operands: 40 of the 120 MFMAs per core and state) and every tile of dCore went through 32 conflicting ds_add_f32
// per lane (PMC at r = 16: matrix pipe 26 % busy, LDS array 41 % busy, 537 us for 115 200 windows = 11x the forward).
// Layout: a lane is (window wl = lane % 16, k group g = lane / 16) of a 16-window tile; a wave carries two tiles.
//   state v[l, w]           : B-operand layout, register s of lane (wl, g) holds l = 4 s + g
//   U = core x v            : M = (r', qq) in 16-row tiles (row i <-> r' = 4 mt + (i >> 2), qq = i & 3), K = l; the
//                             accumulator gives lane (wl, g) the rows r' = 4 mt + g, qq = 0..3: the lane-local sum
//                             over qq IS the next state register s = mt - the chain stays in registers
//   adjoint dv and df       : ONE product W = core^T x G with the bond legs exchanged (rows (l, qq), K = r', B = G from
//                             its state registers; second LDS pack of the cores): dv[l] = sum_qq f[qq] W[(l, qq)] lands in
//                             state layout like the forward's epilogue, df[qq] = sum_l v[l] W[(l, qq)]
//   dCore += Z^T v          : windows are k: Z = f (x) G and v go through per-wave LDS tiles [window][feature]
//                             (b128 row writes, conflict-free b32 fragment reads); the 16x16 tiles of a core's
//                             gradient are added to a workgroup accumulator kept in ACCUMULATOR layout
//                             [tile][reg][lane] (conflict-free ds_add_f32), re-ordered once at the final flush
//   first / last core       : their gradients are lane-local sums kept in registers across all window groups
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int V> struct sbs_ic { static constexpr int value = V; };

template <int CTRL>
__device__ __forceinline__ float sbs_dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sbs_row_sum16(float v) {   // sum over the 16 lanes of a row, in every lane
  v = sbs_dpp_add<0x128>(v);
  v = sbs_dpp_add<0x124>(v);
  v = sbs_dpp_add<0x122>(v);
  return sbs_dpp_add<0x121>(v);
}
__device__ __forceinline__ float sbs_group_sum(float v) {   // sum over the four k groups (lanes wl, wl+16, wl+32, wl+48)
  const int iv = __float_as_int(v);
  const int2v r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
  return half_sum(__int_as_float(r[0]) + __int_as_float(r[1]));
}

// One fragment element of core c for pack position e (ADJ: the adjoint pack, bond legs exchanged).
template <int R, bool ADJ>
__device__ __forceinline__ float pack16_element(const SbsMP& p, int c, int e) {
  constexpr int MT = R / 4, KS = R / 4;
  const int ln = e & 63, i = ln & 15, kg = ln >> 4;
  int t2 = e >> 6;
  const int s = t2 % KS; t2 /= KS;
  const int mt = t2 % MT;
  const int o = t2 / MT;
  const int l = ADJ ? 4 * mt + (i >> 2) : 4 * s + kg;
  const int rp = ADJ ? 4 * s + kg : 4 * mt + (i >> 2);
  const int qq = i & 3;
  return (qq < p.qc && l < p.bl[c] && rp < p.br[c]) ? p.core[c][(long long)((o * p.bl[c] + l) * p.br[c] + rp) * p.qc + qq] : 0.f;
}

// NMID > 0: at most NMID middle cores with at most two outputs each: every global load of the two packs is issued
// before the first LDS store (one round trip for the whole string instead of one per core and pack: the packing was
// 10 us of a 230 us kernel at bond 16).  NMID == 0: any string, core by core.
// FWDPACK = false: only the adjoint packs (a backward that takes its forward states from the training forward never
// applies the cores forwards)
template <int R, int NMID, bool FWDPACK = true>
__device__ __forceinline__ void pack_cores16(float* lds, const SbsMP& p, int tid) {
  constexpr int MT = R / 4, KS = R / 4, PER_O = MT * KS * 64, PER = (PER_O + 255) / 256;
  if constexpr (NMID > 0) {
    float va[NMID][2][PER], vb[NMID][2][PER];
#pragma unroll
    for (int m = 0; m < NMID; ++m)
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const int c = m + 1, e = tid + 256 * j;
          const bool live = c + 1 < p.n && o < p.o[c] && e < PER_O;
          va[m][o][j] = (FWDPACK && live) ? pack16_element<R, false>(p, c, o * PER_O + e) : 0.f;
          vb[m][o][j] = live ? pack16_element<R, true>(p, c, o * PER_O + e) : 0.f;
        }
#pragma unroll
    for (int m = 0; m < NMID; ++m)
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const int c = m + 1, e = tid + 256 * j;
          if (c + 1 < p.n && o < p.o[c] && e < PER_O) {
            if (FWDPACK) lds[p.apack_off[c] + o * PER_O + e] = va[m][o][j];
            lds[p.apack2_off[c] + o * PER_O + e] = vb[m][o][j];
          }
        }
  } else {
    for (int c = 1; c + 1 < p.n; ++c) {
      const int oc = p.o[c];
      for (int e = tid; e < oc * PER_O; e += 256) {
        if (FWDPACK) lds[p.apack_off[c] + e] = pack16_element<R, false>(p, c, e);
        lds[p.apack2_off[c] + e] = pack16_element<R, true>(p, c, e);   // the same fragment order, rows <-> (l, qq), k <-> r'
      }
    }
  }
  for (int e = tid; e < R * 4; e += 256) {
    const int rr = e >> 2, qq = e & 3;
    lds[p.first_off + e] = (qq < p.qc && rr < p.br[0]) ? p.core[0][(long long)rr * p.qc + qq] : 0.f;
    lds[p.last_off + e] = (qq < p.qc && rr < p.bl[p.n - 1]) ? p.core[p.n - 1][(long long)rr * p.last_stride + qq] : 0.f;
  }
}

// NC > 0: strings of at most NC cores (NC = 9: the reference's snakes): every middle core's gradient tiles are REGISTER
// accumulators across all window groups of the wave (slot c - 1; the one core that may have two outputs uses slot
// NC - 2 for its second), selected by a switch on the (runtime) core index around the 16 adds only - PMC showed the
// per-group ds_add_f32 of the NC == 0 form keeping the LDS array busy 48 % of the kernel (~200 cycles per
// wave-instruction, 176 of them per window group and wave) with the matrix pipe at 13 %.  (Unrolling the whole way
// back over the cores instead made the kernel 100 KB of code: it streamed through the 64 KB instruction cache once
// per window group.)
#ifdef DCTN_STAMPS
// diagnostic build only: wall-clock phase stamps of the first window group of every workgroup's wave 0
__device__ unsigned long long sbs_stamps[2048 * 32];
#define SBS_STAMP(SLOT)                                                                          \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    unsigned long long t_;                                                                       \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
    if (threadIdx.x == 0 && first_group) sbs_stamps[(long long)blockIdx.x * 32 + (SLOT)] = t_;   \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
#else
#define SBS_STAMP(SLOT) do { } while (0)
#endif

// ONECH: one input channel (the multi-channel feature products and their gradients are not compiled: 20 KB of the
// bond-16 kernel's 77 KB, which no longer fits the 64 KB instruction cache)
// SAVED: `states` already holds every core's input state (the training forward stored them, convsbs_fwd_mfma_k's
// `save`): the forward sweep is skipped, the last core's input state is loaded like the others
template <int R, int NC, int NT, int ONECH, bool SAVED>
__global__ __launch_bounds__(256, (R == 4 ? 4 : R == 8 ? 2 : 1)) void convsbs_bwd_mfma16_k(const float* __restrict__ x,
                                                            const float* __restrict__ dY,
                                                            float* __restrict__ states,
                                                            float* __restrict__ gxw, SbsMP p, int need_dx) {
  constexpr int SN = R / 4, MT = R / 4, KS = R / 4, KA = R;
  constexpr int ZROW = 4 * R + 16;   // floats per window row of the Z tile: rows 16 banks apart
  constexpr int VROW = 17;           // floats per window row of the V tile
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wl = lane & 15, g = lane >> 4, wv = tid >> 6;
  bool first_group = true;
  (void)first_group;
  SBS_STAMP(0);
  pack_cores16<R, (NC > 2 ? NC - 2 : 0), !SAVED>(lds, p, tid);
  if constexpr (NC == 0) {   // (NC > 0: the accumulator region lies over the packs and is zeroed after the sweep)
    const int z0 = p.dacc_off[0], z1 = p.dacc_off[p.n];
    for (int e = z0 + tid; e < z1; e += 256) lds[e] = 0.f;
  }
  __syncthreads();
  constexpr int WPG = 16 * NT;   // windows per wave iteration: NT tiles of 16
  float* fs = lds + p.fs_off + wv * p.n * 4 * WPG * (ONECH == 2 ? 2 : 1);   // (mode 2: products, then raw values)
  float* zt = lds + p.zt_off + wv * 16 * ZROW;
  float* vt = lds + p.vt_off + wv * 16 * VROW;
  float dfirst[SN][4], dlast[SN][4];   // gradients of the first / last core: lane-local over all its windows
#pragma unroll
  for (int s = 0; s < SN; ++s)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) { dfirst[s][qq] = 0.f; dlast[s][qq] = 0.f; }
  constexpr int NSLOT = NC > 2 ? NC - 1 : 1;
  f32x4 dreg[NSLOT][MT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) dreg[i][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long long wave = (long long)blockIdx.x * 4 + wv;
  const long long nwaves = (long long)gridDim.x * 4;
  const int hw = p.Ho * p.Wo;
  SBS_STAMP(1);
  for (long long grp = wave; grp < p.ngroups; grp += nwaves) {
    // the wave's 16 NT windows: staging uses one lane per window, the sweep's tile t holds windows 16 t + wl
    long long wt[NT], bt[NT];
    int hot[NT], wot[NT];
    bool vt_ok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      wt[t] = grp * WPG + 16 * t + wl;
      vt_ok[t] = wt[t] < p.Wn;
      const long long ww = vt_ok[t] ? wt[t] : 0;
      bt[t] = ww / hw;
      const int rem = (int)(ww - bt[t] * hw);
      hot[t] = rem / p.Wo;
      wot[t] = rem - hot[t] * p.Wo;
    }
    {
      const long long w = grp * WPG + (lane & (WPG - 1));
      const bool valid = w < p.Wn;
      const long long ww = valid ? w : 0;
      const long long b = ww / hw;
      const int rem = (int)(ww - b * hw);
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      stage_features<WPG, 9, ONECH>(x, p, b, ho, wo, valid, fs, lane);   // a 9-core string's pixels in one round trip
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    SBS_STAMP(2);
    float dy[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      dy[0][t] = vt_ok[t] ? dY[wt[t] * p.ostride + p.obase] : 0.f;
      dy[1][t] = (vt_ok[t] && p.Otot > 1) ? dY[wt[t] * p.ostride + p.obase + p.ostep] : 0.f;
    }
    float f[4][NT];
    auto load_f = [&](int c) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) f[qq][t] = fs[(c * 4 + qq) * WPG + 16 * t + wl];
    };
    auto write_dx = [&](int c, const float (&df)[4][NT]) {
      if (!need_dx || g != 0) return;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!vt_ok[t]) continue;
        if constexpr (ONECH == 2) {   // two channels of two values: products index (d, e) = 2 d + e, raw values from LDS
          const float* xsp = fs + (p.n * 4 + c * 4) * WPG + 16 * t + wl;
          const float x00 = xsp[0], x01 = xsp[WPG], x10 = xsp[2 * WPG], x11 = xsp[3 * WPG];
          const float gx[4] = {df[0][t] * x10 + df[1][t] * x11, df[2][t] * x10 + df[3][t] * x11,    // channel 0, d = 0, 1
                               df[0][t] * x00 + df[2][t] * x01, df[1][t] * x00 + df[3][t] * x01};   // channel 1, e = 0, 1
          // (slices after the first add to what is there: the four old values in ONE round of loads - `*gp = accum ? *gp + v : v`
          // per element was a load and a wait on the spot per element, 36+ dependent round trips per group of windows)
          float* gp0 = gxw + (long long)(c * 4) * p.Wn + wt[t];
          float old[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.accum) {
#pragma unroll
            for (int k = 0; k < 4; ++k) old[k] = gp0[(long long)k * p.Wn];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) gp0[(long long)k * p.Wn] = old[k] + gx[k];
          continue;
        }
        if (ONECH == 1 || p.C == 1) {   // one channel: the feature IS the pixel's value index (no integer divisions, no re-reads of x)
          float* gp0 = gxw + (long long)(c * p.q) * p.Wn + wt[t];
          float old[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.accum) {
#pragma unroll
            for (int qv = 0; qv < 4; ++qv) old[qv] = gp0[(long long)(qv < p.q ? qv : 0) * p.Wn];
          }
#pragma unroll
          for (int qv = 0; qv < 4; ++qv)
            if (qv < p.q) gp0[(long long)qv * p.Wn] = old[qv] + df[qv][t];
          continue;
        }
        if constexpr (ONECH == 0)
        for (int ch = 0; ch < p.C; ++ch)
          for (int qv = 0; qv < p.q; ++qv) {
            float gsum = 0.f;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              if (qq >= p.qc || p.digit[qq][ch] != qv) continue;
              float pr = 1.f;
              for (int c2 = 0; c2 < p.C; ++c2)
                if (c2 != ch)
                  pr *= x[c2 * p.s[0] + bt[t] * p.s[1] + (long long)(hot[t] + p.ph[c]) * p.s[2] +
                          (long long)(wot[t] + p.pw[c]) * p.s[3] + p.digit[qq][c2] * p.s[4]];
              gsum += df[qq][t] * pr;
            }
            float* gp = gxw + (long long)((c * p.C + ch) * p.q + qv) * p.Wn + wt[t];
            *gp = p.accum ? *gp + gsum : gsum;
          }
      }
    };
    auto store_state = [&](int c, int oacc, const float (&a)[2][SN][NT]) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!vt_ok[t]) continue;
#pragma unroll
        for (int s = 0; s < SN; ++s) {
          states[(p.st_off[c] + 4 * s + g) * p.Wn + wt[t]] = a[0][s][t];
          if (oacc > 1) states[(p.st_off[c] + R + 4 * s + g) * p.Wn + wt[t]] = a[1][s][t];
        }
      }
    };
    // U tiles of core c, output o, applied to state `vin`: D[mt][t] (rows r' = 4 mt + g, qq = register)
    auto u_tiles = [&](int c, int o, const float (&vin)[SN][NT], f32x4 (&D)[MT][NT]) {
      const float* ap = lds + p.apack_off[c] + o * MT * KS * 64 + lane;
      float av[MT][KS];   // the whole A operand first: the LDS round trips overlap instead of one per MFMA pair
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < KS; ++s) av[mt][s] = ap[(mt * KS + s) * 64];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int t = 0; t < NT; ++t) D[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int t = 0; t < NT; ++t) D[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][s], vin[s][t], D[mt][t], 0, 0, 0);
      }
    };

    // ---------------- forward sweep (input state of every core stored for the way back)
    float v[2][SN][NT];
    int oacc = 1;
    if constexpr (SAVED) {
      for (int c = 1; c + 1 < p.n; ++c) oacc *= p.o[c];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < SN; ++s) {
          const float l0 = states[(p.st_off[p.n - 1] + 4 * s + g) * p.Wn + (vt_ok[t] ? wt[t] : 0)];   // (clamped address + select)
          const float l1 = states[(p.st_off[p.n - 1] + (oacc > 1 ? R : 0) + 4 * s + g) * p.Wn + (vt_ok[t] ? wt[t] : 0)];
          v[0][s][t] = vt_ok[t] ? l0 : 0.f;
          v[1][s][t] = (vt_ok[t] && oacc > 1) ? l1 : 0.f;
        }
    } else {
      load_f(0);
  #pragma unroll
      for (int s = 0; s < SN; ++s) {
        const float* cp = lds + p.first_off + (4 * s + g) * 4;
  #pragma unroll
        for (int t = 0; t < NT; ++t) {
          v[0][s][t] = cp[0] * f[0][t] + cp[1] * f[1][t] + cp[2] * f[2][t] + cp[3] * f[3][t];
          v[1][s][t] = 0.f;
        }
      }
      for (int c = 1; c + 1 < p.n; ++c) {
        store_state(c, oacc, v);
        const int oc = p.o[c];
        load_f(c);
        float nv[2][SN][NT];
  #pragma unroll
        for (int s = 0; s < SN; ++s)
  #pragma unroll
          for (int t = 0; t < NT; ++t) { nv[0][s][t] = 0.f; nv[1][s][t] = 0.f; }
        auto forward_pair = [&](auto A_, auto O_) {   // (a, o) as constants, as in the way back
          constexpr int a = decltype(A_)::value, o = decltype(O_)::value;
          f32x4 D[MT][NT];
          u_tiles(c, o, v[a], D);
  #pragma unroll
          for (int mt = 0; mt < MT; ++mt)
  #pragma unroll
            for (int t = 0; t < NT; ++t)
              nv[(a == 0 && o == 0) ? 0 : 1][mt][t] =
                  f[0][t] * D[mt][t][0] + f[1][t] * D[mt][t][1] + f[2][t] * D[mt][t][2] + f[3][t] * D[mt][t][3];
        };
        forward_pair(sbs_ic<0>{}, sbs_ic<0>{});
        if (oc > 1) forward_pair(sbs_ic<0>{}, sbs_ic<1>{});
        if (oacc > 1) forward_pair(sbs_ic<1>{}, sbs_ic<0>{});
  #pragma unroll
        for (int s = 0; s < SN; ++s)
  #pragma unroll
          for (int t = 0; t < NT; ++t) { v[0][s][t] = nv[0][s][t]; v[1][s][t] = nv[1][s][t]; }
        oacc *= oc;
      }

    }

    SBS_STAMP(3);
    // The way back reads the stored input state of core c while core c + 1 is being processed: the loads of the
    // NEXT core's states are issued one core ahead (a state row is a fresh line; with one wave per SIMD nothing
    // else hides the ~2 us round trip, 8 of them per window group).
    float vnext[2][SN][NT];
    int oacc_pf = oacc;   // number of input states of the core whose states are in flight
    auto prefetch_state = [&](int c, int nstates) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < SN; ++s) {
          // clamped address + select: as predicated loads each of these was a branch with a wait behind it - the prefetch
          // one core ahead waited for itself
          const float l0 = states[(p.st_off[c] + 4 * s + g) * p.Wn + (vt_ok[t] ? wt[t] : 0)];
          const float l1 = states[(p.st_off[c] + (nstates > 1 ? R : 0) + 4 * s + g) * p.Wn + (vt_ok[t] ? wt[t] : 0)];
          vnext[0][s][t] = vt_ok[t] ? l0 : 0.f;
          vnext[1][s][t] = (vt_ok[t] && nstates > 1) ? l1 : 0.f;
        }
    };
    if (p.n > 2) {
      oacc_pf = oacc / p.o[p.n - 2];
      prefetch_state(p.n - 2, oacc_pf);
    }
    // ---------------- last core
    float G[2][SN][NT];
    {
      load_f(p.n - 1);
      float df[4][NT];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int t = 0; t < NT; ++t) df[qq][t] = 0.f;
#pragma unroll
      for (int s = 0; s < SN; ++s) {
        const float* cp = lds + p.last_off + (4 * s + g) * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float tl = cp[0] * f[0][t] + cp[1] * f[1][t] + cp[2] * f[2][t] + cp[3] * f[3][t];
          G[0][s][t] = dy[0][t] * tl;
          G[1][s][t] = dy[1][t] * tl;
          const float u = dy[0][t] * v[0][s][t] + dy[1][t] * v[1][s][t];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            df[qq][t] += u * cp[qq];
            dlast[s][qq] += u * f[qq][t];
          }
        }
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int t = 0; t < NT; ++t) df[qq][t] = sbs_group_sum(df[qq][t]);
      write_dx(p.n - 1, df);
    }

    SBS_STAMP(4);
    // ---------------- middle cores, right to left
    int oacc_out = oacc;
    for (int c = p.n - 2; c >= 1; --c) {
      const int oc = p.o[c];
      const int oacc_in = oacc_out / oc;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < SN; ++s) { v[0][s][t] = vnext[0][s][t]; v[1][s][t] = vnext[1][s][t]; }
      if (c > 1) prefetch_state(c - 1, oacc_in / p.o[c - 1]);
      load_f(c);
      float df[4][NT];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int t = 0; t < NT; ++t) df[qq][t] = 0.f;
      float d[2][SN][NT];
#pragma unroll
      for (int s = 0; s < SN; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) { d[0][s][t] = 0.f; d[1][s][t] = 0.f; }
      // (input state a, output o) -> adjoint state a oc + o; at most two states, so the four (a, o) bodies are written
      // out with a and o as constants: which state / adjoint register a body touches is then fixed at compile time (the
      // runtime form picked them with ~50 v_cndmask per body, a third of its vector instructions)
      auto adjoint_pair = [&](auto A_, auto O_) {
          constexpr int a = decltype(A_)::value, o = decltype(O_)::value;
          constexpr int gi = (a == 0 && o == 0) ? 0 : 1;   // a oc + o is 0 only for (0, 0); otherwise 1 (two states at most)
          const float (&Gs)[SN][NT] = G[gi];
          const float (&vin)[SN][NT] = v[a];
          // (1)+(2) ONE product serves d/d(features) and the adjoint: W[(l, qq), w] = sum_r' core[o, l, r', qq] G[r', w]
          // (the forward product with the core's bond legs exchanged: rows i <-> l = 4 mt + (i >> 2), qq = i & 3, k = r',
          // B = G straight from its state registers).  The accumulator gives lane (w, g) the rows l = 4 mt + g, qq = 0..3:
          //   dv[l]  = sum_qq f[qq] W[(l, qq)]   lane-local, lands in state register s = mt;
          //   df[qq] = sum_l  v[l]  W[(l, qq)]   lane-local over the lane's l, summed over the k groups after the loop.
          // (The first version computed U = core x v again for df and a separate K = 4R product for dv: 64 MFMAs, now 32.)
          {
            const float* ap2 = lds + p.apack2_off[c] + o * MT * KS * 64 + lane;
            float av[MT][KS];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int s = 0; s < KS; ++s) av[mt][s] = ap2[(mt * KS + s) * 64];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              f32x4 Wt[NT];
#pragma unroll
              for (int t = 0; t < NT; ++t) Wt[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) Wt[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][s], Gs[s][t], Wt[t], 0, 0, 0);
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                const float dvl = f[0][t] * Wt[t][0] + f[1][t] * Wt[t][1] + f[2][t] * Wt[t][2] + f[3][t] * Wt[t][3];
                d[a][mt][t] += dvl;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) df[qq][t] += vin[mt][t] * Wt[t][qq];
              }
            }
          }
          // (3) dCore[o, l, r', qq] += sum_w v[l, w] f[qq, w] G[r', w]: windows on k through the wave's LDS tiles
          {
            f32x4 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();   // the previous tile's fragment reads are done
#pragma unroll
              for (int s = 0; s < SN; ++s) {
                // (the 16-byte slot inside a 16-float block is XORed with bits 1..2 of the row, the V rows are 17 floats
                // long: without either, the 8 rows of one parity land on the same banks in both stores)
                *reinterpret_cast<f32x4*>(zt + wl * ZROW + 16 * s + 4 * (g ^ ((wl >> 1) & 3))) =
                    f32x4{f[0][t] * Gs[s][t], f[1][t] * Gs[s][t], f[2][t] * Gs[s][t], f[3][t] * Gs[s][t]};
                vt[wl * VROW + 4 * s + g] = vin[s][t];
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              float bv[4], av[4][MT];   // all fragments of the tile first, then its MFMAs
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) {
                bv[ks] = vt[(4 * ks + g) * VROW + wl];                                          // V[w = 4 ks + g][l = wl]
                const int zc = wl ^ (4 * ((2 * ks + (g >> 1)) & 3));                            // the row's slot swizzle
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[ks][mt] = zt[(4 * ks + g) * ZROW + 16 * mt + zc];   // Z[w][feature 16 mt + wl]
              }
#pragma unroll
              for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks][mt], bv[ks], acc[mt], 0, 0, 0);
            }
            if constexpr (NC > 0) {
              const int slot = o == 0 ? c - 1 : NC - 2;   // wave-uniform
#define SBS_ACC_CASE(I)                                                        \
  case I:                                                                      \
    if constexpr (I < NSLOT) {                                                 \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) dreg[I][mt] += acc[mt]; \
    }                                                                          \
    break;
              switch (slot) {
                SBS_ACC_CASE(0) SBS_ACC_CASE(1) SBS_ACC_CASE(2) SBS_ACC_CASE(3)
                SBS_ACC_CASE(4) SBS_ACC_CASE(5) SBS_ACC_CASE(6) SBS_ACC_CASE(7)
              }
#undef SBS_ACC_CASE
            } else {
              float* dc = lds + p.dacc_off[c] + o * MT * 256 + lane;
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int vv = 0; vv < 4; ++vv) atomicAdd(&dc[(mt * 4 + vv) * 64], acc[mt][vv]);
            }
          }
      };
      adjoint_pair(sbs_ic<0>{}, sbs_ic<0>{});
      if (oc > 1) adjoint_pair(sbs_ic<0>{}, sbs_ic<1>{});        // (the string has at most two output values in all,
      if (oacc_in > 1) adjoint_pair(sbs_ic<1>{}, sbs_ic<0>{});   //  so (1, 1) does not occur)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int t = 0; t < NT; ++t) df[qq][t] = sbs_group_sum(df[qq][t]);
      write_dx(c, df);
#pragma unroll
      for (int s = 0; s < SN; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) { G[0][s][t] = d[0][s][t]; G[1][s][t] = d[1][s][t]; }
      oacc_out = oacc_in;
      SBS_STAMP(5 + (p.n - 2 - c));
    }

    // ---------------- first core
    {
      load_f(0);
      float df[4][NT];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int t = 0; t < NT; ++t) df[qq][t] = 0.f;
#pragma unroll
      for (int s = 0; s < SN; ++s) {
        const float* cp = lds + p.first_off + (4 * s + g) * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            df[qq][t] += G[0][s][t] * cp[qq];
            dfirst[s][qq] += G[0][s][t] * f[qq][t];
          }
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int t = 0; t < NT; ++t) df[qq][t] = sbs_group_sum(df[qq][t]);
      write_dx(0, df);
    }
    SBS_STAMP(20);
    first_group = false;
  }
  first_group = true;
  SBS_STAMP(21);

  if constexpr (NC > 0) {   // the register accumulators join the workgroup's LDS accumulator once
    // one wave after the other (wave 0 stores, the others read-modify-write whole tiles, a barrier between the waves):
    // the record a workgroup writes is then the same bits in every run, and with the fixed-order reduce so are the
    // gradients.  (The first barrier also says every wave is done with the packs the accumulator region lies over.)
    float d0s[SN][4], dls[SN][4];
#pragma unroll
    for (int s = 0; s < SN; ++s)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        d0s[s][qq] = sbs_row_sum16(dfirst[s][qq]);
        dls[s][qq] = sbs_row_sum16(dlast[s][qq]);
      }
    auto join_tiles = [&](float* dc, const f32x4 (&regs)[MT], bool first) {
      float cur[MT][4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) cur[mt][vv] = first ? 0.f : dc[(mt * 4 + vv) * 64];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) dc[(mt * 4 + vv) * 64] = cur[mt][vv] + regs[mt][vv];
    };
    for (int turn = 0; turn < 4; ++turn) {
      __syncthreads();
      if (wv != turn) continue;
#pragma unroll
      for (int c = 1; c + 1 < NC; ++c) {
        if (c + 1 >= p.n) continue;   // shorter strings
        float* dc = lds + p.dacc_off[c] + lane;
        join_tiles(dc, dreg[c - 1], turn == 0);
        if (p.o[c] > 1) join_tiles(dc + MT * 256, dreg[NC - 2], turn == 0);
      }
      if (wl == 0) {   // first / last core: one lane per k group holds the sum over the group's 16 window lanes
        float* d0p = lds + p.dacc_off[0];
        float* dlp = lds + p.dacc_off[p.n - 1];
#pragma unroll
        for (int s = 0; s < SN; ++s)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
            if (qq < p.qc) {
              const int e = (4 * s + g) * p.qc + qq;
              d0p[e] = (turn == 0 ? 0.f : d0p[e]) + d0s[s][qq];
              dlp[e] = (turn == 0 ? 0.f : dlp[e]) + dls[s][qq];
            }
      }
    }
  } else {
    // ---- first / last core: sum the 16 window lanes of a k group, one lane per group adds to the LDS accumulator
    float* d0p = lds + p.dacc_off[0];
    float* dlp = lds + p.dacc_off[p.n - 1];
#pragma unroll
    for (int s = 0; s < SN; ++s)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float a = sbs_row_sum16(dfirst[s][qq]), b = sbs_row_sum16(dlast[s][qq]);
        if (wl == 0 && qq < p.qc) {
          atomicAdd(&d0p[(4 * s + g) * p.qc + qq], a);
          atomicAdd(&dlp[(4 * s + g) * p.qc + qq], b);
        }
      }
  }
  // ---- flush the workgroup's accumulators (middle cores: accumulator layout -> the core's own layout): one record of
  // partial gradients per workgroup, summed in a fixed order by convsbs_dcore_reduce_k (deterministic; 256+ workgroups
  // adding into the same 34 KB with float atomics ran at the contended atomic rate), or atomics when no room was given
  __syncthreads();
  SBS_STAMP(22);
  float* rec = p.partials ? p.partials + (long long)blockIdx.x * p.core_off[p.n] : nullptr;
  for (int c = 0; c < p.n; ++c) {
    const float* src = lds + p.dacc_off[c];
    if (c == 0 || c == p.n - 1) {
      const int E = (c == 0 ? p.br[0] : p.bl[c]) * p.qc;   // [bond index][q]: the rows below the core's bond are a prefix of the padded table
      for (int e = tid; e < E; e += 256) {
        if (rec) rec[p.core_off[c] + e] = src[e];
        else atomicAdd(&p.dcore[c][c == 0 ? e : (e / p.qc) * p.last_stride + e % p.qc], src[e]);
      }
    } else {
      const int E = p.o[c] * MT * 256;
      for (int e = tid; e < E; e += 256) {
        const int ln = e & 63, l = ln & 15, gg = ln >> 4, vv = (e >> 6) & 3;
        const int mt = (e >> 8) % MT, o = (e >> 8) / MT;
        const int rp = 4 * mt + gg;
        if (l < p.bl[c] && rp < p.br[c] && vv < p.qc) {
          const int idx = ((o * p.bl[c] + l) * p.br[c] + rp) * p.qc + vv;
          if (rec) rec[p.core_off[c] + idx] = src[e]; else atomicAdd(&p.dcore[c][idx], src[e]);
        }
      }
    }
  }
  SBS_STAMP(23);
}

#ifdef DCTN_STAMPS
extern "C" int dctn_debug_read_sbs_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sbs_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

// first stage for many records (small cores, many workgroups): records 32 y .. 32 y + 31 -> record `out0 + y`
__global__ __launch_bounds__(256) void convsbs_dcore_prereduce_k(float* __restrict__ partials, int total, int nrec, int out0) {
  const int e = (int)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int r0 = (int)blockIdx.y * 32, r1 = r0 + 32 < nrec ? r0 + 32 : nrec;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += partials[(long long)r * total + e];
    a1 += partials[(long long)(r + 1) * total + e];
    a2 += partials[(long long)(r + 2) * total + e];
    a3 += partials[(long long)(r + 3) * total + e];
  }
  for (; r < r1; ++r) a0 += partials[(long long)r * total + e];
  partials[(long long)(out0 + blockIdx.y) * total + e] = (a0 + a1) + (a2 + a3);
}

// dCore_c[e] = sum over the workgroups' records, fixed order: 64 elements per workgroup, 4 record subsets, LDS join
__global__ __launch_bounds__(256) void convsbs_dcore_reduce_k(SbsMP p, int nrec) {
  __shared__ float red[4][64];
  const int total = p.core_off[p.n];
  const int e = (int)blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < total) {
    int r = sub;
    for (; r + 12 < nrec; r += 16) {
      a0 += p.partials[(long long)r * total + e];
      a1 += p.partials[(long long)(r + 4) * total + e];
      a2 += p.partials[(long long)(r + 8) * total + e];
      a3 += p.partials[(long long)(r + 12) * total + e];
    }
    for (; r < nrec; r += 4) a0 += p.partials[(long long)r * total + e];
  }
  red[sub][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sub == 0 && e < total) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    int c = 0;
    while (c + 1 < p.n && e >= p.core_off[c + 1]) ++c;
    const int el = e - p.core_off[c];
    float* dst = p.dcore[c] + (c == p.n - 1 ? (el / p.qc) * p.last_stride + el % p.qc : el);
    *dst = p.accum ? *dst + v : v;   // (slices of a many-output string: one launch after the other on the stream)
  }
}

}  // namespace

// Family check + parameter block shared by forward and backward; *lds_floats = floats of LDS used
// by the core packs and tables.  DCTN_ERR_UNSUPPORTED when the string is outside the family.
static int sbsm_fill(SbsMP& p, int& R, int& lds_floats, const int64_t xs[5], const void* const* cores, int n,
                     const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                     int C, int B, int H, int W, int q, int dtype) {
  if (dtype != DCTN_F32 || n < 3 || n > SBSM_MAXC) return DCTN_ERR_UNSUPPORTED;
  // any bonds 1..32 (they need not be equal) run on the next of the kernels' tile sizes {4, 8, 16, 32} above the largest:
  // packs and tables are zero beyond a core's own bonds, so are the states, and only the real entries of a gradient are
  // written back
  if (bond_sizes[0] != 1) return DCTN_ERR_UNSUPPORTED;
  int Ra = 0;
  for (int c = 1; c < n; ++c) {
    if (bond_sizes[c] < 1 || bond_sizes[c] > 32) return DCTN_ERR_UNSUPPORTED;
    Ra = bond_sizes[c] > Ra ? bond_sizes[c] : Ra;
  }
  if (Ra < 2) return DCTN_ERR_UNSUPPORTED;
  R = Ra <= 4 ? 4 : Ra <= 8 ? 8 : Ra <= 16 ? 16 : 32;
  p.Ra = Ra;
  for (int c = 0; c < n; ++c) {
    p.bl[c] = c == 0 ? 1 : bond_sizes[c];
    p.br[c] = c == n - 1 ? 1 : bond_sizes[c + 1];
  }
  long long qc = 1;
  for (int c = 0; c < C; ++c) qc *= q;
  if (qc > 4 || C > 2 || q > 4) return DCTN_ERR_UNSUPPORTED;
  for (int qq = 0; qq < 4; ++qq) {
    int t = qq;
    for (int ch = C - 1; ch >= 0; --ch) { p.digit[qq][ch] = (unsigned char)(t % q); t /= q; }
    for (int ch = C; ch < 4; ++ch) p.digit[qq][ch] = 0;
  }
  long long otot = 1;
  for (int c = 0; c < n; ++c) {
    if (out_sizes[c] < 1 || out_sizes[c] > 2) return DCTN_ERR_UNSUPPORTED;
    otot *= out_sizes[c];
  }
  if (otot > 2 || out_sizes[0] != 1 || out_sizes[n - 1] != 1) return DCTN_ERR_UNSUPPORTED;
  p.n = n; p.C = C; p.B = B; p.H = H; p.W = W; p.q = q; p.qc = (int)qc; p.Otot = (int)otot;
  p.ostride = (int)otot; p.obase = 0; p.ostep = 1; p.accum = 0; p.out_accum = 0; p.last_stride = (int)qc; p.save = nullptr;
  int max_h = 0, max_w = 0;
  for (int c = 0; c < n; ++c) {
    p.o[c] = out_sizes[c]; p.ph[c] = pos_h[c]; p.pw[c] = pos_w[c];
    p.core[c] = (const float*)cores[c];
    p.dcore[c] = nullptr;
    p.partials = nullptr;
    max_h = pos_h[c] > max_h ? pos_h[c] : max_h;
    max_w = pos_w[c] > max_w ? pos_w[c] : max_w;
  }
  p.Ho = H - max_h; p.Wo = W - max_w;
  if (p.Ho < 1 || p.Wo < 1) return DCTN_ERR_BAD_SHAPE;
  p.Wn = (long long)B * p.Ho * p.Wo;
  p.ngroups = (p.Wn + 31) / 32;
  for (int i = 0; i < 5; ++i) p.s[i] = xs[i];
  const int KS = R / 2, TILES = R >= 8 ? R / 8 : 1;
  int off = 0;
  for (int c = 1; c + 1 < n; ++c) {
    p.apack_off[c] = off;
    off += p.o[c] * TILES * KS * ROWP;
  }
  p.first_off = off; off += R * 4;
  p.last_off = off; off += R * 4;
  p.fs_off = off; off += 4 * n * 128;
  lds_floats = off;
  return DCTN_OK;
}

// Strings with ONE many-valued middle core (the final string of the reference's ConvSBS classifier, mnist.py:214-224:
// ten labels on core 4) run as slices of at most two output values: slice j takes core m's outputs 2j, 2j + 1 (its
// [o][l][r][q] layout makes a slice a pointer offset), writes columns 2j.. of out and reads those of dY; the gradients of
// the other cores and of x add up over the slices.  Returns the index of that core, -1 when the string has none, -2 when
// the outputs have another shape (the generic sweep takes those).
struct SbsSlice {   // one launch of a string that runs in slices (many-valued core and / or ring); sliced == 0: the whole string
  int sliced, obase, ostride, accum, out_accum, last_stride, ostep;
};

static int convsbs_fwd_mfma_one(const void* x, const int64_t xs[5], const void* const* cores, void* out, int n,
                                const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                                int C, int B, int H, int W, int q, int dtype, hipStream_t st, const SbsSlice& sl,
                                float* save) {
  SbsMP p;
  int R, off;
  const int rcf = sbsm_fill(p, R, off, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  if (rcf != DCTN_OK) return rcf;
  if (sl.sliced) { p.ostride = sl.ostride; p.obase = sl.obase; p.ostep = sl.ostep; p.out_accum = sl.out_accum; p.last_stride = sl.last_stride; }
  if (save && !sl.sliced && R <= 16) {   // input states of cores 1 .. n-1 for the backward (same offsets as its own sweep)
    long long so = 0;
    int oacc = 1;
    for (int c = 0; c < n; ++c) {
      p.st_off[c] = so;
      if (c >= 1) so += (long long)oacc * R;
      oacc *= p.o[c];
    }
    p.st_off[n] = so;
    p.save = save;
  }
  const size_t lds = (size_t)off * sizeof(float);
  if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
  long long blocks = (p.ngroups + 3) / 4;
  long long fwd_per_cu = lds > 0 ? (160 * 1024) / (long long)lds : 8;   // persistent: the core pack is paid once per workgroup
  if (fwd_per_cu < 2) fwd_per_cu = 2;
  if (fwd_per_cu > 8) fwd_per_cu = 8;
  if (blocks > 256 * fwd_per_cu) blocks = 256 * fwd_per_cu;
  const int chmode = p.C == 1 ? 1 : (p.C == 2 && p.q == 2) ? 2 : 0;
#define SBS_LAUNCH_CH(RR, CHV)                                                                    \
  do {                                                                                            \
    (void)hipFuncSetAttribute((const void*)convsbs_fwd_mfma_k<RR, CHV>,                           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
    hipLaunchKernelGGL((convsbs_fwd_mfma_k<RR, CHV>), dim3((unsigned)blocks), dim3(256), lds, st, \
                       (const float*)x, (float*)out, p);                                          \
  } while (0)
#define SBS_LAUNCH(RR)                                                                            \
  do {                                                                                            \
    if (chmode == 1) SBS_LAUNCH_CH(RR, 1); else if (chmode == 2) SBS_LAUNCH_CH(RR, 2); else SBS_LAUNCH_CH(RR, 0); \
  } while (0)
  switch (R) {
    case 4: SBS_LAUNCH(4); break;
    case 8: SBS_LAUNCH(8); break;
    case 16: SBS_LAUNCH(16); break;
    case 32: SBS_LAUNCH(32); break;
  }
#undef SBS_LAUNCH
#undef SBS_LAUNCH_CH
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("convsbs_fwd_mfma_f32");
  return DCTN_OK;
}

// The slices of a string.  One launch handles an open chain with at most two outputs, both on one middle core.  Anything
// else runs as several launches over VIEWS of the same cores:
//   * a ring (bond_sizes[0] > 1): Tr(prod T_c) = sum over the closing bond value l0 of the open chain whose first core is
//     row l0 of core 0 (contiguous in its [o][l][r][q] layout) and whose last core is column l0 of core n-1 (a strided view);
//   * the PAIR core m = the last middle core with more than one output: its outputs go two per launch;
//   * every other core with several outputs is walked one output at a time (the [l][r][q] block of that output).
// The flat output index is row-major over the cores in string order, so a slice's two outputs are prod(o_c, c > m) apart.
// `visit` gets the slice's core views (as element offsets into the cores - the gradients use the same), output sizes, bonds
// and SbsSlice; a non-OK return stops the walk (the first slice decides: all later ones have the same LDS plan).
constexpr int SBSM_MAX_SLICES = 96;   // beyond that the generic sweep's single launch wins
static bool sbsm_whole(int n, const int* out_sizes, const int* bond_sizes) {
  long long otot = 1;
  for (int c = 0; c < n; ++c) otot *= out_sizes[c];
  return bond_sizes[0] == 1 && otot <= 2 && out_sizes[0] == 1 && out_sizes[n - 1] == 1;
}
template <typename F>
static int sbsm_for_slices(int n, const void* const* cores, const int* out_sizes, const int* bond_sizes, int C, int q, F visit) {
  if (n < 3 || n > SBSM_MAXC) return DCTN_ERR_UNSUPPORTED;
  long long otot = 1, qc = 1;
  for (int c = 0; c < n; ++c) {
    if (out_sizes[c] < 1 || out_sizes[c] > 64 || bond_sizes[c] < 1) return DCTN_ERR_UNSUPPORTED;
    otot *= out_sizes[c];
  }
  for (int c = 0; c < C; ++c) qc *= q;
  if (otot > (1 << 20)) return DCTN_ERR_UNSUPPORTED;
  int outs[SBSM_MAXC], bonds[SBSM_MAXC];
  long long coff[SBSM_MAXC];
  for (int c = 0; c < n; ++c) { outs[c] = out_sizes[c]; bonds[c] = bond_sizes[c]; coff[c] = 0; }
  if (sbsm_whole(n, out_sizes, bond_sizes)) {
    const SbsSlice whole{0, 0, 0, 0, 0, 0, 1};
    return visit(coff, outs, bonds, whole);
  }
  const int R0 = bond_sizes[0];   // the closing bond of a ring (1: open chain)
  const bool ring = R0 > 1;
  int m = -1;
  for (int c = 1; c + 1 < n; ++c)
    if (out_sizes[c] > 1) m = c;
  long long ostr[SBSM_MAXC], per_o[SBSM_MAXC], nsl = ring ? R0 : 1;
  for (int c = n - 1, acc = 1; c >= 0; --c) { ostr[c] = acc; acc *= out_sizes[c]; }
  for (int c = 0; c < n; ++c) {
    per_o[c] = (long long)bond_sizes[c] * bond_sizes[(c + 1) % n] * qc;   // a core is [o][l][r][q...]
    nsl *= c == m ? (out_sizes[c] + 1) / 2 : out_sizes[c];
  }
  // (Slices pay a launch and a recomputed prefix each; with the compiled two-channel mode they beat the generic sweep's
  // single launch at every bond - ten labels, C = 2, 61 952 windows, device time of fwd + bwd: bond 2 0.49 ms generic /
  // 0.41 ms sliced, bond 3 0.63 / 0.41, bond 4 1.2 / 0.41; rings bond 2: 0.74 / 0.35, bond 3: 1.29 / 0.48.)
  if (nsl > SBSM_MAX_SLICES) return DCTN_ERR_UNSUPPORTED;
  bonds[0] = 1;   // every slice is an open chain
  int idx[SBSM_MAXC] = {};   // the walked output of every core but the pair core; the first output of the pair core's slice
  bool first = true;
  for (;;) {
    for (int l0 = 0; l0 < (ring ? R0 : 1); ++l0) {
      long long obase = 0;
      for (int c = 0; c < n; ++c) {
        coff[c] = idx[c] * per_o[c];
        obase += idx[c] * ostr[c];
        outs[c] = 1;
      }
      if (m >= 0) outs[m] = out_sizes[m] - idx[m] < 2 ? out_sizes[m] - idx[m] : 2;
      if (ring) {
        coff[0] += (long long)l0 * bond_sizes[1] * qc;
        coff[n - 1] += (long long)l0 * qc;
      }
      const SbsSlice sl{1, (int)obase, (int)otot, !first, l0 > 0, (int)(ring ? R0 * qc : qc), m >= 0 ? (int)ostr[m] : 1};
      const int rc = visit(coff, outs, bonds, sl);
      if (rc != DCTN_OK) return rc;
      first = false;
    }
    int c = n - 1;   // next combination of outputs (the last core runs fastest)
    for (; c >= 0; --c) {
      idx[c] += c == m ? 2 : 1;
      if (idx[c] < out_sizes[c]) break;
      idx[c] = 0;
    }
    if (c < 0) break;
  }
  return DCTN_OK;
}

// Bytes of the forward states a training forward can leave for the backward (0: this string recomputes them - slices,
// bonds above 16, shapes outside the family).
size_t convsbs_saved_states_bytes(int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                                  int C, int B, int H, int W, int q, int dtype) {
  if (dtype != DCTN_F32 || n < 3 || n > SBSM_MAXC) return 0;
  if (!sbsm_whole(n, out_sizes, bond_sizes)) return 0;
  // strings the band-owning backward takes (convsbs_band.hip) recompute the chain in registers: nothing to keep
  if (convsbs_band_covers(n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype)) return 0;
  SbsMP p;
  int R, off;
  const int64_t dummy[5] = {0, 0, 0, 0, 1};
  const void* none[SBSM_MAXC] = {};
  if (sbsm_fill(p, R, off, dummy, none, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype) != DCTN_OK || R > 16) return 0;
  if (R < 8) return 0;   // bond <= 4: storing costs the forward what the backward saves (80 -> 83 us at the cfg4 shape)
  long long so = 0;
  int oacc = 1;
  for (int c = 0; c < n; ++c) {
    if (c >= 1) so += (long long)oacc * R;
    oacc *= p.o[c];
  }
  return (size_t)so * (size_t)p.Wn * sizeof(float);
}

int convsbs_fwd_mfma(const void* x, const int64_t xs[5], const void* const* cores, void* out, int n,
                     const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                     int C, int B, int H, int W, int q, int dtype, hipStream_t st, float* save_states) {
  if (dtype != DCTN_F32) return DCTN_ERR_UNSUPPORTED;
  return sbsm_for_slices(n, cores, out_sizes, bond_sizes, C, q,
                         [&](const long long* coff, const int* outs, const int* bonds, const SbsSlice& sl) {
                           const void* cp[SBSM_MAXC];
                           for (int c = 0; c < n; ++c) cp[c] = (const float*)cores[c] + coff[c];
                           return convsbs_fwd_mfma_one(x, xs, cp, out, n, outs, bonds, pos_h, pos_w, C, B, H, W, q, dtype, st, sl,
                                                       save_states);
                         });
}

// Backward of the same family (R <= 16).  `states` must hold sum_c oacc_c * R floats per window
// (the generic kernels' state region is large enough), `gxw` the per-window feature gradients
// [(c*C + ch)*q + qv][Wn] (may be NULL when dX is not needed), `dcores[c]` zero-initialised float
// accumulators (may be NULL array when no core gradient is needed... the kernel still runs its
// dCore part into LDS only).
static int convsbs_bwd_mfma_one(const void* x, const int64_t xs[5], const void* const* cores, const void* dY,
                                float* states, float* gxw, float* const* dcores, int n, const int* out_sizes,
                                const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W,
                                int q, int dtype, hipStream_t st, float* partials, size_t partial_bytes,
                                const SbsSlice& sl, const float* saved) {
  SbsMP p;
  int R, off;
  const int rcf = sbsm_fill(p, R, off, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  if (rcf != DCTN_OK) return rcf;
  if (R > 16 || !dcores || !states) return DCTN_ERR_UNSUPPORTED;
  if (sl.sliced) { p.ostride = sl.ostride; p.obase = sl.obase; p.ostep = sl.ostep; p.accum = sl.accum; p.last_stride = sl.last_stride; }
  // the training forward's stored states replace this launch's forward sweep (whole strings only: a slice has its own)
  const bool use_saved = saved != nullptr && !sl.sliced;
  if (use_saved) states = const_cast<float*>(saved);
  long long so = 0;
  int oacc = 1;
  for (int c = 0; c < n; ++c) {
    p.st_off[c] = so;
    if (c >= 1) so += (long long)oacc * R;
    oacc *= p.o[c];
    p.dcore[c] = dcores[c];
  }
  p.st_off[n] = so;
  {
    // second version (16x16x4 tiles): its own LDS plan - two packs of every middle core, the accumulators in
    // accumulator layout, the per-wave feature slices and transposition tiles
    const int MT = R / 4, KS = R / 4, KA = R, ZROW = 4 * R + 16;
    int o2 = 0;
    SbsMP q2 = p;
    for (int c = 1; c + 1 < n; ++c) { q2.apack_off[c] = o2; o2 += p.o[c] * MT * KS * 64; }
    for (int c = 1; c + 1 < n; ++c) { q2.apack2_off[c] = o2; o2 += p.o[c] * MT * KS * 64; }
    q2.first_off = o2; o2 += R * 4;
    q2.last_off = o2; o2 += R * 4;
    const int NT16 = 2;   // window tiles per wave iteration (4 was tried for r = 16: 892 bytes of scratch per lane, slower)
    const int chmode = p.C == 1 ? 1 : (p.C == 2 && p.q == 2) ? 2 : 0;   // compiled channel handling (ONECH)
    q2.fs_off = o2; o2 += 4 * n * 4 * 16 * NT16 * (chmode == 2 ? 2 : 1);
    q2.zt_off = o2; o2 += 4 * 16 * ZROW;
    q2.vt_off = o2; o2 += 4 * 16 * 17;
    // the workgroup's dCore accumulator: with the register accumulators (n <= 9) it is only used by the final flush,
    // when the packs are dead, and lies over them; otherwise it is a region of its own
    int o3 = (n <= 9) ? 0 : o2;
    for (int c = 0; c < n; ++c) {
      q2.dacc_off[c] = o3;
      o3 += (c == 0 || c == n - 1) ? ((R * p.qc + 3) / 4 * 4) : p.o[c] * MT * 256;
    }
    q2.dacc_off[n] = o3;
    if (o3 > o2) o2 = o3;
    const size_t lds2 = (size_t)o2 * sizeof(float);
    p.ngroups = (p.Wn + 16 * NT16 - 1) / (16 * NT16);
    q2.ngroups = p.ngroups;
    q2.core_off[0] = 0;
    for (int c = 0; c < n; ++c) {
      q2.core_off[c + 1] = q2.core_off[c] + p.o[c] * p.bl[c] * p.br[c] * p.qc;
    }
    if (lds2 <= DCTN_LDS_BUDGET) {
      long long blocks = (p.ngroups + 3) / 4;
      // latency-bound sweeps: as many workgroups as the LDS plan lets a CU hold (the small-bond strings fit several;
      // 3 600 window groups then run in one round instead of two or four)
      long long per_cu2 = (160 * 1024) / (long long)lds2;
      if (per_cu2 < 1) per_cu2 = 1;
      if (per_cu2 > 8) per_cu2 = 8;
      if (blocks > 256 * per_cu2) blocks = 256 * per_cu2;
      if (blocks > SBS_MAX_PARTIAL_RECORDS - 64) blocks = SBS_MAX_PARTIAL_RECORDS - 64;   // room for the second-stage records
      q2.partials = (partials && partial_bytes >= (size_t)blocks * q2.core_off[n] * sizeof(float)) ? partials : nullptr;
#define SBS_LAUNCH_B16_1(RR, NCV, NTV, CHV, SV)                                                    \
  do {                                                                                            \
    (void)hipFuncSetAttribute((const void*)convsbs_bwd_mfma16_k<RR, NCV, NTV, CHV, SV>,           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);             \
    hipLaunchKernelGGL((convsbs_bwd_mfma16_k<RR, NCV, NTV, CHV, SV>), dim3((unsigned)blocks), dim3(256), lds2, st, \
                       (const float*)x, (const float*)dY, states, gxw, q2, gxw != nullptr);       \
  } while (0)
#define SBS_LAUNCH_B16_S(RR, NCV, NTV, SV)                                                        \
  do {                                                                                            \
    if (chmode == 1) SBS_LAUNCH_B16_1(RR, NCV, NTV, 1, SV);                                       \
    else if (chmode == 2) SBS_LAUNCH_B16_1(RR, NCV, NTV, 2, SV);                                  \
    else SBS_LAUNCH_B16_1(RR, NCV, NTV, 0, SV);                                                   \
  } while (0)
#define SBS_LAUNCH_B16(RR, NCV, NTV)                                                              \
  do {                                                                                            \
    if (use_saved) SBS_LAUNCH_B16_S(RR, NCV, NTV, true); else SBS_LAUNCH_B16_S(RR, NCV, NTV, false); \
  } while (0)
      switch (R) {   // up to 9 cores (mnist.py:189-223): register accumulators; longer strings: LDS accumulators
        case 4: if (n <= 9) SBS_LAUNCH_B16(4, 9, 2); else SBS_LAUNCH_B16(4, 0, 2); break;
        case 8: if (n <= 9) SBS_LAUNCH_B16(8, 9, 2); else SBS_LAUNCH_B16(8, 0, 2); break;
        case 16: if (n <= 9) SBS_LAUNCH_B16(16, 9, 2); else SBS_LAUNCH_B16(16, 0, 2); break;
        default: return DCTN_ERR_UNSUPPORTED;
      }
#undef SBS_LAUNCH_B16
#undef SBS_LAUNCH_B16_S
#undef SBS_LAUNCH_B16_1
      DCTN_CHECK_LAUNCH();
      if (q2.partials) {
        int nrec = (int)blocks;
        const int total = q2.core_off[n], n2 = (nrec + 31) / 32;
        // many records of a small core: two stages (the one-stage form ran 13 workgroups over 768 records: 21 us)
        if (nrec > 64 && partial_bytes >= (size_t)(nrec + n2) * total * sizeof(float)) {
          hipLaunchKernelGGL(convsbs_dcore_prereduce_k, dim3((unsigned)((total + 255) / 256), (unsigned)n2), dim3(256), 0, st,
                             q2.partials, total, nrec, nrec);
          DCTN_CHECK_LAUNCH();
          q2.partials += (long long)nrec * total;
          nrec = n2;
        }
        hipLaunchKernelGGL(convsbs_dcore_reduce_k, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, q2, nrec);
        DCTN_CHECK_LAUNCH();
      }
      dctn_set_last_kernel("convsbs_bwd_mfma_f32");
      return DCTN_OK;
    }
  }
  // first version (32x32x2 tiles, R <= 16): strings whose packs do not fit the second version's LDS plan
  bool full_tiles = true;   // every inner bond equal to the tile size
  for (int c = 1; c < n; ++c) full_tiles = full_tiles && bond_sizes[c] == R;
  if (sl.sliced || !full_tiles || use_saved) return DCTN_ERR_UNSUPPORTED;   // (slices, padded bonds, saved states: second version only)
  p.ngroups = (p.Wn + 31) / 32;
  so = 0;
  oacc = 1;
  int dacc = off;
  for (int c = 0; c < n; ++c) {
    p.st_off[c] = so;
    if (c >= 1) so += (long long)oacc * R;
    oacc *= p.o[c];
    p.dacc_off[c] = dacc;
    const int L = c == 0 ? 1 : R, Rr = c == n - 1 ? 1 : R;
    dacc += p.o[c] * L * Rr * p.qc;
    p.dcore[c] = dcores[c];
  }
  p.st_off[n] = so;
  p.dacc_off[n] = dacc;
  const size_t lds = (size_t)(dacc + 1) * sizeof(float);
  if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
  long long per_cu = (160 * 1024) / (long long)lds;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 2) per_cu = 2;
  long long blocks = (p.ngroups + 3) / 4;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
#define SBS_LAUNCH_B(RR)                                                                          \
  (void)hipFuncSetAttribute((const void*)convsbs_bwd_mfma_k<RR>,                                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
  hipLaunchKernelGGL((convsbs_bwd_mfma_k<RR>), dim3((unsigned)blocks), dim3(256), lds, st,        \
                     (const float*)x, (const float*)dY, states, gxw, p, gxw != nullptr)
  switch (R) {
    case 4: SBS_LAUNCH_B(4); break;
    case 8: SBS_LAUNCH_B(8); break;
    case 16: SBS_LAUNCH_B(16); break;
    default: return DCTN_ERR_UNSUPPORTED;
  }
#undef SBS_LAUNCH_B
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("convsbs_bwd_mfma_f32");
  return DCTN_OK;
}

int convsbs_bwd_mfma(const void* x, const int64_t xs[5], const void* const* cores, const void* dY,
                     float* states, float* gxw, float* const* dcores, int n, const int* out_sizes,
                     const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W,
                     int q, int dtype, hipStream_t st, float* partials, size_t partial_bytes, const float* saved_states) {
  if (dtype != DCTN_F32 || !dcores) return DCTN_ERR_UNSUPPORTED;
  return sbsm_for_slices(n, cores, out_sizes, bond_sizes, C, q,
                         [&](const long long* coff, const int* outs, const int* bonds, const SbsSlice& sl) {
                           const void* cp[SBSM_MAXC];
                           float* dcp[SBSM_MAXC];   // the gradient views follow the core views
                           for (int c = 0; c < n; ++c) {
                             cp[c] = (const float*)cores[c] + coff[c];
                             dcp[c] = dcores[c] + coff[c];
                           }
                           return convsbs_bwd_mfma_one(x, xs, cp, dY, states, gxw, dcp, n, outs, bonds, pos_h, pos_w, C, B, H,
                                                       W, q, dtype, st, partials, partial_bytes, sl, saved_states);
                         });
}
