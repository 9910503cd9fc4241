// MFMA ConvSBS forward sweep: open chain (bond_sizes[0] == 1), uniform internal bond r in
// {4, 8, 16, 32}, q^C <= 4, at most 2 outputs in total, float32 (exact: v_mfma_f32_32x32x2_f32).
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
  long long Wn, ngroups;
  long long s[5];
  int o[SBSM_MAXC], ph[SBSM_MAXC], pw[SBSM_MAXC];
  int apack_off[SBSM_MAXC];   // float offset of core c's packed A operand in LDS (middle cores)
  int dacc_off[SBSM_MAXC + 1];  // backward: float offset of core c's dCore accumulator in LDS
  long long st_off[SBSM_MAXC + 1];  // backward: element offsets of the stored forward states
  float* dcore[SBSM_MAXC];    // backward: global dCore (zero-initialised by the caller)
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
__device__ __forceinline__ void stage_features(const float* __restrict__ x, const SbsMP& p, long long b, int ho,
                                               int wo, bool valid, float* fs, int lane) {
  const float* win = x + b * p.s[1] + (long long)ho * p.s[2] + (long long)wo * p.s[3];
  for (int c0 = 0; c0 < p.n; c0 += 8) {
    float raw[8][2][4];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int c = c0 + cc < p.n ? c0 + cc : p.n - 1;
      const float* base = win + (long long)p.ph[c] * p.s[2] + (long long)p.pw[c] * p.s[3];
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int d = 0; d < 4; ++d)
          raw[cc][ch][d] = (valid && ch < p.C && d < p.q) ? base[ch * p.s[0] + d * p.s[4]] : 1.f;
    }
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
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
          fs[((c0 + cc) * 4 + qq) * 32 + (lane & 31)] = pr;   // both lane halves hold the same window
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
      if (rp < R && qq < p.qc) v = p.core[c][(long long)((o * R + l) * R + rp) * p.qc + qq];
      lds[p.apack_off[c] + e] = v;
    }
  }
  // first core (1, 1, R, qc) as [r'][4]; last core (1, R, 1, qc) as [l][4]
  for (int e = tid; e < R * 4; e += 256) {
    const int rr = e >> 2, qq = e & 3;
    lds[p.first_off + e] = qq < p.qc ? p.core[0][(long long)rr * p.qc + qq] : 0.f;
    lds[p.last_off + e] = qq < p.qc ? p.core[p.n - 1][(long long)rr * p.qc + qq] : 0.f;
  }
}

template <int R>
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
    stage_features(x, p, b, ho, wo, valid, fs, lane);
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
    // ---- middle cores on the matrix pipe
    for (int c = 1; c + 1 < p.n; ++c) {
      const int oc = p.o[c];
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
      out[w * p.Otot] = r0;
      if (p.Otot > 1) out[w * p.Otot + 1] = r1;
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
    const float dy0 = valid ? dY[w * p.Otot] : 0.f;
    const float dy1 = (valid && p.Otot > 1) ? dY[w * p.Otot + 1] : 0.f;
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
        v0[s] = valid ? states[(p.st_off[c] + 2 * s + h) * p.Wn + w] : 0.f;
        v1[s] = (valid && oacc_in > 1) ? states[(p.st_off[c] + R + 2 * s + h) * p.Wn + w] : 0.f;
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

}  // namespace

// Family check + parameter block shared by forward and backward; *lds_floats = floats of LDS used
// by the core packs and tables.  DCTN_ERR_UNSUPPORTED when the string is outside the family.
static int sbsm_fill(SbsMP& p, int& R, int& lds_floats, const int64_t xs[5], const void* const* cores, int n,
                     const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                     int C, int B, int H, int W, int q, int dtype) {
  if (dtype != DCTN_F32 || n < 3 || n > SBSM_MAXC) return DCTN_ERR_UNSUPPORTED;
  R = bond_sizes[1];
  if (bond_sizes[0] != 1 || (R != 4 && R != 8 && R != 16 && R != 32)) return DCTN_ERR_UNSUPPORTED;
  for (int c = 2; c < n; ++c)
    if (bond_sizes[c] != R) return DCTN_ERR_UNSUPPORTED;
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
  int max_h = 0, max_w = 0;
  for (int c = 0; c < n; ++c) {
    p.o[c] = out_sizes[c]; p.ph[c] = pos_h[c]; p.pw[c] = pos_w[c];
    p.core[c] = (const float*)cores[c];
    p.dcore[c] = nullptr;
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

int convsbs_fwd_mfma(const void* x, const int64_t xs[5], const void* const* cores, void* out, int n,
                     const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                     int C, int B, int H, int W, int q, int dtype, hipStream_t st) {
  SbsMP p;
  int R, off;
  const int rcf = sbsm_fill(p, R, off, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  if (rcf != DCTN_OK) return rcf;
  const size_t lds = (size_t)off * sizeof(float);
  if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
  long long blocks = (p.ngroups + 3) / 4;
  if (blocks > 256 * 2) blocks = 256 * 2;   // persistent: the core pack is paid once per workgroup
#define SBS_LAUNCH(RR)                                                                            \
  (void)hipFuncSetAttribute((const void*)convsbs_fwd_mfma_k<RR>,                                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
  hipLaunchKernelGGL((convsbs_fwd_mfma_k<RR>), dim3((unsigned)blocks), dim3(256), lds, st,        \
                     (const float*)x, (float*)out, p)
  switch (R) {
    case 4: SBS_LAUNCH(4); break;
    case 8: SBS_LAUNCH(8); break;
    case 16: SBS_LAUNCH(16); break;
    case 32: SBS_LAUNCH(32); break;
  }
#undef SBS_LAUNCH
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("convsbs_fwd_mfma_f32");
  return DCTN_OK;
}

// Backward of the same family (R <= 16).  `states` must hold sum_c oacc_c * R floats per window
// (the generic kernels' state region is large enough), `gxw` the per-window feature gradients
// [(c*C + ch)*q + qv][Wn] (may be NULL when dX is not needed), `dcores[c]` zero-initialised float
// accumulators (may be NULL array when no core gradient is needed... the kernel still runs its
// dCore part into LDS only).
int convsbs_bwd_mfma(const void* x, const int64_t xs[5], const void* const* cores, const void* dY,
                     float* states, float* gxw, float* const* dcores, int n, const int* out_sizes,
                     const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W,
                     int q, int dtype, hipStream_t st) {
  SbsMP p;
  int R, off;
  const int rcf = sbsm_fill(p, R, off, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  if (rcf != DCTN_OK) return rcf;
  if (R > 16 || !dcores || !states) return DCTN_ERR_UNSUPPORTED;
  long long so = 0;
  int oacc = 1, dacc = off;
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
