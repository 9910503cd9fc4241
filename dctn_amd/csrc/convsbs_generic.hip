// Generic ConvSBS kernels: any string (arbitrary core order, ring or open chain, per-bond sizes,
// outputs on several cores, C channels), float32 / float64 / bf16 storage.
//
// Replaces dctn/conv_sbs.py:258-304 (ConvSBS.forward) and torch autograd through it.
// The reference materialises, per core, a (B,H',W',o,l,r) tensor in HBM and then multiplies the
// chain.  Here one lane owns one window and sweeps the string left to right carrying only the
// state v[o_acc][bond] in its LDS column; the per-window matrices T_c are never written out.
// A ring (bond_sizes[0] > 1) is closed by running the sweep once per value of the traced bond.
//
// Backward: the forward states are kept in a caller-supplied workspace ([element][window],
// coalesced), then an adjoint sweep right to left produces dT_c, from which
//   dCore_c += dT_c (x) f_c   (wave reduction, one float atomic per element per wave)
//   df_c -> d/d(pixel features), written per window and summed per pixel by a gather kernel.
#include "common.h"

#define SBS_MAXC 32

struct SbsP {
  int n, C, B, H, W, q, qc, Ho, Wo, l0, Otot, vmax;
  int cs;   // window columns per workgroup (power of two <= 64); lanes t and t+cs mirror one window
  int cst;  // LDS column stride (cs forward; cs + 1 backward: conflict-free for both lane roles)
  int cmax;                          // largest core (elements): size of the LDS core tile
  long long core_off[SBS_MAXC + 1];  // offsets of the cores in the workgroup's LDS dCore accumulator
  long long Wn;
  long long s[5];
  int o[SBS_MAXC], bl[SBS_MAXC], br[SBS_MAXC], ph[SBS_MAXC], pw[SBS_MAXC];
  int oacc[SBS_MAXC + 1];        // product of out sizes of cores < c
  long long st_off[SBS_MAXC + 1];  // element offsets of the stored forward states
  const void* core[SBS_MAXC];
  void* dcore[SBS_MAXC];  // A-typed accumulators (== dCores for f32/f64)
};

namespace {

struct WinCoord {
  long long b;
  int ho, wo;
};
__device__ __forceinline__ WinCoord win_coord(const SbsP& p, long long w) {
  WinCoord c;
  const int hw = p.Ho * p.Wo;
  c.b = w / hw;
  const int rem = (int)(w - c.b * hw);
  c.ho = rem / p.Wo;
  c.wo = rem - c.ho * p.Wo;
  return c;
}

// f[qq] = prod_ch x[ch][pixel of core c][digit_ch(qq)], channel 0 most significant
template <typename S, typename A>
__device__ __forceinline__ void pixel_features(const S* __restrict__ x, const SbsP& p, int c,
                                               bool valid, const WinCoord& wc, A* f, int col) {
  for (int qq = 0; qq < p.qc; ++qq) {
    A pr = A(1);
    int t = qq;
    for (int ch = p.C - 1; ch >= 0; --ch) {
      const int dg = t % p.q;
      t /= p.q;
      const S* px = x + ch * p.s[0] + wc.b * p.s[1] + (long long)(wc.ho + p.ph[c]) * p.s[2] +
                    (long long)(wc.wo + p.pw[c]) * p.s[3] + dg * p.s[4];
      pr *= valid ? (A)(*px) : A(0);
    }
    f[qq * p.cst + col] = pr;
  }
}

// one sweep step: vb[(a*oc+o)*R + r] = sum_l va[a*L + l] * sum_qq core[o,l,r,qq] f[qq]
template <typename S, typename A>
__device__ __forceinline__ void sweep_step(const SbsP& p, int c, const A* va, A* vb, const A* f,
                                           int col) {
  const int L = p.bl[c], R = p.br[c], oc = p.o[c], Oacc = p.oacc[c];
  const S* core = (const S*)p.core[c];
  for (int e = 0; e < Oacc * oc * R; ++e) vb[e * p.cst + col] = A(0);
  for (int o = 0; o < oc; ++o)
    for (int l = 0; l < L; ++l)
      for (int r = 0; r < R; ++r) {
        const S* cp = core + (long long)((o * L + l) * R + r) * p.qc;
        A t = A(0);
        for (int qq = 0; qq < p.qc; ++qq) t += (A)cp[qq] * f[qq * p.cst + col];
        for (int a = 0; a < Oacc; ++a)
          vb[((a * oc + o) * R + r) * p.cst + col] += va[(a * L + l) * p.cst + col] * t;
      }
}

template <typename S, typename A>
__global__ __launch_bounds__(DCTN_WAVE) void convsbs_fwd_generic_k(const S* __restrict__ x,
                                                                   S* __restrict__ out, SbsP p) {
  extern __shared__ __align__(16) unsigned char smem[];
  A* va = reinterpret_cast<A*>(smem);
  A* vb = va + (size_t)p.vmax * p.cs;
  A* f = vb + (size_t)p.vmax * p.cs;
  A* oa = f + (size_t)p.qc * p.cs;
  const int tid = threadIdx.x;
  const int col = tid & (p.cs - 1);
  const bool primary = tid < p.cs;  // lanes tid >= cs mirror the window of lane tid % cs
  const long long w = (long long)blockIdx.x * p.cst + col;
  const bool valid = w < p.Wn;
  WinCoord wc = {0, 0, 0};
  if (valid) wc = win_coord(p, w);
  for (int a = 0; a < p.Otot; ++a) oa[a * p.cst + col] = A(0);
  for (int s = 0; s < p.l0; ++s) {
    for (int l = 0; l < p.l0; ++l) va[l * p.cst + col] = (l == s) ? A(1) : A(0);
    A* cur = va;
    A* nxt = vb;
    for (int c = 0; c < p.n; ++c) {
      pixel_features<S, A>(x, p, c, valid, wc, f, col);
      sweep_step<S, A>(p, c, cur, nxt, f, col);
      A* tmp = cur; cur = nxt; nxt = tmp;
    }
    for (int a = 0; a < p.Otot; ++a)
      oa[a * p.cst + col] += cur[(a * p.l0 + s) * p.cst + col];
  }
  if (valid && primary)
    for (int a = 0; a < p.Otot; ++a) out[w * p.Otot + a] = (S)oa[a * p.cst + col];
}

// Backward.  Lanes play two roles per core:
//   role "window"  (lane = window column): forward sweep (states to the workspace), adjoint sweep:
//       dT, d(state), d(pixel features); v_c, dv_{c+1} and f_c stay in the lane's LDS column;
//   role "element" (lane = core element (o,l,r,qq)): dCore_c[e] += sum over the block's windows of
//       f[qq] * sum_a v[a,l] * dv[(a,o),r], read from the same LDS image (column stride cs + 1 keeps
//       both access patterns conflict-free).  No shuffles, no atomics inside the loop.
// Workgroups are persistent: dCore is accumulated in LDS over all their window groups and
// flushed with one atomic per element per workgroup at the end.
template <typename S, typename A>
__global__ __launch_bounds__(256) void convsbs_bwd_generic_k(
    const S* __restrict__ x, const S* __restrict__ dY, A* __restrict__ states,
    A* __restrict__ gxw, SbsP p, int need_dx, int need_dcore, long long ngroups) {
  extern __shared__ __align__(16) unsigned char smem[];
  A* va = reinterpret_cast<A*>(smem);
  A* vb = va + (size_t)p.vmax * p.cst;
  A* vc = vb + (size_t)p.vmax * p.cst;
  A* f = vc + (size_t)p.vmax * p.cst;
  A* df = f + (size_t)p.qc * p.cst;
  A* dys = df + (size_t)p.qc * p.cst;
  A* dacc = dys + (size_t)p.Otot * p.cst;   // [core_off[n]] dCore accumulator of this workgroup
  // a workgroup has blockDim.x / 64 waves; with several waves every wave carries 64 columns
  // (cs == 64), with one wave the mirror-lane scheme applies (cs <= 64)
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int csb = p.cst - 1;                       // window columns of the workgroup
  const int col = nthr > DCTN_WAVE ? tid : (tid & (p.cs - 1));
  const bool primary = nthr > DCTN_WAVE ? true : tid < p.cs;
  if (need_dcore)
    for (long long e = tid; e < p.core_off[p.n]; e += nthr) dacc[e] = A(0);

  for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long long w = grp * csb + col;
    const bool valid = w < p.Wn;
    const long long wcol = valid ? w : 0;
    WinCoord wc = {0, 0, 0};
    if (valid) wc = win_coord(p, w);
    __syncthreads();
    for (int a = 0; a < p.Otot; ++a) dys[a * p.cst + col] = valid ? (A)dY[w * p.Otot + a] : A(0);

    for (int s = 0; s < p.l0; ++s) {
      // ---- forward sweep, storing the input state of every core
      for (int l = 0; l < p.l0; ++l) va[l * p.cst + col] = (l == s) ? A(1) : A(0);
      A* cur = va;
      A* nxt = vb;
      for (int c = 0; c < p.n; ++c) {
        const int ne = p.oacc[c] * p.bl[c];
        if (valid)  // mirror lanes store the same values: every lane later reads what it wrote itself
          for (int e = 0; e < ne; ++e)
            states[(p.st_off[c] + e) * p.Wn + wcol] = cur[e * p.cst + col];
        if (c + 1 < p.n) {
          pixel_features<S, A>(x, p, c, valid, wc, f, col);
          sweep_step<S, A>(p, c, cur, nxt, f, col);
          A* tmp = cur; cur = nxt; nxt = tmp;
        }
      }
      // ---- adjoint sweep.  dv: gradient wrt the OUTPUT state of core c, shape [Oacc*oc][R]
      A* dv = vb;
      A* dvn = vc;
      for (int a = 0; a < p.Otot; ++a)
        for (int r = 0; r < p.l0; ++r)
          dv[(a * p.l0 + r) * p.cst + col] = (r == s) ? dys[a * p.cst + col] : A(0);
      for (int c = p.n - 1; c >= 0; --c) {
        const int L = p.bl[c], R = p.br[c], oc = p.o[c], Oacc = p.oacc[c];
        const S* core = (const S*)p.core[c];
        for (int e = 0; e < Oacc * L; ++e)
          va[e * p.cst + col] = valid ? states[(p.st_off[c] + e) * p.Wn + wcol] : A(0);
        pixel_features<S, A>(x, p, c, valid, wc, f, col);
        for (int e = 0; e < Oacc * L; ++e) dvn[e * p.cst + col] = A(0);
        for (int qq = 0; qq < p.qc; ++qq) df[qq * p.cst + col] = A(0);
        for (int o = 0; o < oc; ++o)
          for (int l = 0; l < L; ++l)
            for (int r = 0; r < R; ++r) {
              const int cbase = ((o * L + l) * R + r) * p.qc;
              A t = A(0);
              for (int qq = 0; qq < p.qc; ++qq) t += (A)core[cbase + qq] * f[qq * p.cst + col];
              A dT = A(0);
              for (int a = 0; a < Oacc; ++a) {
                const A g = dv[((a * oc + o) * R + r) * p.cst + col];
                dT += va[(a * L + l) * p.cst + col] * g;
                dvn[(a * L + l) * p.cst + col] += t * g;
              }
              if (need_dx)
                for (int qq = 0; qq < p.qc; ++qq) df[qq * p.cst + col] += dT * (A)core[cbase + qq];
            }
        if (need_dcore) {
          // role "element": this lane owns elements e = tid, tid + 64, ... of core c
          __syncthreads();
          const int E = oc * L * R * p.qc;
          for (int e = tid; e < E; e += nthr) {
            const int qq = e % p.qc;
            int t2 = e / p.qc;
            const int r = t2 % R; t2 /= R;
            const int l = t2 % L;
            const int o = t2 / L;
            A acc = A(0);
            for (int wl = 0; wl < csb; ++wl) {
              A dT = A(0);
              for (int a = 0; a < Oacc; ++a)
                dT += va[(a * L + l) * p.cst + wl] * dv[((a * oc + o) * R + r) * p.cst + wl];
              acc += dT * f[qq * p.cst + wl];
            }
            dacc[p.core_off[c] + e] += acc;
          }
          __syncthreads();
        }
        if (need_dx && valid && primary) {
          // d/d x[ch][pixel_c][qv] = sum_{qq: digit_ch(qq) = qv} df[qq] * prod_{ch' != ch} x[ch'][digit]
          for (int ch = 0; ch < p.C; ++ch)
            for (int qv = 0; qv < p.q; ++qv) {
              A g = A(0);
              for (int qq = 0; qq < p.qc; ++qq) {
                int t = qq;
                A pr = A(1);
                bool hit = false;
                for (int c2 = p.C - 1; c2 >= 0; --c2) {
                  const int dg = t % p.q;
                  t /= p.q;
                  if (c2 == ch) {
                    hit = (dg == qv);
                  } else {
                    pr *= (A)x[c2 * p.s[0] + wc.b * p.s[1] + (long long)(wc.ho + p.ph[c]) * p.s[2] +
                               (long long)(wc.wo + p.pw[c]) * p.s[3] + dg * p.s[4]];
                  }
                }
                if (hit) g += df[qq * p.cst + col] * pr;
              }
              A* dst = &gxw[(long long)((c * p.C + ch) * p.q + qv) * p.Wn + w];
              if (s == 0) *dst = g; else *dst += g;
            }
        }
        A* tmp = dv; dv = dvn; dvn = tmp;
      }
    }
  }
  if (need_dcore) {
    __syncthreads();
    for (int c = 0; c < p.n; ++c) {
      A* dcore = (A*)p.dcore[c];
      const long long E = p.core_off[c + 1] - p.core_off[c];
      for (long long e = tid; e < E; e += nthr) atomicAdd(&dcore[e], dacc[p.core_off[c] + e]);
    }
  }
}

// dX[ch,b,h,w,qv] = sum_c gxw[((c*C+ch)*q+qv)][window (h-ph_c, w-pw_c)]
template <typename S, typename A>
__global__ void convsbs_gather_dx_k(const A* __restrict__ gxw, S* __restrict__ dX, SbsP p) {
  const long long total = (long long)p.C * p.B * p.H * p.W * p.q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    long long t = idx;
    const int qv = (int)(t % p.q); t /= p.q;
    const int wi = (int)(t % p.W); t /= p.W;
    const int hi = (int)(t % p.H); t /= p.H;
    const int b = (int)(t % p.B);
    const int ch = (int)(t / p.B);
    A acc = A(0);
    for (int c = 0; c < p.n; ++c) {
      const int ho = hi - p.ph[c], wo = wi - p.pw[c];
      if (ho < 0 || ho >= p.Ho || wo < 0 || wo >= p.Wo) continue;
      const long long win = ((long long)b * p.Ho + ho) * p.Wo + wo;
      acc += gxw[(long long)((c * p.C + ch) * p.q + qv) * p.Wn + win];
    }
    dX[idx] = (S)acc;
  }
}

template <typename S, typename A>
__global__ void convert_sbs_k(const A* __restrict__ src, S* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    dst[i] = (S)src[i];
}

int fill(SbsP& p, const int64_t xs[5], int n, const int* out_sizes, const int* bond_sizes,
         const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q) {
  if (n < 1 || n > SBS_MAXC) return n < 1 ? DCTN_ERR_BAD_SHAPE : DCTN_ERR_UNSUPPORTED;
  if (C < 1 || B < 1 || q < 1) return DCTN_ERR_BAD_SHAPE;
  int max_h = 0, max_w = 0, min_h = 1 << 30, min_w = 1 << 30;
  for (int c = 0; c < n; ++c) {
    if (out_sizes[c] < 1 || bond_sizes[c] < 1 || pos_h[c] < 0 || pos_w[c] < 0)
      return DCTN_ERR_BAD_SHAPE;
    max_h = pos_h[c] > max_h ? pos_h[c] : max_h;
    max_w = pos_w[c] > max_w ? pos_w[c] : max_w;
    min_h = pos_h[c] < min_h ? pos_h[c] : min_h;
    min_w = pos_w[c] < min_w ? pos_w[c] : min_w;
  }
  if (min_h != 0 || min_w != 0) return DCTN_ERR_BAD_SHAPE;  // dctn/align.py:18-19
  if (H <= max_h || W <= max_w) return DCTN_ERR_BAD_SHAPE;
  p.n = n; p.C = C; p.B = B; p.H = H; p.W = W; p.q = q;
  long long qc = 1;
  for (int c = 0; c < C; ++c) { qc *= q; if (qc > 4096) return DCTN_ERR_UNSUPPORTED; }
  p.qc = (int)qc;
  p.Ho = H - max_h; p.Wo = W - max_w;
  p.Wn = (long long)B * p.Ho * p.Wo;
  for (int i = 0; i < 5; ++i) p.s[i] = xs ? xs[i] : 0;
  p.l0 = bond_sizes[0];
  long long oacc = 1, off = 0;
  int vmax = p.l0;
  for (int c = 0; c < n; ++c) {
    p.o[c] = out_sizes[c];
    p.bl[c] = bond_sizes[c];
    p.br[c] = bond_sizes[(c + 1) % n];
    p.ph[c] = pos_h[c];
    p.pw[c] = pos_w[c];
    p.oacc[c] = (int)oacc;
    p.st_off[c] = off;
    off += oacc * p.bl[c];
    oacc *= out_sizes[c];
    if (oacc * p.br[c] > (1 << 20)) return DCTN_ERR_UNSUPPORTED;
    if ((int)(oacc * p.br[c]) > vmax) vmax = (int)(oacc * p.br[c]);
  }
  p.oacc[n] = (int)oacc;
  p.st_off[n] = off;
  p.cmax = 0;
  for (int c = 0; c < n; ++c) {
    const long long e = (long long)p.o[c] * p.bl[c] * p.br[c] * p.qc;
    if (e > (1 << 20)) return DCTN_ERR_UNSUPPORTED;
    if ((int)e > p.cmax) p.cmax = (int)e;
  }
  p.Otot = (int)oacc;
  p.vmax = vmax;
  return DCTN_OK;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

long long core_elems(const SbsP& p, int c) {
  return (long long)p.o[c] * p.bl[c] * p.br[c] * p.qc;
}

template <typename S, typename A>
int fwd_launch(const void* x, void* out, SbsP& p, hipStream_t st) {
  const size_t per_col = ((size_t)2 * p.vmax + p.qc + p.Otot) * sizeof(A);
  p.cs = DCTN_WAVE;
  while (p.cs > 1 && per_col * p.cs > DCTN_LDS_BUDGET) p.cs >>= 1;
  const size_t lds = per_col * p.cs;
  if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
  p.cst = p.cs;
  const unsigned grid = (unsigned)((p.Wn + p.cs - 1) / p.cs);
  (void)hipFuncSetAttribute((const void*)convsbs_fwd_generic_k<S, A>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((convsbs_fwd_generic_k<S, A>), dim3(grid), dim3(DCTN_WAVE), lds, st,
                     (const S*)x, (S*)out, p);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("convsbs_fwd_generic");
  return DCTN_OK;
}

// State elements per window of the workspace's first region: the generic sweep's own need, or what a slice of the MFMA
// sweep stores (at most two states of the PADDED bond per core) when that is more - a padded bond (6 -> 8) or a ring
// (whose generic states are counted differently) would otherwise let the MFMA states run into the feature-gradient
// region behind them, which ring / many-output slices accumulate across launches.
long long state_elems(const SbsP& p) {
  long long need = p.st_off[p.n];
  int bmax = 1;
  for (int c = 0; c < p.n; ++c) bmax = p.bl[c] > bmax ? p.bl[c] : bmax;
  if (bmax <= 16) {
    const int rpad = bmax <= 4 ? 4 : bmax <= 8 ? 8 : 16;
    const long long mfma = (long long)p.n * 2 * rpad;
    if (mfma > need) need = mfma;
  }
  return need;
}

size_t bwd_ws(const SbsP& p, int dtype) {
  const size_t asz = dtype == DCTN_F64 ? 8 : 4;
  size_t total = align256((size_t)state_elems(p) * p.Wn * asz);      // forward states
  total += align256((size_t)p.n * p.C * p.q * p.Wn * asz);           // per-window d/d(pixel features)
  if (dtype == DCTN_BF16)
    for (int c = 0; c < p.n; ++c) total += align256((size_t)core_elems(p, c) * asz);
  if (dtype == DCTN_F32) {   // per-workgroup partial gradients of the MFMA backward (fixed-order reduce instead of atomics)
    size_t ce = 0;
    for (int c = 0; c < p.n; ++c) ce += (size_t)core_elems(p, c);
    total += align256((size_t)SBS_MAX_PARTIAL_RECORDS * ce * sizeof(float));
  }
  return total;
}

template <typename S, typename A>
int bwd_launch(const void* x, const void* dY, void* dX, void* const* dCores, void* ws,
               size_t ws_bytes, SbsP& p, int dtype, hipStream_t st, const void* saved = nullptr) {
  if (!ws || bwd_ws(p, dtype) > ws_bytes) return DCTN_ERR_WORKSPACE;
  p.core_off[0] = 0;
  for (int c = 0; c < p.n; ++c) p.core_off[c + 1] = p.core_off[c] + core_elems(p, c);
  const int need_dcore_lds = dCores != nullptr;
  const size_t acc_bytes = need_dcore_lds ? (size_t)p.core_off[p.n] * sizeof(A) : 0;
  if (acc_bytes > DCTN_LDS_BUDGET / 2) return DCTN_ERR_UNSUPPORTED;
  const size_t per_col = ((size_t)3 * p.vmax + 2 * p.qc + p.Otot) * sizeof(A);
  p.cs = DCTN_WAVE;
  while (p.cs > 1 && per_col * (p.cs + 1) + acc_bytes > DCTN_LDS_BUDGET) p.cs >>= 1;
  int waves = 1;  // several waves share one dCore accumulator when every wave has its 64 columns
  if (p.cs == DCTN_WAVE && acc_bytes >= 4096)  // small strings: more, smaller workgroups win
    while (waves < 4 && per_col * (2 * waves * DCTN_WAVE + 1) + acc_bytes <= DCTN_LDS_BUDGET) waves *= 2;
  const int csb = waves * p.cs;
  p.cst = csb + 1;
  const size_t lds = per_col * p.cst + acc_bytes;
  if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
  unsigned char* wsp = (unsigned char*)ws;
  A* states = (A*)wsp;
  wsp += align256((size_t)state_elems(p) * p.Wn * sizeof(A));
  A* gxw = (A*)wsp;
  wsp += align256((size_t)p.n * p.C * p.q * p.Wn * sizeof(A));
  const int need_dcore = dCores != nullptr;
  if (need_dcore) {
    for (int c = 0; c < p.n; ++c) {
      if (!dCores[c]) return DCTN_ERR_NULL;
      if constexpr (sizeof(S) == sizeof(A)) {
        p.dcore[c] = dCores[c];
      } else {
        p.dcore[c] = wsp;
        wsp += align256((size_t)core_elems(p, c) * sizeof(A));
      }
    }
    // the accumulators start from zero: ONE fill when they sit back to back (the Python wrapper allocates the
    // gradients of a string as one flat buffer; nine separate fills were 30 us of a 200 us backward), else one each
    bool flat = true;
    size_t total = 0;
    for (int c = 0; c < p.n; ++c) {
      if (c > 0 && (unsigned char*)p.dcore[c] != (unsigned char*)p.dcore[0] + total) flat = false;
      total += (size_t)core_elems(p, c) * sizeof(A);
    }
    if (flat) {
      if (dctn_zero_async(p.dcore[0], total, st) != DCTN_OK) return DCTN_ERR_LAUNCH;
    } else {
      for (int c = 0; c < p.n; ++c)
        if (dctn_zero_async(p.dcore[c], (size_t)core_elems(p, c) * sizeof(A), st) != DCTN_OK)
          return DCTN_ERR_LAUNCH;
    }
  }
  if constexpr (sizeof(S) == 4 && sizeof(A) == 4) {
    // float32 strings of the MFMA family: register-resident sweep on the matrix cores
    if (need_dcore) {
      int outs[SBS_MAXC], bonds[SBS_MAXC];
      float* dcp[SBS_MAXC];
      const void* cp[SBS_MAXC];
      for (int c = 0; c < p.n; ++c) { outs[c] = p.o[c]; bonds[c] = p.bl[c]; dcp[c] = (float*)p.dcore[c]; cp[c] = p.core[c]; }
      size_t ce = 0;
      for (int c = 0; c < p.n; ++c) ce += (size_t)core_elems(p, c);
      const size_t pbytes = (size_t)SBS_MAX_PARTIAL_RECORDS * ce * sizeof(float);
      float* partials = ((size_t)((unsigned char*)ws + ws_bytes - wsp) >= pbytes) ? (float*)wsp : nullptr;
      const int rcm = convsbs_bwd_mfma(x, (const int64_t*)p.s, cp, dY, (float*)states, dX ? (float*)gxw : nullptr, dcp, p.n, outs,
                                       bonds, p.ph, p.pw, p.C, p.B, p.H, p.W, p.q, dtype, st, partials, pbytes, (const float*)saved);
      if (rcm == DCTN_OK) {
        if (dX) {
          const long long total = (long long)p.C * p.B * p.H * p.W * p.q;
          const unsigned g2 = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
          hipLaunchKernelGGL((convsbs_gather_dx_k<S, A>), dim3(g2), dim3(256), 0, st, gxw, (S*)dX, p);
          DCTN_CHECK_LAUNCH();
        }
        return DCTN_OK;
      }
      if (rcm != DCTN_ERR_UNSUPPORTED) return rcm;
    }
  }
  const long long ngroups = (p.Wn + csb - 1) / csb;
  // persistent workgroups: as many as can be resident (LDS-limited), at most one per window group
  long long per_cu = (long long)(160 * 1024) / (long long)(lds ? lds : 1);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  long long grid = 256 * per_cu;
  if (grid > ngroups) grid = ngroups;
  (void)hipFuncSetAttribute((const void*)convsbs_bwd_generic_k<S, A>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((convsbs_bwd_generic_k<S, A>), dim3((unsigned)grid), dim3(DCTN_WAVE * waves), lds, st,
                     (const S*)x, (const S*)dY, states, gxw, p, dX != nullptr, need_dcore, ngroups);
  DCTN_CHECK_LAUNCH();
  if (dX) {
    const long long total = (long long)p.C * p.B * p.H * p.W * p.q;
    const unsigned g2 = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((convsbs_gather_dx_k<S, A>), dim3(g2), dim3(256), 0, st, gxw, (S*)dX, p);
    DCTN_CHECK_LAUNCH();
  }
  if constexpr (sizeof(S) != sizeof(A)) {
    if (need_dcore)
      for (int c = 0; c < p.n; ++c) {
        const long long ne = core_elems(p, c);
        hipLaunchKernelGGL((convert_sbs_k<S, A>), dim3((unsigned)((ne + 255) / 256)), dim3(256), 0,
                           st, (const A*)p.dcore[c], (S*)dCores[c], ne);
        DCTN_CHECK_LAUNCH();
      }
  }
  dctn_set_last_kernel("convsbs_bwd_generic");
  return DCTN_OK;
}

}  // namespace

static int sbs_largest_bond(int n, const int* bond_sizes) {
  int m = 0;
  for (int c = 0; c < n; ++c) m = bond_sizes[c] > m ? bond_sizes[c] : m;
  return m;
}

extern "C" {

size_t dctn_convsbs_workspace_bytes(int n_cores, const int* out_sizes, const int* bond_sizes,
                                    int C, int B, int H, int W, int q, const int* pos_h,
                                    const int* pos_w, int dtype_flags, int backward) {
  const int dtype = dtype_flags & DCTN_DTYPE_MASK;   // (the query covers every family: flags only select among them)
  SbsP p;
  if (!out_sizes || !bond_sizes || !pos_h || !pos_w) return 0;
  if (fill(p, nullptr, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q) != DCTN_OK)
    return 0;
  if (!backward) return 256;
  const size_t a = bwd_ws(p, dtype), b = convsbs_reg_bwd_workspace(n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  const size_t c = convsbs_band_bwd_workspace(n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  const size_t m = a > b ? a : b;
  return (m > c ? m : c) + 256;
}

size_t dctn_convsbs_saved_states_bytes(int n_cores, const int* out_sizes, const int* bond_sizes, int C, int B, int H,
                                       int W, int q, const int* pos_h, const int* pos_w, int dtype_flags) {
  const int dtype = dtype_flags & DCTN_DTYPE_MASK;
  if (!out_sizes || !bond_sizes || !pos_h || !pos_w) return 0;
  return convsbs_saved_states_bytes(n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
}

int dctn_convsbs_fwd(const void* x, const int64_t x_strides[5], const void* const* cores,
                     void* out, int n_cores, const int* out_sizes, const int* bond_sizes,
                     const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q,
                     void* workspace, size_t workspace_bytes, int dtype_flags, void* stream) {
  if (!x || !x_strides || !cores || !out || !out_sizes || !bond_sizes || !pos_h || !pos_w)
    return DCTN_ERR_NULL;
  const int dtype = dtype_flags & DCTN_DTYPE_MASK;
  if (dtype_flags & ~(DCTN_DTYPE_MASK | DCTN_SBS_MATRIX_CORE_SWEEP)) return DCTN_ERR_UNSUPPORTED;
  SbsP p;
  int rc = fill(p, x_strides, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q);
  if (rc != DCTN_OK) return rc;
  for (int c = 0; c < n_cores; ++c) {
    if (!cores[c]) return DCTN_ERR_NULL;
    p.core[c] = cores[c];
    p.dcore[c] = nullptr;
  }
  hipStream_t st = (hipStream_t)stream;
  // a workspace of dctn_convsbs_saved_states_bytes(...) bytes: the forward leaves its states there for
  // dctn_convsbs_bwd_saved (a training forward); anything smaller: plain forward
  // small bonds: the register-resident sweep (keeps nothing: its backward recomputes the chain in registers)
  if (!(dtype_flags & DCTN_SBS_MATRIX_CORE_SWEEP)) {
    rc = convsbs_fwd_reg(x, x_strides, cores, out, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype, st);
    if (rc != DCTN_ERR_UNSUPPORTED) return rc;
  }
  // bonds 5..16: the band family's forward (nothing kept: its backward recomputes the chain in registers); under
  // DCTN_SBS_MATRIX_CORE_SWEEP bonds up to 8 stay on the matrix-core sweep below (the tests hold both against the oracle)
  if (!((dtype_flags & DCTN_SBS_MATRIX_CORE_SWEEP) && sbs_largest_bond(n_cores, bond_sizes) <= 8)) {
    rc = convsbs_fwd_band(x, x_strides, cores, out, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype, st);
    if (rc != DCTN_ERR_UNSUPPORTED) return rc;
  }
  const size_t sb = convsbs_saved_states_bytes(n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  float* save = (sb > 0 && workspace && workspace_bytes >= sb) ? (float*)workspace : nullptr;
  rc = convsbs_fwd_mfma(x, x_strides, cores, out, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q,
                        dtype, st, save);
  // the caller learns whether the states were WRITTEN: only the matrix-core sweep writes them, and it can still decline
  // a string the size query accepted (its LDS plan); the generic sweep below leaves the buffer untouched
  if (rc == DCTN_OK) return save ? DCTN_SAVED : DCTN_OK;
  if (rc != DCTN_ERR_UNSUPPORTED) return rc;
  switch (dtype) {
    case DCTN_F32: return fwd_launch<float, float>(x, out, p, st);
    case DCTN_F64: return fwd_launch<double, double>(x, out, p, st);
    case DCTN_BF16: return fwd_launch<bf16_t, float>(x, out, p, st);
  }
  return DCTN_ERR_BAD_DTYPE;
}

int dctn_convsbs_bwd(const void* x, const int64_t x_strides[5], const void* const* cores,
                     const void* dY, void* dX, void* const* dCores, int n_cores,
                     const int* out_sizes, const int* bond_sizes, const int* pos_h,
                     const int* pos_w, int C, int B, int H, int W, int q, void* workspace,
                     size_t workspace_bytes, int dtype_flags, void* stream) {
  const int dtype = dtype_flags;   // (decoded by dctn_convsbs_bwd_saved)
  return dctn_convsbs_bwd_saved(x, x_strides, cores, dY, dX, dCores, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H,
                                W, q, workspace, workspace_bytes, nullptr, 0, dtype, stream);
}

int dctn_convsbs_bwd_saved(const void* x, const int64_t x_strides[5], const void* const* cores,
                           const void* dY, void* dX, void* const* dCores, int n_cores,
                           const int* out_sizes, const int* bond_sizes, const int* pos_h,
                           const int* pos_w, int C, int B, int H, int W, int q, void* workspace,
                           size_t workspace_bytes, const void* saved_states, size_t saved_states_bytes, int dtype_flags,
                           void* stream) {
  if (!x || !x_strides || !cores || !dY || !out_sizes || !bond_sizes || !pos_h || !pos_w)
    return DCTN_ERR_NULL;
  const int dtype = dtype_flags & DCTN_DTYPE_MASK;
  if (dtype_flags & ~(DCTN_DTYPE_MASK | DCTN_SBS_MATRIX_CORE_SWEEP)) return DCTN_ERR_UNSUPPORTED;
  if (saved_states) {   // only what the forward of this very shape can have written
    const size_t sb = convsbs_saved_states_bytes(n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
    if (sb == 0 || saved_states_bytes < sb) saved_states = nullptr;
  }
  if (!dX && !dCores) return DCTN_OK;
  SbsP p;
  int rc = fill(p, x_strides, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q);
  if (rc != DCTN_OK) return rc;
  for (int c = 0; c < n_cores; ++c) {
    if (!cores[c]) return DCTN_ERR_NULL;
    p.core[c] = cores[c];
    p.dcore[c] = nullptr;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCTN_F32 && !(dtype_flags & DCTN_SBS_MATRIX_CORE_SWEEP)) {
    for (int c = 0; dCores && c < n_cores; ++c)
      if (!dCores[c]) return DCTN_ERR_NULL;
    rc = convsbs_bwd_reg(x, x_strides, cores, dY, dX, (float* const*)dCores, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H,
                         W, q, dtype, st, workspace, workspace_bytes);
    if (rc != DCTN_ERR_UNSUPPORTED) return rc;
  }
  if (dtype == DCTN_F32 && !((dtype_flags & DCTN_SBS_MATRIX_CORE_SWEEP) && sbs_largest_bond(n_cores, bond_sizes) <= 8)) {
    // bonds 5..16: the band-owning backward (recomputes the chain; no saved states, no helper launches)
    for (int c = 0; dCores && c < n_cores; ++c)
      if (!dCores[c]) return DCTN_ERR_NULL;
    rc = convsbs_bwd_band(x, x_strides, cores, dY, dX, (float* const*)dCores, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H,
                          W, q, dtype, st, workspace, workspace_bytes);
    if (rc != DCTN_ERR_UNSUPPORTED) return rc;
  }
  switch (dtype) {
    case DCTN_F32:
      return bwd_launch<float, float>(x, dY, dX, dCores, workspace, workspace_bytes, p, dtype, st, saved_states);
    case DCTN_F64:
      return bwd_launch<double, double>(x, dY, dX, dCores, workspace, workspace_bytes, p, dtype, st);
    case DCTN_BF16:
      return bwd_launch<bf16_t, float>(x, dY, dX, dCores, workspace, workspace_bytes, p, dtype, st);
  }
  return DCTN_ERR_BAD_DTYPE;
}

// ---- several strings of one layer (ManyConvSBS, dctn/conv_sbs.py:314-370) in one launch each way
size_t dctn_convsbs_many_workspace_bytes(int n_strings, int n_cores, const int* out_sizes, const int* bond_sizes, int C, int B,
                                         int H, int W, int q, const int* pos_h, const int* pos_w, int dtype) {
  if (!out_sizes || !bond_sizes || !pos_h || !pos_w) return 0;
  const size_t a = convsbs_many_reg_bwd_workspace(n_strings, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q,
                                                  dtype & DCTN_DTYPE_MASK);
  if (a > 0) return a;
  return convsbs_many_band_bwd_workspace(n_strings, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q,
                                         dtype & DCTN_DTYPE_MASK);
}

int dctn_convsbs_many_fwd(const void* x, const int64_t x_strides[5], const void* const* cores, void* const* outs,
                          int n_strings, int n_cores, const int* out_sizes, const int* bond_sizes, const int* pos_h,
                          const int* pos_w, int C, int B, int H, int W, int q, int dtype, void* stream) {
  if (!x || !x_strides || !cores || !outs || !out_sizes || !bond_sizes || !pos_h || !pos_w) return DCTN_ERR_NULL;
  if (n_strings < 1 || n_cores < 1) return DCTN_ERR_BAD_SHAPE;
  if (dtype & ~DCTN_DTYPE_MASK) return DCTN_ERR_UNSUPPORTED;
  for (int i = 0; i < n_strings * n_cores; ++i)
    if (!cores[i]) return DCTN_ERR_NULL;
  const int rc = convsbs_many_fwd_reg(x, x_strides, cores, outs, n_strings, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H,
                                      W, q, dtype, (hipStream_t)stream);
  if (rc != DCTN_ERR_UNSUPPORTED) return rc;
  return convsbs_many_fwd_band(x, x_strides, cores, outs, n_strings, n_cores, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q,
                               dtype, (hipStream_t)stream);
}

int dctn_convsbs_many_bwd(const void* x, const int64_t x_strides[5], const void* const* cores, const void* const* dYs, void* dX,
                          void* const* dCores, int n_strings, int n_cores, const int* out_sizes, const int* bond_sizes,
                          const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q, void* workspace,
                          size_t workspace_bytes, int dtype, void* stream) {
  if (!x || !x_strides || !cores || !dYs || !out_sizes || !bond_sizes || !pos_h || !pos_w) return DCTN_ERR_NULL;
  if (n_strings < 1 || n_cores < 1) return DCTN_ERR_BAD_SHAPE;
  if (dtype & ~DCTN_DTYPE_MASK) return DCTN_ERR_UNSUPPORTED;
  if (!dX && !dCores) return DCTN_OK;
  for (int i = 0; i < n_strings * n_cores; ++i)
    if (!cores[i] || (dCores && !dCores[i])) return DCTN_ERR_NULL;
  const int rc = convsbs_many_bwd_reg(x, x_strides, cores, dYs, dX, (float* const*)dCores, n_strings, n_cores, out_sizes, bond_sizes,
                                      pos_h, pos_w, C, B, H, W, q, dtype, (hipStream_t)stream, workspace, workspace_bytes);
  if (rc != DCTN_ERR_UNSUPPORTED) return rc;
  return convsbs_many_bwd_band(x, x_strides, cores, dYs, dX, (float* const*)dCores, n_strings, n_cores, out_sizes, bond_sizes,
                               pos_h, pos_w, C, B, H, W, q, dtype, (hipStream_t)stream, workspace, workspace_bytes);
}

}  // extern "C"
