// MFMA EPS kernels, family "bigcore": power-of-two Q, cores that do not fit in registers
// (BASELINE cfg3: K=4/Q=2 cores of 1-2 MiB, K=3/Q=4 of 6 MiB, K=2/Q=8).  Exact float32:
// v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain, so this path keeps the reference's
// float32 numerics (no bf16 rounding) while running on the matrix cores.
//
// One template covers the three GEMMs of the path (reference: dctn/eps.py:25-30 and its autograd):
//     FWD  T[(b,o), w] = sum_a     core[a,b,o] * P0[w,a]            rows (b,o), k = a
//     G0   G0[a, w]    = sum_(b,o) core[a,b,o] * P1[w,b] dY[w,o]    rows a,     k = (b,o)
//     G1   G1[b, w]    = sum_(a,o) core[a,b,o] * P0[w,a] dY[w,o]    rows b,     k = (a,o)
// In every mode the matrix operand is a 32-row tile of the core streamed through LDS (double
// buffered, gathered from the core's natural layout - no packed copy in HBM), and the other
// operand is GENERATED: a lane owns one window (column of the MFMA tile) and produces
// P[w][k] = hi(k) * table[k_lo] from a small per-lane register table and the window's features in
// LDS.  Epilogues are lane-local: FWD weights the 16 accumulator rows with P1[w,b] and reduces
// over b; G0/G1 turn dL/dP0 (dL/dP1) into per-factor gradients by leave-one-out products.
// A wave carries NT column tiles (NT*32 windows) per streamed core tile.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) int int2v;

// The file is compiled three times (Makefile: -DBC_PART=0/1/2) so that its 60-odd kernel
// instantiations build in parallel: part 1 owns the forward launchers for LOGO_T >= 3, part 2 the
// transposed-GEMM launchers, part 0 everything else (dCore kernel, planning, the entry points).
#ifndef BC_PART
#define BC_PART 0
#endif

namespace dctn_bc {
struct BigP {
  int C, B, H, W, K, O, Q, LQ, N, n0, n1, Ho, Wo;
  int OP, LOGO;       // O padded to a power of two
  long long Wn;
  long long s[5];
  int mode;
  int rows;           // GEMM rows (multiple of 32 after padding is handled by guards)
  int kdim;           // GEMM k extent (even)
  int mk, ID;         // inner block: ID = 2*tbl k-values = Q^mk * (mode==FWD ? 1 : OP); tbl <= BC_TBL
  int tbl;
  int nhb;            // hi blocks: kdim / ID
  int khalf_first, khalf_n;   // factors of the half generating the k operand: first factor, count
  int rhalf_first, rhalf_n;   // factors of the half indexing the rows
  int rg_count, mt_per_rg;    // row tiles are split over grid.y (deterministic partial slices)
  int BnO;                    // Bn * O: stride of `a` in the core
  // xo = 1 (O not a power of two): EXACT out size, no padded rows / k-values.
  //   FWD rows R = b*O + o in the core's memory order (b = R / O by multiply-shift);
  //   G0 / G1 put o OUTERMOST in k: k = o*Kh + kh with kh = b (G0) or a (G1), Kh = 2^lkh, so the
  //   generated operand keeps a power-of-two table over the low digits of kh and the hi product
  //   of block hb is dY[w, o = hb >> lnhbo] * KR(high digits, hb & (2^lnhbo - 1)).
  int xo, lkh, lnhbo;
  unsigned odiv_m;            // ceil(2^32 / O)
  // FWD, training: the GEMM result Z = T[(b,o), w] is kept for the backward (the reference's autograd saves it too,
  // dctn/eps.py:25-30 step (0,1)), in row-quad-major order Z[R / 4][w][R % 4] - a lane's four accumulator registers
  // of one row quad are one 16-byte store, 512 contiguous bytes per lane half.  NULL: nothing is kept.
  float* zsave;
};
int launch_fwd_hi(const void* x, const void* core, void* out, const BigP& b, size_t lds, hipStream_t st);
int launch_g(int mode, const void* x, const void* core, const void* dY, void* out, const BigP& b, size_t lds,
             hipStream_t st);
}  // namespace dctn_bc

namespace {

constexpr int BC_WAVES = 4;     // waves per workgroup
constexpr int BC_NT_FWD = 2;    // column tiles (of 32 windows) per wave, forward
constexpr int BC_NT_G = 1;      // ... transposed GEMMs (their LDS also holds the factor gradients)
constexpr int BC_SROW = 33;     // padded row length of a staged k-row (32 rows + 1: conflict-free both ways)
constexpr int BC_TBL_MAX = 16;  // generated-operand table entries per lane (k-steps per hi block): 4, 8 or 16
constexpr int BC_KSTG = 64;     // k-steps (of 2) per LDS stage

enum { MODE_FWD = 0, MODE_G0 = 1, MODE_G1 = 2 };

#if defined(DCTN_STAMPS) && BC_PART != 0
// diagnostic build only (make EXTRA=-DDCTN_STAMPS, tools/stamp_bigcore.py): where a workgroup's wave 0 spends its cycles
// (s_memtime): slot 0 total, 1 prologue, 2 waiting at the stage barrier, 3 core-tile fetch issue, 4 generated-operand +
// MFMA blocks, 5 core-tile commit, 6 epilogue of the row tiles, 7 first-stage fetch + commit of the row tiles
__device__ unsigned long long bc_stamps[16384 * 8];
#define BC_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define BC_ACC(slot, a, b) do { if (threadIdx.x == 0) bc_acc[slot] += (b) - (a); } while (0)
#else
#define BC_T(var) do { } while (0)
#define BC_ACC(slot, a, b) do { } while (0)
#endif

using dctn_bc::BigP;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float half_sum(float v) {
  const int iv = __float_as_int(v);
  const int2v r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}

// source offset of core[a][b][o]
__device__ __forceinline__ long long core_off(const BigP& p, int a, int b, int o) {
  return ((long long)a * (1 << (p.n1 * p.LQ)) + b) * p.O + o;
}

// element (row R, k) of the mode's matrix operand, 0 outside the real extents
__device__ __forceinline__ float amat(const float* __restrict__ core, const BigP& p, int R, int k) {
  int a, b, o;
  if (p.mode == MODE_FWD) {
    a = k; b = R >> p.LOGO; o = R & (p.OP - 1);
  } else if (p.mode == MODE_G0) {
    a = R; b = k >> p.LOGO; o = k & (p.OP - 1);
  } else {
    b = R; a = k >> p.LOGO; o = k & (p.OP - 1);
  }
  if (R >= p.rows || o >= p.O) return 0.f;
  return core[core_off(p, a, b, o)];
}

// Khatri-Rao product over `nf` factors starting at factor `first`, digits of `idx` most significant
// first, features from the block's LDS image xs[(n*Q+q)][BC_WPB]
template <int WPB>
__device__ __forceinline__ float kr(const float* xs, const BigP& p, int first, int nf, int idx, int wl) {
  float v = 1.f;
  for (int d = 0; d < nf; ++d) {
    const int dg = (idx >> ((nf - 1 - d) * p.LQ)) & (p.Q - 1);
    v *= xs[((first + d) * p.Q + dg) * WPB + wl];
  }
  return v;
}

// Same product with a compile-time bound on the number of factors: every LDS read is issued
// before the first multiply (digits past `nf` read a row of ones), so the latency of all of them
// overlaps - and overlaps with MFMAs in flight.
constexpr int BC_MAXD = 8;
template <int WPB>
__device__ __forceinline__ float kr_flat(const float* xs, const BigP& p, int first, int nf, int idx,
                                         int wl, int one_row) {
  float f[BC_MAXD];
#pragma unroll
  for (int d = 0; d < BC_MAXD; ++d) {
    const int sh = d < nf ? (nf - 1 - d) * p.LQ : 0;
    const int dg = (idx >> sh) & (p.Q - 1);
    const int rowi = d < nf ? (first + d) * p.Q + dg : one_row;
    f[d] = xs[rowi * WPB + wl];
  }
  return ((f[0] * f[1]) * (f[2] * f[3])) * ((f[4] * f[5]) * (f[6] * f[7]));
}

// kr_flat for an index whose lowest bit is the lane half: `idx` is the UNIFORM part (even; the digits and row numbers
// are then scalar arithmetic) and the last factor is read one feature row further on in the upper half (`wl_last` =
// wl + h * WPB): 1 VALU instruction per digit where the per-lane index takes 5.
template <int WPB>
__device__ __forceinline__ float kr_flat_half(const float* xs, const BigP& p, int first, int nf, int idx, int wl,
                                              int wl_last, int one_row) {
  float f[BC_MAXD];
#pragma unroll
  for (int d = 0; d < BC_MAXD; ++d) {
    const int sh = d < nf ? (nf - 1 - d) * p.LQ : 0;
    const int dg = (idx >> sh) & (p.Q - 1);
    const int rowi = d < nf ? (first + d) * p.Q + dg : one_row;
    f[d] = xs[rowi * WPB + (d == nf - 1 ? wl_last : wl)];
  }
  return ((f[0] * f[1]) * (f[2] * f[3])) * ((f[4] * f[5]) * (f[6] * f[7]));
}

// The same with the number of factors at compile time (ND = 1..4, chosen by the launcher: no control
// flow in the main loop): no reads of the ones row and no index arithmetic for absent digits, which
// were 2/3 of the VALU instructions of the main loop.
template <int WPB, int ND>
__device__ __forceinline__ float kr_exact(const float* xs, const BigP& p, int first, int idx, int wl) {
  float f[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const int dg = (idx >> ((ND - 1 - d) * p.LQ)) & (p.Q - 1);
    f[d] = xs[((first + d) * p.Q + dg) * WPB + wl];
  }
  float v = f[0];
#pragma unroll
  for (int d = 1; d < ND; ++d) v *= f[d];
  return v;
}

// the pair (wl, wl + 32): the two column tiles of a forward wave in one ds_read2_b32 per digit and packed multiplies
template <int WPB, int ND>
__device__ __forceinline__ f32x2 kr_exact2(const float* xs, const BigP& p, int first, int idx, int wl) {
  f32x2 f[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const int dg = (idx >> ((ND - 1 - d) * p.LQ)) & (p.Q - 1);
    const float* r = xs + ((first + d) * p.Q + dg) * WPB + wl;
    f[d] = f32x2{r[0], r[32]};
  }
  f32x2 v = f[0];
#pragma unroll
  for (int d = 1; d < ND; ++d) v *= f[d];
  return v;
}

// a * {b.x, b.x} and a * {b.y, b.y} as one packed multiply each (the compiler forms them only now and then)
__device__ __forceinline__ f32x2 pk_mul_lo(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_mul_hi(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// LOGO_T: log2 of the padded out size (compile time for FWD, whose epilogue selects output slots
// statically); ignored (0) by the G modes.
// ND: number of hi digits of the generated operand (khalf_n - mk) when in 1..4, else -1 (generic).
template <int MODE, int BC_NT, int LOGO_T, int BC_TBL, int ND>
// Two workgroups share a CU (one wave of each per SIMD): told to the compiler, which otherwise plans for one wave per
// SIMD and spreads over 280-300 registers (accumulators in AGPRs, 16 more as spill space); within 256 all variants but
// the widest (out sizes 16 / 32 with 16-entry tables or generic digit counts) fit without spills - those keep the default.
__global__ __launch_bounds__(64 * BC_WAVES)
__attribute__((amdgpu_waves_per_eu((LOGO_T >= 4 && (BC_TBL == 16 || ND < 0)) ? 1 : 2))) void eps_bigcore_k(const float* __restrict__ x,
                                                                const float* __restrict__ core,
                                                                const float* __restrict__ dY,
                                                                float* __restrict__ out, BigP p) {
  constexpr int BC_WPB = BC_WAVES * BC_NT * 32;  // windows per workgroup
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NQ = p.N * p.Q;
  float* xs = smem;                                   // [NQ + 1][BC_WPB]; row NQ holds ones
  float* dys = xs + (size_t)(NQ + 1) * BC_WPB;        // [OP][BC_WPB]  (G modes)
  float* stage = dys + (MODE == MODE_FWD ? 0 : (size_t)p.OP * BC_WPB);  // [2][BC_KSTG][64]
  float* gxs = stage + 2 * BC_KSTG * 2 * BC_SROW;     // G modes: [rhalf_n*Q][64*BC_WAVES*BC_NT]
  const int tid = threadIdx.x, lane = tid & 63, wl32 = lane & 31, h = lane >> 5, wv = tid >> 6;
  const long long w_block = (long long)blockIdx.x * BC_WPB;
#if defined(DCTN_STAMPS) && BC_PART != 0
  unsigned long long bc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  BC_T(t_start);

  // ---- window features (and dY rows) of the block's windows -> LDS
  {
    // a thread stages factors n = tid / WPB, + threads / WPB, ... of ONE window (the thread count is a multiple of the
    // block's windows): its (image, row, column) is found once, and four factors x four features = up to 16 loads are in
    // flight (one after the other, a first layer's 32 loads and its per-lane table digits made a prologue of 86 k cycles =
    // 22 % of a workgroup's life)
    static_assert((64 * BC_WAVES) % BC_WPB == 0, "one window per thread");
    constexpr int NSTEP = 64 * BC_WAVES / BC_WPB;   // factor stride of a thread
    const int wl = tid % BC_WPB;
    const long long w = w_block + wl;
    const bool valid = w < p.Wn;
    const long long ww = valid ? w : 0;
    const int hw = p.Ho * p.Wo;
    const long long bb = ww / hw;
    const int rem = (int)(ww - bb * hw);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const float* pw = x + bb * p.s[1] + (long long)ho * p.s[2] + (long long)wo * p.s[3];
    // factor n = (window row dh, column dw, channel ch): scalar counters stepped along with n (a wave's threads stage the
    // same factors), no division per factor
    const int n_first = __builtin_amdgcn_readfirstlane(tid / BC_WPB);
    int f_ch = n_first % p.C, f_dw = (n_first / p.C) % p.K, f_dh = n_first / p.C / p.K;
    auto factor_ptr_next = [&]() {   // the current factor's features, then on by NSTEP factors
      const float* r = pw + f_ch * p.s[0] + (long long)f_dh * p.s[2] + (long long)f_dw * p.s[3];
      f_ch += NSTEP;
      while (f_ch >= p.C) {
        f_ch -= p.C;
        if (++f_dw == p.K) { f_dw = 0; ++f_dh; }
      }
      return r;
    };
    constexpr int FB = 4;   // factors per batch: FB x 4 features = up to 16 loads in flight
    for (int n = n_first; n < p.N; n += FB * NSTEP) {
      const float* px[FB];
#pragma unroll
      for (int f = 0; f < FB; ++f) px[f] = factor_ptr_next();   // (past the last factor: computed, not read)
      for (int q0 = 0; q0 < p.Q; q0 += 4) {
        float a[FB][4];
#pragma unroll
        for (int f = 0; f < FB; ++f)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool ok = valid && q0 + u < p.Q && n + f * NSTEP < p.N;
            a[f][u] = ok ? px[f][(long long)(q0 + u) * p.s[4]] : 0.f;
          }
#pragma unroll
        for (int f = 0; f < FB; ++f)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (q0 + u < p.Q && n + f * NSTEP < p.N) xs[((n + f * NSTEP) * p.Q + q0 + u) * BC_WPB + wl] = a[f][u];
      }
    }
  }
  for (int e = tid; e < BC_WPB; e += 64 * BC_WAVES) xs[NQ * BC_WPB + e] = 1.f;
  if (MODE != MODE_FWD) {
    for (int e = tid; e < BC_WPB * p.OP; e += 64 * BC_WAVES) {
      const int wl = e % BC_WPB, o = e / BC_WPB;
      const long long w = w_block + wl;
      dys[o * BC_WPB + wl] = (w < p.Wn && o < p.O) ? dY[w * p.O + o] : 0.f;
    }
    for (int e = tid; e < p.rhalf_n * p.Q * 64 * BC_WAVES * BC_NT; e += 64 * BC_WAVES) gxs[e] = 0.f;
  }
  __syncthreads();

  // ---- per-lane table of the low part of the generated operand: entry t <-> inner k = 2t + h
  f32x2 tab[BC_NT][BC_TBL / 2];   // entry t = tab[nt][t / 2][t & 1]: pairs of k-steps, one packed multiply each
#pragma unroll
  for (int nt = 0; nt < BC_NT; ++nt) {
    const int wl = (wv * BC_NT + nt) * 32 + wl32;
#pragma unroll
    for (int t = 0; t < BC_TBL; ++t) {
      const int kin = 2 * t + h;
      // kin = 2 t + h: everything but the lane half is compile-time / scalar - the digits of the even part are SALU work
      // and the lane half moves the last factor's read one feature row on; all reads of a product ahead of its multiplies
      // (per-lane digits and a chain of LDS round trips per product made a first layer's prologue 22 % of a workgroup's life)
      float tv;
      if (MODE == MODE_FWD || p.xo) {
        tv = kr_flat_half<BC_WPB>(xs, p, p.khalf_first + p.khalf_n - p.mk, p.mk, 2 * t, wl, wl + h * BC_WPB, NQ);
      } else {   // the lane half is part of o (LOGO >= 1)
        tv = kr_flat_half<BC_WPB>(xs, p, p.khalf_first + p.khalf_n - p.mk, p.mk, (2 * t) >> p.LOGO, wl, wl, NQ) *
             dys[(kin & (p.OP - 1)) * BC_WPB + wl];
      }
      tab[nt][t / 2][t & 1] = tv;
    }
  }

  float oacc[BC_NT][16];  // FWD: output slots per lane (OP <= 4: OP slots; else OP/2 <= 16)
#pragma unroll
  for (int nt = 0; nt < BC_NT; ++nt)
#pragma unroll
    for (int s = 0; s < 16; ++s) oacc[nt][s] = 0.f;

  const int mtiles = (p.rows + 31) / 32;
  const int ksteps = p.kdim / 2;                      // MFMA k-steps in total
  const int nstage = (ksteps + BC_KSTG - 1) / BC_KSTG;
  constexpr int hb_per_stage = BC_KSTG / BC_TBL;      // hi blocks per stage
  constexpr int PER = BC_KSTG * 64 / (64 * BC_WAVES); // staged elements per thread
  float pre[PER];

  // element e of a stage: FWD/G1 walk rows fastest (the core is contiguous along the rows there),
  // G0 walks k fastest (rows = a are Bn*O apart, k = (b,o) is contiguous).  The source offset is
  // a per-thread constant plus a uniform term per (row tile, stage): 32 and 128 are multiples of OP.
  unsigned coff[PER];
  unsigned okmask = 0;  // bit i: element i has o < O
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid + 64 * BC_WAVES * i;
    const int row = MODE == MODE_G0 ? e >> 7 : e & 31;
    const int kl = MODE == MODE_G0 ? e & 127 : e >> 5;
    int o;
    if (p.xo) {
      o = 0;  // every staged element is real: validity is row < rows, k < kdim
      if (MODE == MODE_FWD) coff[i] = (unsigned)kl * p.BnO + row;
      else if (MODE == MODE_G0) coff[i] = (unsigned)row * p.BnO + (kl & ((1 << p.lkh) - 1)) * p.O + (kl >> p.lkh);
      else coff[i] = (unsigned)(kl & ((1 << p.lkh) - 1)) * p.BnO + row * p.O + (kl >> p.lkh);
    } else if (MODE == MODE_FWD) {
      o = row & (p.OP - 1);
      coff[i] = (unsigned)kl * p.BnO + (row >> p.LOGO) * p.O + o;
    } else if (MODE == MODE_G0) {
      o = kl & (p.OP - 1);
      coff[i] = (unsigned)row * p.BnO + (kl >> p.LOGO) * p.O + o;
    } else {
      o = kl & (p.OP - 1);
      coff[i] = (unsigned)(kl >> p.LOGO) * p.BnO + row * p.O + o;
    }
    if (o < p.O) okmask |= 1u << i;
  }
  // The 16 elements of a thread are 16 raw buffer loads at a per-thread constant offset plus a uniform one; an element
  // outside the core's extents (padded output o >= O, row >= rows, k >= kdim) carries an out-of-range offset and reads 0:
  // no exec-masked branch (the predicated global loads cost ~400 instructions per stage and wave - 28 branches - against
  // 128 MFMAs of work, in both workgroups of a CU at the same time).
  const unsigned core_bytes = (unsigned)(((long long)1 << (p.N * p.LQ)) * p.O * 4);   // < 2^31 (fill_big)
  const __amdgpu_buffer_rsrc_t rs_core = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(core), 0, (int)core_bytes, 0x00020000);
  unsigned voff[PER];   // byte offset of element i, or past the core's end where its padded output index is not real
#pragma unroll
  for (int i = 0; i < PER; ++i) voff[i] = ((okmask >> i) & 1u) ? coff[i] * 4u : core_bytes;
  auto stage_fetch = [&](int mt, int st) {
    unsigned u;
    if (p.xo) {
      // k0 = 128 st = o0 * Kh + kh0 (128 and Kh are powers of two: no carry into the per-thread part)
      const unsigned k0 = (unsigned)st * 128u, kh0 = k0 & ((1u << p.lkh) - 1u), o0 = k0 >> p.lkh;
      if (MODE == MODE_FWD) u = k0 * p.BnO + (unsigned)mt * 32u;
      else if (MODE == MODE_G0) u = (unsigned)mt * 32u * p.BnO + kh0 * p.O + o0;
      else u = kh0 * p.BnO + (unsigned)mt * 32u * p.O + o0;
    } else if (MODE == MODE_FWD) u = (unsigned)st * 128u * p.BnO + (unsigned)mt * (32 >> p.LOGO) * p.O;
    else if (MODE == MODE_G0) u = (unsigned)mt * 32u * p.BnO + (unsigned)st * (128 >> p.LOGO) * p.O;
    else u = (unsigned)st * (128 >> p.LOGO) * p.BnO + (unsigned)mt * 32u * p.O;
    const unsigned ub = (unsigned)__builtin_amdgcn_readfirstlane((int)(u * 4u));
    const int kleft = p.kdim - st * BC_KSTG * 2, rleft = p.rows - mt * 32;   // valid k-values / rows from this stage / tile on
    if (kleft >= BC_KSTG * 2 && rleft >= 32) {
      // a whole stage of a whole row tile (all but the last of either): the byte offsets were masked once, at kernel start -
      // 16 loads and nothing else (the per-element validity below is ~5 instructions an element, and a VALU-heavy phase
      // crawls while the co-resident workgroup keeps the SIMD's issue busy with MFMAs)
#pragma unroll
      for (int i = 0; i < PER; ++i)
        pre[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_core, voff[i], ub, 0));
      return;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 64 * BC_WAVES * i;
      const int row = MODE == MODE_G0 ? e >> 7 : e & 31;
      const int kl = MODE == MODE_G0 ? e & 127 : e >> 5;
      const bool ok = ((okmask >> i) & 1u) && kl < kleft && row < rleft;
      pre[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_core, ok ? coff[i] * 4u : core_bytes, ub, 0));
    }
  };
  // element i of a thread sits a constant step behind element i - 1 in the stage image (rows 2 i further on for G0, k-values
  // 8 i for the others): one address and 16 immediate offsets
  float* const commit0 = stage + (MODE == MODE_G0 ? (tid & 127) * BC_SROW + (tid >> 7) : (tid >> 5) * BC_SROW + (tid & 31));
  constexpr int commit_step = MODE == MODE_G0 ? 64 * BC_WAVES / 128 : (64 * BC_WAVES / 32) * BC_SROW;
  auto stage_commit = [&](int buf) {
    float* dst = commit0 + buf * BC_KSTG * 2 * BC_SROW;
#pragma unroll
    for (int i = 0; i < PER; ++i) dst[i * commit_step] = pre[i];
  };

  BC_T(t_pro);
  BC_ACC(1, t_start, t_pro);
  const int mt_begin = blockIdx.y * p.mt_per_rg;
  const int mt_end = mt_begin + p.mt_per_rg < mtiles ? mt_begin + p.mt_per_rg : mtiles;
  for (int mt = mt_begin; mt < mt_end; ++mt) {
    f32x16 acc[BC_NT];
#pragma unroll
    for (int nt = 0; nt < BC_NT; ++nt)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[nt][v] = 0.f;

    BC_T(t_f0);
    stage_fetch(mt, 0);
    __syncthreads();  // previous tile's readers are done with both buffers
    stage_commit(0);
    BC_T(t_f1);
    BC_ACC(7, t_f0, t_f1);
    // hi product of block hb (xo, G modes: times dY[w, o] of the block's o)
    auto hi_of = [&](int hb, int nt) {
      const int wl = (wv * BC_NT + nt) * 32 + wl32;
      const int hidx = (MODE != MODE_FWD && p.xo) ? (hb & ((1 << p.lnhbo) - 1)) : hb;
      float v;
      if constexpr (ND > 0) v = kr_exact<BC_WPB, ND>(xs, p, p.khalf_first, hidx, wl);
      else v = kr_flat<BC_WPB>(xs, p, p.khalf_first, p.khalf_n - p.mk, hidx, wl, NQ);
      if (MODE != MODE_FWD && p.xo) v *= dys[(hb >> p.lnhbo) * BC_WPB + wl];
      return v;
    };
    // the hi products of both column tiles as a pair ({hi, hi} with one tile)
    auto hi_pair = [&](int hb) {
      if constexpr (MODE == MODE_FWD && BC_NT == 2 && ND > 0) {
        return kr_exact2<BC_WPB, ND>(xs, p, p.khalf_first, hb, wv * BC_NT * 32 + wl32);
      } else {
        f32x2 r;
        r.x = hi_of(hb, 0);
        r.y = BC_NT == 2 ? hi_of(hb, BC_NT - 1) : r.x;
        return r;
      }
    };
    f32x2 hi = hi_pair(0);
    for (int st = 0; st < nstage; ++st) {
      BC_T(t_b0);
      __syncthreads();  // stage st visible; buffer (st+1)&1 free
      BC_T(t_b1);
      BC_ACC(2, t_b0, t_b1);
      if (st + 1 < nstage) stage_fetch(mt, st + 1);
      BC_T(t_b2);
      BC_ACC(3, t_b1, t_b2);
      const float* sb = stage + (st & 1) * BC_KSTG * 2 * BC_SROW + h * BC_SROW + wl32;
      int nhb_here = p.nhb - st * hb_per_stage;
      if (nhb_here > hb_per_stage) nhb_here = hb_per_stage;
      float av[BC_TBL], avn[BC_TBL];  // matrix-operand values of the current / next hi block
      f32x2 hin;
#pragma unroll
      for (int t = 0; t < BC_TBL; ++t) av[t] = sb[2 * t * BC_SROW];
      // One hi block: software pipeline - the NEXT block's hi products and operand values (into hi_n / av_n) are
      // fetched while this block's MFMAs execute (LDS latency hidden behind the matrix pipe).  The generated operands
      // of a BATCH of MFMAs come first (8: 4 k-steps x 2 column tiles, packed multiplies of two k-steps), then the
      // batch back to back: every VALU instruction between MFMAs costs the matrix pipe ~9 cycles
      // (tools/probes/mfma_f32_rate.hip: a v_mul in front of every MFMA holds it at 0.85, batches of 8 at 0.93).
      constexpr int BT = BC_TBL / 2 < 4 ? BC_TBL / 2 : 4;   // k-steps per batch
      auto hi_block = [&](int hb, const float (&av_c)[BC_TBL], const f32x2& hi_c, float (&av_n)[BC_TBL], f32x2& hi_n) {
        const int hbi = st * hb_per_stage + hb;       // global hi-block index
        const int hbn = hbi + 1 < p.nhb ? hbi + 1 : hbi;
        const int hbl = hb + 1 < nhb_here ? hb + 1 : hb;
        auto mfma_batch = [&](int t0) {
          f32x2 bop[BC_NT][BT / 2];
#pragma unroll
          for (int t = 0; t < BT; t += 2)
#pragma unroll
            for (int nt = 0; nt < BC_NT; ++nt)
              bop[nt][t / 2] = nt == 0 ? pk_mul_lo(tab[nt][(t0 + t) / 2], hi_c) : pk_mul_hi(tab[nt][(t0 + t) / 2], hi_c);
          // An MFMA may read a VALU result 2 wait states after it at the earliest, and the compiler does not see through
          // the inline-asm multiplies: the gap is put in by hand, tied to the last product (with a 4-k-step table and one
          // column tile that multiply is the only one, right in front of its MFMA).
          asm volatile("s_nop 1" : "+v"(bop[BC_NT - 1][BT / 2 - 1]));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int nt = 0; nt < BC_NT; ++nt)
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_c[t0 + t], bop[nt][t / 2][t & 1], acc[nt], 0, 0, 0);
        };
#pragma unroll
        for (int t0 = 0; t0 < BC_TBL / 2; t0 += BT) mfma_batch(t0);
        hi_n = hi_pair(hbn);
#pragma unroll
        for (int t = 0; t < BC_TBL; ++t) av_n[t] = sb[2 * (hbl * BC_TBL + t) * BC_SROW];
#pragma unroll
        for (int t0 = BC_TBL / 2; t0 < BC_TBL; t0 += BT) mfma_batch(t0);
      };
      // two blocks per turn, the register sets swapping roles: no copies of the next block's values
      int hb = 0;
      for (; hb + 1 < nhb_here; hb += 2) {
        hi_block(hb, av, hi, avn, hin);
        hi_block(hb + 1, avn, hin, av, hi);
      }
      if (hb < nhb_here) {
        hi_block(hb, av, hi, avn, hin);
        hi = hin;   // (av is reloaded at the next stage's start)
      }
      BC_T(t_m1);
      BC_ACC(4, t_b2, t_m1);
      if (st + 1 < nstage) stage_commit((st + 1) & 1);
      BC_T(t_c1);
      BC_ACC(5, t_m1, t_c1);
    }
    BC_T(t_e0);

    if (MODE == MODE_FWD && p.zsave) {
      // accumulator register 4j + i of lane (wl32, h) is row mt*32 + 8j + 4h + i: row quad mt*8 + 2j + h
#pragma unroll
      for (int nt = 0; nt < BC_NT; ++nt) {
        const long long w = w_block + (wv * BC_NT + nt) * 32 + wl32;
        if (w < p.Wn) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 zv = {acc[nt][4 * j], acc[nt][4 * j + 1], acc[nt][4 * j + 2], acc[nt][4 * j + 3]};
            __builtin_nontemporal_store(zv, (f32x4*)(p.zsave + ((long long)(mt * 8 + 2 * j + h) * p.Wn + w) * 4));
          }
        }
      }
    }

    // ---- epilogue of this row tile
    if (MODE == MODE_FWD && p.xo) {
      // rows are (b, o) in memory order: slot = o = R mod O picked by a select chain (no dynamic
      // register indexing); both lane halves hold partial sums of every o
      constexpr int OPT = (1 << LOGO_T) < 16 ? (1 << LOGO_T) : 16;
#pragma unroll
      for (int nt = 0; nt < BC_NT; ++nt) {
        const int wl = (wv * BC_NT + nt) * 32 + wl32;
        // the rows of an accumulator quad are consecutive (R0 .. R0 + 3, O >= 3: at most one step from b to b + 1 inside it):
        // one division and the row-half products of b and b + 1 per quad, not one of each per row
        const int bmax = (p.rows - 1) / p.O;   // (uniform) b + 1 past the last b is never a real row's
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int R0 = mt * 32 + 4 * h + 8 * j;
          const int bq = (int)__umulhi((unsigned)R0, p.odiv_m), oq0 = R0 - bq * p.O;
          const float pa = kr<BC_WPB>(xs, p, p.rhalf_first, p.rhalf_n, bq < bmax ? bq : bmax, wl);
          const float pb = kr<BC_WPB>(xs, p, p.rhalf_first, p.rhalf_n, bq + 1 < bmax ? bq + 1 : bmax, wl);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool wrap = oq0 + i >= p.O;
            const int oq = wrap ? oq0 + i - p.O : oq0 + i;
            const float val = R0 + i < p.rows ? acc[nt][4 * j + i] * (wrap ? pb : pa) : 0.f;
#pragma unroll
            for (int oo = 0; oo < OPT; ++oo) oacc[nt][oo] += oo == oq ? val : 0.f;
          }
        }
      }
    } else if (MODE == MODE_FWD) {
      constexpr int OPT = 1 << LOGO_T;
      constexpr int STEPV = OPT <= 4 ? OPT : OPT / 2;
#pragma unroll
      for (int nt = 0; nt < BC_NT; ++nt) {
        const int wl = (wv * BC_NT + nt) * 32 + wl32;
        float p1 = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          if (v % STEPV == 0) {
            // b = R >> LOGO_T of the lane's row R = mt*32 + (v & 3) + 4 h + 8 (v >> 2) is the same for both lane halves from
            // OP = 8 on: written from the tile index, so that the digits (and the whole address but the window) are
            // SCALAR - 2 VALU instructions per digit instead of 6 (8 binary digits per product in a first layer's 4 x 4
            // window, 8 products per row tile)
            // more than 4 digits (first layers: 8 binary digits): all reads ahead of the first multiply - digit by digit
            // the product is a chain of 8 LDS round trips, ~1 000 cycles, 8 times per row tile of ~16 000
            const bool flat = p.rhalf_n > 4;
            if constexpr (LOGO_T >= 3) {
              const int b0 = (mt * 32 + (v & 3) + 8 * (v >> 2)) >> LOGO_T;
              if (b0 >= (p.rows >> LOGO_T)) p1 = 0.f;
              else if (flat) p1 = kr_flat<BC_WPB>(xs, p, p.rhalf_first, p.rhalf_n, b0, wl, NQ);
              else p1 = kr<BC_WPB>(xs, p, p.rhalf_first, p.rhalf_n, b0, wl);
            } else {   // (both candidates from scalar digits and a select: measured no faster than the per-lane digits)
              const int R = mt * 32 + (v & 3) + 4 * h + 8 * (v >> 2);
              const int Rc = R < p.rows ? R : 0;
              const float pv = flat ? kr_flat<BC_WPB>(xs, p, p.rhalf_first, p.rhalf_n, Rc >> LOGO_T, wl, NQ)
                                    : kr<BC_WPB>(xs, p, p.rhalf_first, p.rhalf_n, Rc >> LOGO_T, wl);
              p1 = R < p.rows ? pv : 0.f;
            }
          }
          constexpr int dummy = 0;
          (void)dummy;
          const int slot = OPT <= 4 ? (v & (OPT - 1)) : ((v & 3) | (((v >> 2) & (OPT / 8 - 1)) << 2));
          oacc[nt][slot] += acc[nt][v] * p1;
        }
      }
    } else {
      // dL/dP[w][R] -> per-factor gradients of the row half by leave-one-out products
      const int nf = p.rhalf_n;
#pragma unroll
      for (int nt = 0; nt < BC_NT; ++nt) {
        const int wl = (wv * BC_NT + nt) * 32 + wl32;
        float* gcol = gxs + ((wv * BC_NT + nt) * 64 + lane);
        const int gstride = 64 * BC_WAVES * BC_NT;
        if (p.LQ >= 2) {
          // Q >= 4: the four rows of an accumulator quad (R0 .. R0 + 3, R0 a multiple of 4) differ in the last digit only.
          // The leave-one-out products of the other digits are formed once per quad and meet the quad's
          // sum_i g_i x_last[i]; the last digit's own gradient takes the product of all the others: ~16 VALU instructions
          // per value where the row-by-row form below takes ~60 (each costs the matrix pipe ~9 cycles, the co-resident
          // workgroup's MFMAs included).
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int R0 = mt * 32 + 4 * h + 8 * j;
            if (R0 < p.rows) {
              const int dl0 = R0 & (p.Q - 1);
              const float* xl = xs + ((p.rhalf_first + nf - 1) * p.Q + dl0) * BC_WPB + wl;
              float* gl = gcol + ((nf - 1) * p.Q + dl0) * gstride;
              float xlv[4], S = 0.f;
#pragma unroll
              for (int i = 0; i < 4; ++i) { xlv[i] = xl[i * BC_WPB]; S += acc[nt][4 * j + i] * xlv[i]; }
              float xv[BC_MAXD - 1], suf[BC_MAXD];
              int slot[BC_MAXD - 1];
#pragma unroll
              for (int d = 0; d < BC_MAXD - 1; ++d) {
                const int sh = d < nf - 1 ? (nf - 1 - d) * p.LQ : 0;
                const int dg = (R0 >> sh) & (p.Q - 1);
                slot[d] = d * p.Q + dg;
                xv[d] = xs[(d < nf - 1 ? (p.rhalf_first + d) * p.Q + dg : NQ) * BC_WPB + wl];
              }
              suf[BC_MAXD - 1] = 1.f;
#pragma unroll
              for (int d = BC_MAXD - 2; d >= 0; --d) suf[d] = suf[d + 1] * xv[d];
              float pre_p = 1.f;
#pragma unroll
              for (int d = 0; d < BC_MAXD - 1; ++d) {
                if (d < nf - 1) gcol[slot[d] * gstride] += pre_p * suf[d + 1] * S;
                pre_p *= xv[d];
              }
#pragma unroll
              for (int i = 0; i < 4; ++i) gl[i * gstride] += acc[nt][4 * j + i] * pre_p;
            }
          }
          continue;
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int R = mt * 32 + (v & 3) + 4 * h + 8 * (v >> 2);
          if (R < p.rows) {
            const float g = acc[nt][v];
            // leave-one-out products from one read per factor: suffix products stored, prefix running
            // (nf <= BC_MAXD; absent digits read the ones row, so every loop bound is compile time)
            float xv[BC_MAXD], suf[BC_MAXD + 1];
            int slot[BC_MAXD];
#pragma unroll
            for (int d = 0; d < BC_MAXD; ++d) {
              const int sh = d < nf ? (nf - 1 - d) * p.LQ : 0;
              const int dg = (R >> sh) & (p.Q - 1);
              slot[d] = d * p.Q + dg;
              xv[d] = xs[(d < nf ? (p.rhalf_first + d) * p.Q + dg : NQ) * BC_WPB + wl];
            }
            suf[BC_MAXD] = 1.f;
#pragma unroll
            for (int d = BC_MAXD - 1; d >= 0; --d) suf[d] = suf[d + 1] * xv[d];
            float pre_p = g;
#pragma unroll
            for (int d = 0; d < BC_MAXD; ++d) {
              if (d < nf) gcol[slot[d] * gstride] += pre_p * suf[d + 1];
              pre_p *= xv[d];
            }
          }
        }
      }
    }
    { BC_T(t_e1); BC_ACC(6, t_e0, t_e1); }
  }

#if defined(DCTN_STAMPS) && BC_PART != 0
  if (threadIdx.x == 0) {
    bc_acc[0] = __builtin_amdgcn_s_memtime() - t_start;
    const long long wg = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    if (wg < 16384)
      for (int i = 0; i < 8; ++i) bc_stamps[wg * 8 + i] = bc_acc[i];
  }
#endif
  // ---- results (slice blockIdx.y of the output: row groups are summed by a fixed-order reduce)
  if (MODE == MODE_FWD) {
    out += (long long)blockIdx.y * p.Wn * p.O;
#pragma unroll
    for (int nt = 0; nt < BC_NT; ++nt) {
      const long long w = w_block + (wv * BC_NT + nt) * 32 + wl32;
      constexpr int OPT = 1 << LOGO_T;
      if (p.xo) {
        constexpr int OS = OPT < 16 ? OPT : 16;
#pragma unroll
        for (int s2 = 0; s2 < OS; ++s2) {
          const float r = half_sum(oacc[nt][s2]);
          if (s2 < p.O && h == 0 && w < p.Wn) out[w * p.O + s2] = r;
        }
      } else if constexpr (OPT <= 4) {
#pragma unroll
        for (int s = 0; s < OPT; ++s) {
          const float r = half_sum(oacc[nt][s]);
          if (s < p.O && h == 0 && w < p.Wn) out[w * p.O + s] = r;
        }
      } else {
#pragma unroll
        for (int s = 0; s < OPT / 2; ++s) {
          const int o = (s & 3) + 4 * h + 8 * (s >> 2);
          if (o < p.O && w < p.Wn) out[w * p.O + o] = oacc[nt][s];
        }
      }
    }
  } else {
    // out = gxw[(factor*Q + q)][Wn] for the factors of the row half (sum of the two lane halves)
    __syncthreads();
    const int nfq = p.rhalf_n * p.Q;
    for (int e = tid; e < nfq * BC_WPB; e += 64 * BC_WAVES) {
      const int wl = e % BC_WPB, f = e / BC_WPB;
      const int grp = wl >> 5, l32 = wl & 31;
      const long long w = w_block + wl;
      if (w < p.Wn) {
        const float* g = gxs + (size_t)f * 64 * BC_WAVES * BC_NT + grp * 64 + l32;
        out[((long long)blockIdx.y * p.N * p.Q + p.rhalf_first * p.Q + f) * p.Wn + w] = g[0] + g[32];
      }
    }
  }
}

// ------------------------------------------------------------------------------------ dCore
// dCore[a][(b,o)] += sum_w P0[w][a] * (P1[w][b] dY[w][o]):  rows a, columns (b,o), k = windows.
// Both MFMA operands are generated: per window the Khatri-Rao halves are kept FACTORED in LDS
// (lo table x hi table, built once per window chunk from the window's features), a lane multiplies
// the two table entries of its row / column for the 2 windows of the k-step.  A wave owns
// DC_AT x DC_BT output tiles; window chunks are spread over grid.y and combined with float atomics.
constexpr int DC_AT = 2, DC_BT = 4;   // tiles per wave (rows, columns)
constexpr int DC_WR = 2, DC_WC2 = 4;  // waves per workgroup along rows / columns (8 waves: 128 x 512 outputs
                                      // per table build instead of 128 x 256)
constexpr int DC_THREADS = 64 * DC_WR * DC_WC2;
constexpr int DC_WC = 128;            // windows per LDS chunk
// The factor tables of a chunk are entry-major: row e holds entry e of every window of the chunk, the even windows
// (lane half 0's k-steps) first, then the odd ones; the rows are 2 floats longer than the chunk so that consecutive
// entries start 2 banks apart (one ds_read_b64 = an entry of two consecutive k-steps, conflict-free over the lanes'
// consecutive entries).  A lane's 10 row addresses are then constant over the chunk and the k-step is an immediate
// offset: no address arithmetic in the MFMA loop (every VALU instruction there costs ~9 cycles of the matrix pipe).
constexpr int DC_ROW = DC_WC + 2;
__device__ __forceinline__ int dc_pos(int wl) { return (wl & 1) * (DC_WC / 2) + (wl >> 1); }

struct DcoreP {
  int C, B, H, W, K, O, Q, LQ, N, n0, n1, Ho, Wo, OP, LOGO;
  long long Wn;
  long long s[5];
  int A, BN, cols;                 // cols = BN * O: column (b, o) = memory order of the core row
  int lb0, lb1;                    // bits of the lo tables of half 0 / half 1 (multiples of LQ)
  int nlo0, nhi0, nlo1, nhi1;      // table sizes
  int tstride;                     // table entries per window (T0lo | T0hi | T1lo | T1hi | dy | one always-zero entry)
  long long win_per_block;
  float* part;                     // per window-chunk slices [gridDim.y][A * cols] (plain stores, summed in a fixed order), or
                                   // NULL: float atomics into the zero-filled dCore
};

// PERX / PERY: register slots of the chunk prefetch (x features / dY values per thread): NQ <= 4 PERX,
// O <= 4 PERY.
#if defined(DCTN_STAMPS) && BC_PART == 0
// diagnostic build only (tools/stamp_bigcore.py dcore): cycles of wave 0 per phase: 0 total, 1 stage commit (+ barrier wait
// before it), 2 next chunk's fetch issue, 3 table build, 4 barrier after the build, 5 MFMA loop, 6 result store
__device__ unsigned long long dc_stamps[16384 * 8];
#define DC_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define DC_ACC(slot, a, b) do { if (threadIdx.x == 0) dc_acc[slot] += (b) - (a); } while (0)
extern "C" int dctn_debug_read_dc_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dc_stamps), (size_t)n * sizeof(unsigned long long));
}
#else
#define DC_T(var) do { } while (0)
#define DC_ACC(slot, a, b) do { } while (0)
#endif

template <int PERX, int PERY>
__global__ __launch_bounds__(DC_THREADS) void eps_bigcore_dcore_k(const float* __restrict__ x,
                                                           const float* __restrict__ dY,
                                                           float* __restrict__ dCore, DcoreP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NQ = p.N * p.Q;
  float* xs = smem;                         // [DC_WC][NQ + 1]
  float* tb = xs + DC_WC * (NQ + 1);        // [tstride][DC_ROW]: entries T0lo | T0hi | T1lo | T1hi | dy | 0, see DC_ROW
  const int o_t0lo = 0, o_t0hi = p.nlo0, o_t1lo = o_t0hi + p.nhi0, o_t1hi = o_t1lo + p.nlo1,
            o_dy = o_t1hi + p.nhi1;
  const int tid = threadIdx.x, lane = tid & 63, il = lane & 31, kk = lane >> 5, wv = tid >> 6;
  const int ntile_c = (p.cols + DC_WC2 * DC_BT * 32 - 1) / (DC_WC2 * DC_BT * 32);
  const int bt_a = blockIdx.x / ntile_c, bt_c = blockIdx.x % ntile_c;
  const int a_tile0 = (bt_a * DC_WR + wv / DC_WC2) * DC_AT;    // first row tile of this wave
  const int c_tile0 = (bt_c * DC_WC2 + wv % DC_WC2) * DC_BT;   // first column tile of this wave

  // per-lane table offsets of its rows / columns (constant over the whole kernel)
  int offa_lo[DC_AT], offa_hi[DC_AT];
  bool a_ok[DC_AT];
#pragma unroll
  for (int at = 0; at < DC_AT; ++at) {
    const int a = (a_tile0 + at) * 32 + il;
    a_ok[at] = a < p.A;
    const int ac = a_ok[at] ? a : 0;
    offa_lo[at] = o_t0lo + (ac & ((1 << p.lb0) - 1));
    offa_hi[at] = o_t0hi + (ac >> p.lb0);
  }
  int offb_lo[DC_BT], offb_hi[DC_BT], offb_dy[DC_BT];
  bool c_ok[DC_BT];
#pragma unroll
  for (int bt = 0; bt < DC_BT; ++bt) {
    const int col = (c_tile0 + bt) * 32 + il;
    const int b = col / p.O, o = col - b * p.O;
    c_ok[bt] = col < p.cols;
    const int bc = col < p.cols ? b : 0;
    offb_lo[bt] = o_t1lo + (bc & ((1 << p.lb1) - 1));
    offb_hi[bt] = o_t1hi + (bc >> p.lb1);
    offb_dy[bt] = o_dy + o;
  }

  f32x16 acc[DC_AT][DC_BT];
#pragma unroll
  for (int at = 0; at < DC_AT; ++at)
#pragma unroll
    for (int bt = 0; bt < DC_BT; ++bt)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[at][bt][v] = 0.f;

  const long long w_begin = (long long)blockIdx.y * p.win_per_block;
  long long w_end = w_begin + p.win_per_block;
  if (w_end > p.Wn) w_end = p.Wn;

  // ---- staging plan of this thread (constant over the kernel): window slot wl_s of every chunk,
  // elements nq = role + 4 i of its NQ features; role r also owns table r of that window
  constexpr int ROLES = DC_THREADS / DC_WC;      // 4
  const int wl_s = tid % DC_WC, role = __builtin_amdgcn_readfirstlane(tid / DC_WC);   // a wave has one role: scalar
  int foff[PERX];
  unsigned okx = 0;   // bit i: element i exists
#pragma unroll
  for (int i = 0; i < PERX; ++i) {
    const int nq = role + ROLES * i;
    const int n = nq / p.Q, q = nq - n * p.Q;
    const int pos = n / p.C, ch = n - pos * p.C;
    const int dh = pos / p.K, dw = pos - dh * p.K;
    foff[i] = nq < NQ ? (int)(ch * p.s[0] + dh * p.s[2] + dw * p.s[3] + q * p.s[4]) : 0;
    if (nq < NQ) okx |= 1u << i;
  }
  float prex[PERX], prey[PERY];
  auto fetch_chunk = [&](long long w0) {   // global loads of chunk w0 into registers (consumed a chunk later)
    const long long w = w0 + wl_s;
    const bool valid = w < w_end;
    const long long ww = valid ? w : 0;
    const int hw = p.Ho * p.Wo;
    const long long bb = ww / hw;
    const int rem = (int)(ww - bb * hw);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const float* px = x + bb * p.s[1] + (long long)ho * p.s[2] + (long long)wo * p.s[3];
#pragma unroll
    for (int i = 0; i < PERX; ++i) prex[i] = (valid && ((okx >> i) & 1u)) ? px[foff[i]] : 0.f;
#pragma unroll
    for (int i = 0; i < PERY; ++i) {
      const int o = role + ROLES * i;
      prey[i] = (valid && o < p.O) ? dY[ww * p.O + o] : 0.f;
    }
  };
  const int o_zero = o_dy + p.OP;   // one entry that is always 0: rows / columns outside the core point at it
#pragma unroll
  for (int at = 0; at < DC_AT; ++at)
    if (!a_ok[at]) { offa_lo[at] = o_zero; offa_hi[at] = o_zero; }
#pragma unroll
  for (int bt = 0; bt < DC_BT; ++bt)
    if (!c_ok[bt]) { offb_lo[bt] = o_zero; offb_hi[bt] = o_zero; offb_dy[bt] = o_zero; }

  // entries of the hi tables that this workgroup's rows / columns index
  const int hi0_lo = __builtin_amdgcn_readfirstlane((bt_a * DC_WR * DC_AT * 32) >> p.lb0);
  int hi0_hi = __builtin_amdgcn_readfirstlane((bt_a * DC_WR * DC_AT * 32 + DC_WR * DC_AT * 32 - 1) >> p.lb0);
  if (hi0_hi > p.nhi0 - 1) hi0_hi = p.nhi0 - 1;
  const int hi1_lo = __builtin_amdgcn_readfirstlane(((bt_c * DC_WC2 * DC_BT * 32) / p.O) >> p.lb1);
  int hi1_hi = __builtin_amdgcn_readfirstlane(((bt_c * DC_WC2 * DC_BT * 32 + DC_WC2 * DC_BT * 32 - 1) / p.O) >> p.lb1);
  if (hi1_hi > p.nhi1 - 1) hi1_hi = p.nhi1 - 1;
  // the lane's table rows, at its half's first k-step
  const float *ba_lo[DC_AT], *ba_hi[DC_AT], *bb_lo[DC_BT], *bb_hi[DC_BT], *bb_dy[DC_BT];
#pragma unroll
  for (int at = 0; at < DC_AT; ++at) {
    ba_lo[at] = tb + offa_lo[at] * DC_ROW + kk * (DC_WC / 2);
    ba_hi[at] = tb + offa_hi[at] * DC_ROW + kk * (DC_WC / 2);
  }
#pragma unroll
  for (int bt = 0; bt < DC_BT; ++bt) {
    bb_lo[bt] = tb + offb_lo[bt] * DC_ROW + kk * (DC_WC / 2);
    bb_hi[bt] = tb + offb_hi[bt] * DC_ROW + kk * (DC_WC / 2);
    bb_dy[bt] = tb + offb_dy[bt] * DC_ROW + kk * (DC_WC / 2);
  }
#if defined(DCTN_STAMPS) && BC_PART == 0
  unsigned long long dc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DC_T(t_start);
  fetch_chunk(w_begin);
  for (long long w0 = w_begin; w0 < w_end; w0 += DC_WC) {
    DC_T(t_a);
    __syncthreads();   // the previous chunk's MFMA loop is done with xs / tb
    {
      float* xw = xs + wl_s * (NQ + 1);
      float* tw = tb + dc_pos(wl_s);
#pragma unroll
      for (int i = 0; i < PERX; ++i)
        if ((okx >> i) & 1u) xw[role + ROLES * i] = prex[i];
#pragma unroll
      for (int i = 0; i < PERY; ++i) {
        const int o = role + ROLES * i;
        if (o < p.OP) tw[(o_dy + o) * DC_ROW] = prey[i];
      }
      if (role == 0) tw[o_zero * DC_ROW] = 0.f;
    }
    __syncthreads();
    DC_T(t_b);
    DC_ACC(1, t_a, t_b);
    if (w0 + DC_WC < w_end) fetch_chunk(w0 + DC_WC);   // in flight during the table build and the MFMA loop
    DC_T(t_c);
    DC_ACC(2, t_b, t_c);
    {
      // The four factored Khatri-Rao tables of window wl_s - of the two hi tables only the entries this workgroup's 128
      // rows / 512 columns index (8 of 64 and 6 of 16 for the 4^9 x 6 core) - dealt round-robin to the window's four
      // threads (role is wave-uniform: digits and loop bounds are scalar).  Every entry is the direct product of its
      // digits' features, U entries in flight: independent LDS reads and the same work for every wave (one table per
      // thread, doubled in place, left three of four wave pairs waiting for the one with the 64-entry table).
      const float* xw = xs + wl_s * (NQ + 1);
      float* tw = tb + dc_pos(wl_s);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int first, nf, off, e_lo, e_hi;   // entries [e_lo, e_hi] are needed
        if (t == 0) { nf = p.lb0 / p.LQ; first = p.n0 - nf; off = o_t0lo; e_lo = 0; e_hi = p.nlo0 - 1; }
        else if (t == 1) { nf = p.n0 - p.lb0 / p.LQ; first = 0; off = o_t0hi; e_lo = hi0_lo; e_hi = hi0_hi; }
        else if (t == 2) { nf = p.lb1 / p.LQ; first = p.N - nf; off = o_t1lo; e_lo = 0; e_hi = p.nlo1 - 1; }
        else { nf = p.n1 - p.lb1 / p.LQ; first = p.n0; off = o_t1hi; e_lo = hi1_lo; e_hi = hi1_hi; }
        constexpr int U = 4;
        for (int e0 = e_lo + role; e0 <= e_hi; e0 += ROLES * U) {
          float v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) v[u] = 1.f;
          for (int d = 0; d < nf; ++d) {
            const int sh = (nf - 1 - d) * p.LQ;
            const float* xd = xw + (first + d) * p.Q;
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int e = e0 + u * ROLES <= e_hi ? e0 + u * ROLES : e0;   // past the range: entry e0 again, not stored
              v[u] *= xd[(e >> sh) & (p.Q - 1)];
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (e0 + u * ROLES <= e_hi) tw[(off + e0 + u * ROLES) * DC_ROW] = v[u];
        }
      }
    }
    DC_T(t_d);
    DC_ACC(3, t_c, t_d);
    __syncthreads();
    DC_T(t_e);
    DC_ACC(4, t_d, t_e);
    // Two k-steps per turn: their factors are one ds_read_b64 per table entry and the products packed multiplies.
    // The reads of the next turn are issued before this turn's 16 MFMAs (a wave sits in their issue for ~1000
    // cycles; reads issued only after them would arrive with the pipe idle), its products stay behind them.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 ra[DC_AT][2], rz[DC_BT][3];
    auto read_factors = [&](int ks) {
#pragma unroll
      for (int at = 0; at < DC_AT; ++at) {
        ra[at][0] = *reinterpret_cast<const f32x2*>(ba_lo[at] + ks);
        ra[at][1] = *reinterpret_cast<const f32x2*>(ba_hi[at] + ks);
      }
#pragma unroll
      for (int bt = 0; bt < DC_BT; ++bt) {
        rz[bt][0] = *reinterpret_cast<const f32x2*>(bb_lo[bt] + ks);
        rz[bt][1] = *reinterpret_cast<const f32x2*>(bb_hi[bt] + ks);
        rz[bt][2] = *reinterpret_cast<const f32x2*>(bb_dy[bt] + ks);
      }
    };
    read_factors(0);
#pragma unroll
    for (int ks = 0; ks < DC_WC / 2; ks += 2) {
      f32x2 pa[DC_AT], pz[DC_BT];
#pragma unroll
      for (int at = 0; at < DC_AT; ++at) pa[at] = ra[at][0] * ra[at][1];
#pragma unroll
      for (int bt = 0; bt < DC_BT; ++bt) pz[bt] = rz[bt][0] * rz[bt][1] * rz[bt][2];
#ifndef DCTN_EXP_NOREAD
      if (ks + 2 < DC_WC / 2) read_factors(ks + 2);
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int at = 0; at < DC_AT; ++at)
#pragma unroll
          for (int bt = 0; bt < DC_BT; ++bt)
            acc[at][bt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[at][g], pz[bt][g], acc[at][bt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    DC_T(t_f);
    DC_ACC(5, t_e, t_f);
  }
  DC_T(t_g);

  // accumulate into dCore (zero-initialised by the launcher): one register = two 128-byte row segments
#pragma unroll
  for (int at = 0; at < DC_AT; ++at)
#pragma unroll
    for (int bt = 0; bt < DC_BT; ++bt) {
      const int col = (c_tile0 + bt) * 32 + il;
      if (col < p.cols) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int a = (a_tile0 + at) * 32 + (v & 3) + 8 * (v >> 2) + 4 * kk;
          if (a < p.A) {
            const long long e = (long long)a * p.cols + col;
            if (p.part) p.part[(long long)blockIdx.y * p.A * p.cols + e] = acc[at][bt][v];
            else atomicAdd(&dCore[e], acc[at][bt][v]);
          }
        }
      }
    }
#if defined(DCTN_STAMPS) && BC_PART == 0
  if (threadIdx.x == 0) {
    const unsigned long long t_h = __builtin_amdgcn_s_memtime();
    dc_acc[6] = t_h - t_g;
    dc_acc[0] = t_h - t_start;
    const long long wg = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    if (wg < 16384)
      for (int i = 0; i < 8; ++i) dc_stamps[wg * 8 + i] = dc_acc[i];
  }
#endif
}

int ilog2i(int v) {
  int r = 0;
  while ((1 << r) < v) ++r;
  return r;
}

bool fill_big(BigP& b, const EpsP& p, int mode) {
  if (p.Q < 2 || (p.Q & (p.Q - 1))) return false;
  b.C = p.C; b.B = p.B; b.H = p.H; b.W = p.W; b.K = p.K; b.O = p.O; b.Q = p.Q; b.N = p.N;
  b.LQ = ilog2i(p.Q);
  b.n0 = (p.N + 1) / 2; b.n1 = p.N - b.n0;
  if (b.n1 < 1) return false;
  if (b.n0 * b.LQ > 20 || b.n1 * b.LQ > 20) return false;
  if (b.n0 > BC_MAXD || b.n1 > BC_MAXD) return false;
  b.Ho = p.Ho; b.Wo = p.Wo; b.Wn = p.Wn;
  for (int i = 0; i < 5; ++i) b.s[i] = p.s[i];
  b.OP = 1;
  while (b.OP < p.O) b.OP <<= 1;
  if (b.OP < 2) b.OP = 2;
  if (b.OP > 32) return false;
  b.LOGO = ilog2i(b.OP);
  b.mode = mode;
  b.zsave = nullptr;
  const int A = 1 << (b.n0 * b.LQ), BN = 1 << (b.n1 * b.LQ);
  b.xo = (p.O >= 3 && (p.O & (p.O - 1)) != 0 && p.O <= 16) ? 1 : 0;
  b.odiv_m = (unsigned)(((1ull << 32) + p.O - 1) / p.O);
  b.lkh = 0; b.lnhbo = 0;
  int khalf_bits;
  if (b.xo) {
    if (mode == MODE_FWD) {
      b.rows = BN * p.O; b.kdim = A;
      b.khalf_first = 0; b.khalf_n = b.n0; b.rhalf_first = b.n0; b.rhalf_n = b.n1;
    } else if (mode == MODE_G0) {
      b.rows = A; b.kdim = p.O * BN; b.lkh = b.n1 * b.LQ;
      b.khalf_first = b.n0; b.khalf_n = b.n1; b.rhalf_first = 0; b.rhalf_n = b.n0;
    } else {
      b.rows = BN; b.kdim = p.O * A; b.lkh = b.n0 * b.LQ;
      b.khalf_first = 0; b.khalf_n = b.n0; b.rhalf_first = b.n0; b.rhalf_n = b.n1;
    }
  } else if (mode == MODE_FWD) {
    b.rows = BN * b.OP; b.kdim = A;
    b.khalf_first = 0; b.khalf_n = b.n0; b.rhalf_first = b.n0; b.rhalf_n = b.n1;
  } else if (mode == MODE_G0) {
    b.rows = A; b.kdim = BN * b.OP;
    b.khalf_first = b.n0; b.khalf_n = b.n1; b.rhalf_first = 0; b.rhalf_n = b.n0;
  } else {
    b.rows = BN; b.kdim = A * b.OP;
    b.khalf_first = 0; b.khalf_n = b.n0; b.rhalf_first = b.n0; b.rhalf_n = b.n1;
  }
  khalf_bits = b.khalf_n * b.LQ;
  (void)khalf_bits;
  // inner block: ID = Q^mk * (FWD ? 1 : OP) k-values, ID/2 <= BC_TBL table entries, mk >= 0 digits
  const int opk = (mode == MODE_FWD || b.xo) ? 1 : b.OP;
  int mk = 0;
  while (mk + 1 <= b.khalf_n && ((1 << ((mk + 1) * b.LQ)) * opk) / 2 <= BC_TBL_MAX) ++mk;
  b.mk = mk;
  b.ID = (1 << (mk * b.LQ)) * opk;
  if (b.ID < 2 || b.ID / 2 > BC_TBL_MAX) return false;
  b.tbl = b.ID / 2;
  if (b.tbl != 4 && b.tbl != 8 && b.tbl != 16) return false;
  b.nhb = b.kdim / b.ID;
  if (b.xo && mode != MODE_FWD) {
    if ((1 << b.lkh) < 128 && (128 % (1 << b.lkh)) != 0) return false;
    b.lnhbo = b.lkh - ilog2i(b.ID);   // hi blocks per o = Kh / ID
    if (b.lnhbo < 0) return false;
  }
  b.BnO = BN * p.O;
  if (p.R * p.O * 4 >= (1LL << 31)) return false;  // 32-bit byte offsets into the core (raw buffer loads)
  b.rg_count = 1;
  b.mt_per_rg = (b.rows + 31) / 32;
  return true;
}

// Split the row tiles over grid.y.  The grid runs in "rounds" of as many workgroups as are resident
// at once, each taking (row tiles per group + a fixed part for staging the window features) tile
// times: pick the split that minimises rounds x that time (a split that leaves the last round almost
// empty costs a whole round: 133 window blocks x 16 groups on 512 slots was 4.2 -> 5 rounds).
void choose_row_groups(BigP& b, int nt, int max_rg, size_t lds_bytes) {
  const long long wpb = (long long)BC_WAVES * nt * 32;
  const long long wblocks = (b.Wn + wpb - 1) / wpb;
  const int mtiles = (b.rows + 31) / 32;
  long long per_cu = lds_bytes > 0 ? (long long)(160 * 1024) / (long long)lds_bytes : 8;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  const long long capacity = 256 * per_cu;
  int top = mtiles < max_rg ? mtiles : max_rg;
  if (top < 1) top = 1;
  // rough cycle model: a row tile = kdim/2 k-steps x nt column tiles x 64 cycles at ~60 % matrix-pipe
  // efficiency; per workgroup ~8k cycles to stage the window features; per slice one write + read of
  // the partial result at ~1.2 KB/cycle
  const double tile = (double)(b.kdim / 2) * nt * 64.0 / 0.6;
  const double ovh = 8000.0;
  const double slice_bytes = b.mode == MODE_FWD ? (double)b.Wn * b.O * 4.0 : (double)b.Wn * b.N * b.Q * 4.0;
  const double slice = 2.0 * slice_bytes / 1200.0;
  double best = 1e300;
  int best_mt = mtiles;
  for (int rg = 1; rg <= top; ++rg) {
    const int mt_per = (mtiles + rg - 1) / rg;
    const int groups = (mtiles + mt_per - 1) / mt_per;
    const long long rounds = (wblocks * groups + capacity - 1) / capacity;
    const double cost = (double)rounds * ((double)mt_per * tile + ovh) + (groups > 1 ? slice * groups : 0.0);
    if (cost < best) { best = cost; best_mt = mt_per; }
  }
  b.mt_per_rg = best_mt;
  b.rg_count = (mtiles + best_mt - 1) / best_mt;
}

constexpr int BC_MAX_RG = 32;

// out[i] = sum_g part[g][i], fixed order.  16 bytes per lane and four slices' loads in flight per turn where the sizes allow
// (4-byte loads in a run-time loop over the slices: 14 us for ten slices of 2 MiB; the additions keep their order).
__global__ void bigcore_sum_slices_k(const float* __restrict__ part, float* __restrict__ out,
                                     long long n, int groups) {
  const bool vec = (n & 3) == 0 && (((uintptr_t)part | (uintptr_t)out) & 15) == 0;
  if (vec) {
    const long long n4 = n >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(part);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      int g = 0;
      for (; g + 4 <= groups; g += 4) {
        const float4 a = p4[(long long)g * n4 + i], b = p4[(long long)(g + 1) * n4 + i], c = p4[(long long)(g + 2) * n4 + i],
                     d = p4[(long long)(g + 3) * n4 + i];
        s.x = (((s.x + a.x) + b.x) + c.x) + d.x;
        s.y = (((s.y + a.y) + b.y) + c.y) + d.y;
        s.z = (((s.z + a.z) + b.z) + c.z) + d.z;
        s.w = (((s.w + a.w) + b.w) + c.w) + d.w;
      }
      for (; g < groups; ++g) {
        const float4 a = p4[(long long)g * n4 + i];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      }
      reinterpret_cast<float4*>(out)[i] = s;
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int g = 0; g < groups; ++g) s += part[g * n + i];
    out[i] = s;
  }
}

// dX from per-window factor gradients stored as slices gxw[part][N*Q][Wn]: the factors of half 0 (n < n0) fill
// `parts0` slices, those of half 1 `parts1` (1 when they come from the saved Z)
__global__ void bigcore_gather_dx_k(const float* __restrict__ gxw, float* __restrict__ dX, EpsP p,
                                    int n0, int parts0, int parts1) {
  const long long total = (long long)p.C * p.B * p.H * p.W * p.Q;
  const long long slice = (long long)p.N * p.Q * p.Wn;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    long long t = idx;
    const int q = (int)(t % p.Q); t /= p.Q;
    const int wi = (int)(t % p.W); t /= p.W;
    const int hi = (int)(t % p.H); t /= p.H;
    const int b = (int)(t % p.B);
    const int ch = (int)(t / p.B);
    float acc = 0.f;
    for (int dh = 0; dh < p.K; ++dh) {
      const int ho = hi - dh;
      if (ho < 0 || ho >= p.Ho) continue;
      for (int dw = 0; dw < p.K; ++dw) {
        const int wo = wi - dw;
        if (wo < 0 || wo >= p.Wo) continue;
        const long long win = ((long long)b * p.Ho + ho) * p.Wo + wo;
        const int n = (dh * p.K + dw) * p.C + ch;
        const long long e = (long long)(n * p.Q + q) * p.Wn + win;
        const int parts = n < n0 ? parts0 : parts1;
        for (int g = 0; g < parts; ++g) acc += gxw[g * slice + e];
      }
    }
    dX[idx] = acc;
  }
}


// ------------------------------------------------------------------------------------ dP1 from the saved Z
// With the forward's GEMM result kept (BigP::zsave), dL/dP1[w,b] = sum_o dY[w,o] Z[w,(b,o)] is a bandwidth pass over
// Z instead of the third GEMM G1 - what the reference's autograd does with the G it saved (dctn/eps.py:25-30).
// A workgroup takes 64 windows (lane = window: a row quad of 64 windows is 1 KiB contiguous), its waves split the b
// range; each lane turns its dP1 values into per-factor gradients of half 1 by leave-one-out products as the G
// epilogue does, in its own LDS column; the waves' columns are summed at the end into gxw slice 0.
struct Dp1P {
  int C, K, Q, LQ, N, n0, n1, Ho, Wo, O, OX;   // OX: Z rows per b (O, or the padded power of two)
  long long Wn;
  long long s[5];
  int S;                 // waves that share the b range
  int b_per_split;       // BN / S
};

constexpr int DP1_THREADS = 256;

constexpr int dp1_gcd4(int v) { return v % 4 == 0 ? 4 : (v % 2 == 0 ? 2 : 1); }

// OXT: OX at compile time (dY in registers, static row -> (b, o) map), 0 = run time (dY from LDS)
// ND1: leave-one-out digits (n1 padded with a row of ones)
template <int OXT, int ND1>
__global__ __launch_bounds__(DP1_THREADS) void eps_bigcore_dp1_k(const float* __restrict__ x,
                                                                 const float* __restrict__ Z,
                                                                 const float* __restrict__ dY,
                                                                 float* __restrict__ gxw, Dp1P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nq1 = p.n1 * p.Q;
  float* xs = smem;                       // [nq1 + 1][64]; row nq1 holds ones
  float* dys = xs + (nq1 + 1) * 64;       // [OX][64]
  float* gacc = dys + p.OX * 64;          // [nq1][DP1_THREADS]
  const int tid = threadIdx.x, wl = tid & 63, sp = tid >> 6;
  const long long w_block = (long long)blockIdx.x * 64;

  for (int e = tid; e < 64 * p.n1; e += DP1_THREADS) {
    const int wle = e & 63, d = e >> 6, n = p.n0 + d;
    const long long w = w_block + wle;
    const bool valid = w < p.Wn;
    const long long ww = valid ? w : 0;
    const int hw = p.Ho * p.Wo;
    const long long bb = ww / hw;
    const int rem = (int)(ww - bb * hw);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const int pos = n / p.C, ch = n - pos * p.C;
    const int dh = pos / p.K, dw = pos - dh * p.K;
    const float* px = x + ch * p.s[0] + bb * p.s[1] + (long long)(ho + dh) * p.s[2] + (long long)(wo + dw) * p.s[3];
    for (int q = 0; q < p.Q; ++q) xs[(d * p.Q + q) * 64 + wle] = valid ? px[q * p.s[4]] : 0.f;
  }
  if (tid < 64) xs[nq1 * 64 + tid] = 1.f;
  for (int e = tid; e < 64 * p.OX; e += DP1_THREADS) {
    const int wle = e & 63, o = e >> 6;
    const long long w = w_block + wle;
    dys[e] = (w < p.Wn && o < p.O) ? dY[w * p.O + o] : 0.f;
  }
  for (int f = 0; f < nq1; ++f) gacc[f * DP1_THREADS + tid] = 0.f;
  __syncthreads();

  const long long w = w_block + wl;
  if (sp < p.S && w < p.Wn) {
    float* gcol = gacc + tid;
    // dL/dP1[w][b] = g -> d/d(factor d of half 1, value digit_d(b)) += g * prod_(d' != d) x[d'][digit_d'(b)]
    auto flush = [&](int b, float g) {
      float xv[ND1], suf[ND1 + 1];
      int slot[ND1];
#pragma unroll
      for (int d = 0; d < ND1; ++d) {
        const bool real = d < p.n1;
        const int sh = real ? (p.n1 - 1 - d) * p.LQ : 0;
        const int dg = (b >> sh) & (p.Q - 1);
        slot[d] = d * p.Q + dg;
        xv[d] = xs[(real ? slot[d] : nq1) * 64 + wl];
      }
      suf[ND1] = 1.f;
#pragma unroll
      for (int d = ND1 - 1; d >= 0; --d) suf[d] = suf[d + 1] * xv[d];
      float pre_p = g;
#pragma unroll
      for (int d = 0; d < ND1; ++d) {
        if (d < p.n1) gcol[slot[d] * DP1_THREADS] += pre_p * suf[d + 1];
        pre_p *= xv[d];
      }
    };
    const int b0 = sp * p.b_per_split;
    const f32x4* zp = (const f32x4*)Z + ((long long)b0 * p.OX / 4) * p.Wn + w;   // quad q of the split: zp[q * Wn]
    if constexpr (OXT > 0) {
      constexpr int L = OXT * 4 / dp1_gcd4(OXT);   // rows per iteration: whole quads and whole b's
      constexpr int NQD = L / 4, NB = L / OXT;
      constexpr int U = NQD >= 8 ? 1 : 8 / NQD;    // iterations whose loads are in flight together
      float dyr[OXT];
#pragma unroll
      for (int o = 0; o < OXT; ++o) dyr[o] = dys[o * 64 + wl];
      const int iters = p.b_per_split / NB;
      int b = b0;
      for (int it = 0; it < iters; it += U) {
        f32x4 z[U][NQD];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int itc = it + u < iters ? it + u : iters - 1;
#pragma unroll
          for (int k = 0; k < NQD; ++k) z[u][k] = __builtin_nontemporal_load(zp + (long long)(itc * NQD + k) * p.Wn);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (it + u < iters) {
            float dp[NB];
#pragma unroll
            for (int bl = 0; bl < NB; ++bl) dp[bl] = 0.f;
#pragma unroll
            for (int e = 0; e < L; ++e) dp[e / OXT] = fmaf(dyr[e % OXT], z[u][e / 4][e % 4], dp[e / OXT]);
#pragma unroll
            for (int bl = 0; bl < NB; ++bl) flush(b + bl, dp[bl]);
            b += NB;
          }
        }
      }
    } else {
      constexpr int U = 8;
      const int quads = p.b_per_split * p.OX / 4;
      int b = b0, o = 0;
      float dp = 0.f;
      for (int q0 = 0; q0 < quads; q0 += U) {
        f32x4 z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int qc = q0 + u < quads ? q0 + u : quads - 1;
          z[u] = __builtin_nontemporal_load(zp + (long long)qc * p.Wn);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (q0 + u < quads) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              dp = fmaf(dys[o * 64 + wl], z[u][i], dp);
              if (++o == p.OX) { flush(b, dp); ++b; o = 0; dp = 0.f; }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < nq1 * 64; e += DP1_THREADS) {
    const int f = e >> 6, wle = e & 63;
    const long long we = w_block + wle;
    if (we < p.Wn) {
      float sum = 0.f;
      for (int s2 = 0; s2 < p.S; ++s2) sum += gacc[f * DP1_THREADS + s2 * 64 + wle];
      gxw[((long long)p.n0 * p.Q + f) * p.Wn + we] = sum;
    }
  }
}

size_t big_lds(const BigP& b) {
  const int nt = b.mode == MODE_FWD ? BC_NT_FWD : BC_NT_G;
  const size_t wpb = (size_t)BC_WAVES * nt * 32;
  size_t f = ((size_t)b.N * b.Q + 1) * wpb + 2 * BC_KSTG * 2 * BC_SROW;
  if (b.mode != MODE_FWD) f += (size_t)b.OP * wpb + (size_t)b.rhalf_n * b.Q * 64 * BC_WAVES * nt;
  return f * sizeof(float);
}

template <int MODE, int NT, int LOGO_T, int TBL, int ND>
int launch_nd(const void* x, const void* core, const void* dY, void* out, const BigP& b, size_t lds,
              hipStream_t st) {
  constexpr int WPB = BC_WAVES * NT * 32;
  const unsigned grid = (unsigned)((b.Wn + WPB - 1) / WPB);
  (void)hipFuncSetAttribute((const void*)eps_bigcore_k<MODE, NT, LOGO_T, TBL, ND>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((eps_bigcore_k<MODE, NT, LOGO_T, TBL, ND>), dim3(grid, b.rg_count), dim3(64 * BC_WAVES), lds,
                     st, (const float*)x, (const float*)core, (const float*)dY, (float*)out, b);
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

template <int MODE, int NT, int LOGO_T, int TBL>
int launch_one(const void* x, const void* core, const void* dY, void* out, const BigP& b, size_t lds,
               hipStream_t st) {
  switch (b.khalf_n - b.mk) {   // hi digits of the generated operand
    case 2: return launch_nd<MODE, NT, LOGO_T, TBL, 2>(x, core, dY, out, b, lds, st);
    case 3: return launch_nd<MODE, NT, LOGO_T, TBL, 3>(x, core, dY, out, b, lds, st);
  }
  return launch_nd<MODE, NT, LOGO_T, TBL, -1>(x, core, dY, out, b, lds, st);
}

template <int MODE, int NT, int LOGO_T>
int launch_tbl(const void* x, const void* core, const void* dY, void* out, const BigP& b, size_t lds,
               hipStream_t st) {
  switch (b.tbl) {
    case 4: return launch_one<MODE, NT, LOGO_T, 4>(x, core, dY, out, b, lds, st);
    case 8: return launch_one<MODE, NT, LOGO_T, 8>(x, core, dY, out, b, lds, st);
    case 16: return launch_one<MODE, NT, LOGO_T, 16>(x, core, dY, out, b, lds, st);
  }
  return DCTN_ERR_UNSUPPORTED;
}

template <int LOGO_T>
int launch_fwd(const void* x, const void* core, void* out, const BigP& b, size_t lds, hipStream_t st) {
  return launch_tbl<MODE_FWD, BC_NT_FWD, LOGO_T>(x, core, nullptr, out, b, lds, st);
}

}  // namespace

#if BC_PART == 1
#ifdef DCTN_STAMPS
extern "C" int dctn_debug_read_bc_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bc_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif
int dctn_bc::launch_fwd_hi(const void* x, const void* core, void* out, const BigP& b, size_t lds, hipStream_t st) {
  switch (b.LOGO) {
    case 3: return launch_fwd<3>(x, core, out, b, lds, st);
    case 4: return launch_fwd<4>(x, core, out, b, lds, st);
    case 5: return launch_fwd<5>(x, core, out, b, lds, st);
  }
  return DCTN_ERR_UNSUPPORTED;
}
#elif BC_PART == 2
#ifdef DCTN_STAMPS
extern "C" int dctn_debug_read_bc_stamps_g(unsigned long long* host, int n) {   // the transposed (dX) launches' stamps
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bc_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif
int dctn_bc::launch_g(int mode, const void* x, const void* core, const void* dY, void* out, const BigP& b,
                      size_t lds, hipStream_t st) {
  if (mode == MODE_G0) return launch_tbl<MODE_G0, BC_NT_G, 0>(x, core, dY, out, b, lds, st);
  return launch_tbl<MODE_G1, BC_NT_G, 0>(x, core, dY, out, b, lds, st);
}
#else

// worthwhile only when the core is large: small cores stay on the register family / generic path
static bool bigcore_wanted(const EpsP& p) {
  return p.R * p.O >= 1024;
}

bool eps_bigcore_covers(const EpsP& p, int dtype, int precision) {
  if (dtype != DCTN_F32 || precision != DCTN_PREC_EXACT || !bigcore_wanted(p)) return false;
  BigP b;
  return fill_big(b, p, MODE_FWD);
}

size_t eps_fwd_bigcore_workspace(const EpsP& p, int dtype, int precision) {
  if (dtype != DCTN_F32 || precision != DCTN_PREC_EXACT || !bigcore_wanted(p)) return 0;
  BigP b;
  if (!fill_big(b, p, MODE_FWD)) return 0;
  choose_row_groups(b, BC_NT_FWD, BC_MAX_RG, big_lds(b));
  return b.rg_count > 1 ? (size_t)b.rg_count * p.Wn * p.O * sizeof(float) : 0;
}

// plan of the dP1 pass over a saved Z (false: the shape keeps nothing and the backward runs G1)
static bool dp1_plan(const EpsP& p, const BigP& bf, Dp1P& d, size_t& lds, size_t& zbytes) {
  d.C = p.C; d.K = p.K; d.Q = p.Q; d.LQ = bf.LQ; d.N = p.N; d.n0 = bf.n0; d.n1 = bf.n1; d.Ho = p.Ho; d.Wo = p.Wo;
  d.O = p.O; d.OX = bf.xo ? p.O : bf.OP;
  d.Wn = p.Wn;
  for (int i = 0; i < 5; ++i) d.s[i] = p.s[i];
  const int BN = 1 << (bf.n1 * bf.LQ);
  d.S = 0;
  for (int S = 4; S >= 1; S >>= 1)
    if (BN % S == 0 && ((long long)(BN / S) * d.OX) % 4 == 0) { d.S = S; break; }
  if (d.S == 0) return false;
  d.b_per_split = BN / d.S;
  lds = ((size_t)(d.n1 * d.Q + 1) * 64 + (size_t)d.OX * 64 + (size_t)d.n1 * d.Q * DP1_THREADS) * sizeof(float);
  if (lds > 64 * 1024) return false;
  const size_t mtiles = ((size_t)bf.rows + 31) / 32;
  zbytes = mtiles * 32 * (size_t)p.Wn * sizeof(float);
  return zbytes <= ((size_t)16 << 30);   // bounded: beyond 16 GiB the backward recomputes
}

size_t eps_bigcore_saved_bytes(const EpsP& p, int dtype, int precision) {
  if (dtype != DCTN_F32 || precision != DCTN_PREC_EXACT || !bigcore_wanted(p)) return 0;
  BigP b;
  if (!fill_big(b, p, MODE_FWD) || big_lds(b) > DCTN_LDS_BUDGET) return 0;
  BigP b0;
  if (!fill_big(b0, p, MODE_G0) || big_lds(b0) > DCTN_LDS_BUDGET) return 0;
  Dp1P d;
  size_t lds, zbytes;
  if (!dp1_plan(p, b, d, lds, zbytes)) return 0;
  return zbytes;
}

int eps_fwd_bigcore(const void* x, const void* core, void* out, void* ws, size_t ws_bytes,
                    const EpsP& p, int dtype, int precision, hipStream_t st, void* zsave) {
  if (dtype != DCTN_F32 || precision != DCTN_PREC_EXACT || !bigcore_wanted(p)) return DCTN_ERR_UNSUPPORTED;
  BigP b;
  if (!fill_big(b, p, MODE_FWD)) return DCTN_ERR_UNSUPPORTED;
  const size_t lds = big_lds(b);
  if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;
  b.zsave = (float*)zsave;
  choose_row_groups(b, BC_NT_FWD, BC_MAX_RG, lds);
  const size_t need = b.rg_count > 1 ? (size_t)b.rg_count * p.Wn * p.O * sizeof(float) : 0;
  if (need > 0 && (!ws || ws_bytes < need)) {  // no scratch: keep every row tile in one workgroup
    b.rg_count = 1;
    b.mt_per_rg = (b.rows + 31) / 32;
  }
  void* dst = b.rg_count > 1 ? ws : out;
  int rc = DCTN_ERR_UNSUPPORTED;
  switch (b.LOGO) {
    case 1: rc = launch_fwd<1>(x, core, dst, b, lds, st); break;
    case 2: rc = launch_fwd<2>(x, core, dst, b, lds, st); break;
    case 3: case 4: case 5: rc = dctn_bc::launch_fwd_hi(x, core, dst, b, lds, st); break;
  }
  if (rc != DCTN_OK) return rc;
  if (b.rg_count > 1) {
    const long long n = p.Wn * p.O;
    const long long nt = (n + 3) / 4;   // (a thread takes four values where the sizes allow)
    const unsigned g = (unsigned)((nt + 255) / 256 < 2048 ? (nt + 255) / 256 : 2048);
    hipLaunchKernelGGL(bigcore_sum_slices_k, dim3(g), dim3(256), 0, st, (const float*)ws, (float*)out, n,
                       b.rg_count);
    DCTN_CHECK_LAUNCH();
  }
  dctn_set_last_kernel(zsave ? "eps_fwd_mfma_bigcore_f32_saving" : "eps_fwd_mfma_bigcore_f32");
  return DCTN_OK;
}

template <int OXT>
static int launch_dp1_nd(const void* x, const void* Z, const void* dY, float* gxw, const Dp1P& d, size_t lds, hipStream_t st) {
  const unsigned grid = (unsigned)((d.Wn + 63) / 64);
#define DP1_GO(ND)                                                                                                 \
  do {                                                                                                             \
    (void)hipFuncSetAttribute((const void*)eps_bigcore_dp1_k<OXT, ND>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                              (int)lds);                                                                           \
    hipLaunchKernelGGL((eps_bigcore_dp1_k<OXT, ND>), dim3(grid), dim3(DP1_THREADS), lds, st, (const float*)x,      \
                       (const float*)Z, (const float*)dY, gxw, d);                                                 \
  } while (0)
  if (d.n1 <= 2) DP1_GO(2);
  else if (d.n1 <= 4) DP1_GO(4);
  else DP1_GO(8);
#undef DP1_GO
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

static int launch_dp1(const void* x, const void* Z, const void* dY, float* gxw, const Dp1P& d, size_t lds, hipStream_t st) {
  switch (d.OX) {
    case 2: return launch_dp1_nd<2>(x, Z, dY, gxw, d, lds, st);
    case 4: return launch_dp1_nd<4>(x, Z, dY, gxw, d, lds, st);
    case 6: return launch_dp1_nd<6>(x, Z, dY, gxw, d, lds, st);
    case 8: return launch_dp1_nd<8>(x, Z, dY, gxw, d, lds, st);
    case 16: return launch_dp1_nd<16>(x, Z, dY, gxw, d, lds, st);
  }
  return launch_dp1_nd<0>(x, Z, dY, gxw, d, lds, st);
}

// dX through the two transposed GEMMs G0, G1: per-window factor gradients into
// gxw[row group][N*Q][Wn], then a deterministic gather (sum over row groups and over the K*K
// windows covering each pixel).
static bool dfactor_plan(const EpsP& p, BigP& b0, BigP& b1) {
  if (!fill_big(b0, p, MODE_G0) || !fill_big(b1, p, MODE_G1)) return false;
  if (big_lds(b0) > DCTN_LDS_BUDGET || big_lds(b1) > DCTN_LDS_BUDGET) return false;
  const int mt0 = (b0.rows + 31) / 32, mt1 = (b1.rows + 31) / 32;
  int cap = mt0 < mt1 ? mt0 : mt1;
  if (cap > BC_MAX_RG) cap = BC_MAX_RG;
  choose_row_groups(b0, BC_NT_G, cap, big_lds(b0));
  choose_row_groups(b1, BC_NT_G, cap, big_lds(b1));
  // both halves must fill the same number of slices: take the smaller count for both
  const int rg = b0.rg_count < b1.rg_count ? b0.rg_count : b1.rg_count;
  b0.mt_per_rg = (mt0 + rg - 1) / rg; b0.rg_count = (mt0 + b0.mt_per_rg - 1) / b0.mt_per_rg;
  b1.mt_per_rg = (mt1 + rg - 1) / rg; b1.rg_count = (mt1 + b1.mt_per_rg - 1) / b1.mt_per_rg;
  return b0.rg_count == b1.rg_count;
}

// G0 alone (half 1 comes from the saved Z): its own optimum split
static bool g0_plan(const EpsP& p, BigP& b0) {
  if (!fill_big(b0, p, MODE_G0) || big_lds(b0) > DCTN_LDS_BUDGET) return false;
  choose_row_groups(b0, BC_NT_G, BC_MAX_RG, big_lds(b0));
  return true;
}

size_t eps_bwd_dfactor_bigcore_workspace(const EpsP& p, int dtype, int precision) {
  if (dtype != DCTN_F32 || precision != DCTN_PREC_EXACT || !bigcore_wanted(p)) return 0;
  BigP b0, b1;
  if (!dfactor_plan(p, b0, b1)) return 0;
  int rg = b0.rg_count;
  BigP g0;
  if (g0_plan(p, g0) && g0.rg_count > rg) rg = g0.rg_count;
  return (size_t)rg * p.N * p.Q * p.Wn * sizeof(float);
}

int eps_bwd_dx_bigcore(const void* x, const void* core, const void* dY, void* dX, void* ws,
                       size_t ws_bytes, const EpsP& p, int dtype, int precision, hipStream_t st,
                       const void* zsaved, size_t zsaved_bytes) {
  if (dtype != DCTN_F32 || precision != DCTN_PREC_EXACT || !bigcore_wanted(p)) return DCTN_ERR_UNSUPPORTED;
  const long long total = (long long)p.C * p.B * p.H * p.W * p.Q;
  const unsigned g2 = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (zsaved) {
    // the forward kept Z: half 0 by the GEMM G0, half 1 by one pass over Z
    BigP bf, b0;
    Dp1P d;
    size_t lds1, zbytes;
    if (fill_big(bf, p, MODE_FWD) && dp1_plan(p, bf, d, lds1, zbytes) && zsaved_bytes >= zbytes && g0_plan(p, b0)) {
      const size_t need = (size_t)b0.rg_count * p.N * p.Q * p.Wn * sizeof(float);
      if (!ws || ws_bytes < need) return DCTN_ERR_WORKSPACE;
      float* gxw = (float*)ws;
      int rc = dctn_bc::launch_g(MODE_G0, x, core, dY, gxw, b0, big_lds(b0), st);
      if (rc != DCTN_OK) return rc;
      rc = launch_dp1(x, zsaved, dY, gxw, d, lds1, st);
      if (rc != DCTN_OK) return rc;
      hipLaunchKernelGGL(bigcore_gather_dx_k, dim3(g2), dim3(256), 0, st, (const float*)gxw, (float*)dX, p, b0.n0,
                         b0.rg_count, 1);
      DCTN_CHECK_LAUNCH();
      dctn_set_last_kernel("eps_bwd_mfma_bigcore_f32_savedz");
      return DCTN_OK;
    }
  }
  BigP b0, b1;
  if (!dfactor_plan(p, b0, b1)) return DCTN_ERR_UNSUPPORTED;
  const size_t need = (size_t)b0.rg_count * p.N * p.Q * p.Wn * sizeof(float);
  if (!ws || ws_bytes < need) return DCTN_ERR_WORKSPACE;
  float* gxw = (float*)ws;
  int rc = dctn_bc::launch_g(MODE_G0, x, core, dY, gxw, b0, big_lds(b0), st);
  if (rc != DCTN_OK) return rc;
  rc = dctn_bc::launch_g(MODE_G1, x, core, dY, gxw, b1, big_lds(b1), st);
  if (rc != DCTN_OK) return rc;
  hipLaunchKernelGGL(bigcore_gather_dx_k, dim3(g2), dim3(256), 0, st, (const float*)gxw, (float*)dX, p, b0.n0,
                     b0.rg_count, b0.rg_count);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("eps_bwd_mfma_bigcore_f32");
  return DCTN_OK;
}

// plan of the dCore product: parameters, workgroup tiles, window chunks (grid.y), LDS bytes
static bool dcore_plan(const EpsP& p, int dtype, int precision, DcoreP& d, long long& tiles, long long& chunks, size_t& lds) {
  if (dtype != DCTN_F32 || precision != DCTN_PREC_EXACT || !bigcore_wanted(p)) return false;
  if (p.Q < 2 || (p.Q & (p.Q - 1))) return false;
  d.C = p.C; d.B = p.B; d.H = p.H; d.W = p.W; d.K = p.K; d.O = p.O; d.Q = p.Q; d.N = p.N;
  d.LQ = ilog2i(p.Q);
  d.n0 = (p.N + 1) / 2; d.n1 = p.N - d.n0;
  if (d.n1 < 1 || d.n0 * d.LQ > 20 || d.n1 * d.LQ > 20) return false;
  d.Ho = p.Ho; d.Wo = p.Wo; d.Wn = p.Wn;
  for (int i = 0; i < 5; ++i) d.s[i] = p.s[i];
  d.OP = 2;
  while (d.OP < p.O) d.OP <<= 1;
  if (d.OP > 32) return false;
  d.LOGO = ilog2i(d.OP);
  d.A = 1 << (d.n0 * d.LQ); d.BN = 1 << (d.n1 * d.LQ); d.cols = d.BN * p.O;
  // lo tables: as many whole digits as fit in 5 bits (32 entries), at least one digit
  auto lo_bits = [&](int nfac) {
    int m = 5 / d.LQ;
    if (m < 1) m = 1;
    if (m > nfac) m = nfac;
    return m * d.LQ;
  };
  d.lb0 = lo_bits(d.n0); d.lb1 = lo_bits(d.n1);
  d.nlo0 = 1 << d.lb0; d.nhi0 = d.A >> d.lb0; d.nlo1 = 1 << d.lb1; d.nhi1 = d.BN >> d.lb1;
  if (p.N * p.Q > 80 || d.OP > 32) return false;   // register staging plan of the kernel
  const int tstride = d.nlo0 + d.nhi0 + d.nlo1 + d.nhi1 + d.OP + 1;   // + the always-zero entry
  d.tstride = tstride;
  lds = ((size_t)(DC_WC * (p.N * p.Q + 1) + 1) / 2 * 2 + (size_t)DC_ROW * tstride) * sizeof(float);
  if (lds > DCTN_LDS_BUDGET) return false;
  const int ntile_a = (d.A + DC_WR * DC_AT * 32 - 1) / (DC_WR * DC_AT * 32);
  const int ntile_c = (d.cols + DC_WC2 * DC_BT * 32 - 1) / (DC_WC2 * DC_BT * 32);
  tiles = (long long)ntile_a * ntile_c;
  // one workgroup per CU (the kernel's LDS admits one): a workgroup's prologue, result store (128 accumulator values per
  // lane, one 64-bit address each) and its slice of the partial-sum pass are paid once per window chunk - with four rounds
  // of workgroups (1024 / tiles chunks) cfg3b's step took 2.39 ms, with one 2.12 ms (cfg3a 7.34 -> 7.08 ms)
  chunks = 256 / tiles;
  if (chunks < 1) chunks = 1;
  const long long max_chunks = (p.Wn + DC_WC - 1) / DC_WC;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks > 65535) chunks = 65535;
  long long wpb = (p.Wn + chunks - 1) / chunks;
  wpb = (wpb + DC_WC - 1) / DC_WC * DC_WC;
  chunks = (p.Wn + wpb - 1) / wpb;
  d.win_per_block = wpb;
  d.part = nullptr;
  return true;
}

// room for one dCore slice per window chunk: the chunks are then summed in a fixed order (bit-reproducible dCore)
size_t eps_bwd_dcore_bigcore_workspace(const EpsP& p, int dtype, int precision) {
  DcoreP d;
  long long tiles, chunks;
  size_t lds;
  if (!dcore_plan(p, dtype, precision, d, tiles, chunks, lds) || chunks < 2) return 0;
  return (size_t)chunks * p.R * p.O * sizeof(float);
}

int eps_bwd_dcore_bigcore(const void* x, const void* dY, void* dCore, const EpsP& p, int dtype,
                          int precision, hipStream_t st, void* ws, size_t ws_bytes) {
  DcoreP d;
  long long tiles, chunks;
  size_t lds;
  if (!dcore_plan(p, dtype, precision, d, tiles, chunks, lds)) return DCTN_ERR_UNSUPPORTED;
  const size_t slices = (size_t)chunks * p.R * p.O * sizeof(float);
  float* target = (float*)dCore;
  if (chunks < 2) {
    // one chunk: every element has one writer; (the kernel adds) start from zero
    if (dctn_zero_async(dCore, (size_t)p.R * p.O * sizeof(float), st) != DCTN_OK) return DCTN_ERR_LAUNCH;
  } else if (ws && ws_bytes >= slices) {
    d.part = (float*)ws;
  } else {
    // no room for the slices: float atomics in arrival order (the low bits of dCore then differ from run to run)
    if (dctn_zero_async(dCore, (size_t)p.R * p.O * sizeof(float), st) != DCTN_OK) return DCTN_ERR_LAUNCH;
  }
#define DC_LAUNCH(PX, PY)                                                                                  \
  do {                                                                                                     \
    (void)hipFuncSetAttribute((const void*)eps_bigcore_dcore_k<PX, PY>,                                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
    hipLaunchKernelGGL((eps_bigcore_dcore_k<PX, PY>), dim3((unsigned)tiles, (unsigned)chunks),            \
                       dim3(DC_THREADS), lds, st, (const float*)x, (const float*)dY, target, d);          \
  } while (0)
  if (p.N * p.Q <= 40 && d.OP <= 8) DC_LAUNCH(10, 2);
  else DC_LAUNCH(20, 8);
#undef DC_LAUNCH
  DCTN_CHECK_LAUNCH();
  if (d.part) {
    const long long n = (long long)p.R * p.O;
    const long long nt = (n + 3) / 4;
    const unsigned g = (unsigned)((nt + 255) / 256 < 4096 ? (nt + 255) / 256 : 4096);
    hipLaunchKernelGGL(bigcore_sum_slices_k, dim3(g), dim3(256), 0, st, (const float*)d.part, target, n, (int)chunks);
    DCTN_CHECK_LAUNCH();
  }
  dctn_set_last_kernel("eps_bwd_mfma_bigcore_f32");
  return DCTN_OK;
}
#endif  // BC_PART
