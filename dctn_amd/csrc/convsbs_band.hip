// ConvSBS forward and backward for bonds 5..16 (largest bond of the string) on the f32 matrix cores, one launch + one tail:
// float32 open chains of at most nine cores (the snakes of mnist.py:189-223), at most two output values, both on one
// middle core, q^C <= 4 - BASELINE cfg4 at r = 16 and r = 8 (bonds <= 8: two state values per lane, template NS = 2).
//
// Replaces torch autograd through dctn/conv_sbs.py:258-304 for these strings, and the round-2 backward of
// convsbs_mfma.hip (convsbs_bwd_mfma16_k: forward states through HBM - 81 MB written by the forward and read back -,
// per-window feature gradients through HBM, four helper launches, one wave per SIMD with 415 registers).  Here:
//   * a workgroup owns a BAND of window rows of ONE image.  It keeps the per-window feature gradients of the band in
//     LDS and writes the band's dX itself, in a fixed order; the max_h pixel rows two bands share go to a small side
//     buffer as two partial sums that the tail kernel adds.  No per-window gradient tensor, no zero-fill, no gather.
//   * nothing is kept by the forward: the chain is recomputed per 16-window tile and the input state of every core
//     stays in REGISTERS until the way back (4 registers per state: 64 for nine cores).
//   * TWO ROLES, one wave of each per SIMD (waves w and w + 4 of a 512-thread workgroup share a SIMD):
//       chain waves (0-3): forward sweep, adjoint sweep, d/d(features) - everything that is a dependent chain;
//       gradient waves (4-7): dCore += G^T (f v) with the WINDOWS on the k index; they own the dCore accumulators
//       (8 slots x q^C tiles x 4 registers + one tile each for the first / last core) across the whole band, and they
//       stage their chain wave's tiles (pixels, dY) two tiles ahead.  The chain wave hands over G and v of a pair through
//       LDS (two buffers, one workgroup barrier per hand-over, stored BEFORE the pair's product so that the barrier waits
//       while the matrix pipe works); the gradient wave runs one hand-over behind (operands of hand-over k fetched while
//       hand-over k - 1 is multiplied), and the chain wave is free of the 128-160 accumulator registers that held the old
//       kernel at one wave per SIMD.
//   * tiles per feature value instead of rows padded to four: U_qq[r', w] = sum_l core[o, l, r', qq] v[l, w] is one
//       16x16 tile per qq (v_mfma_f32_16x16x4_f32, K = l), so q = 3 costs 12 MFMAs per product, not 16, and q = 2 costs 8.
//       State layout: register s of lane (w, g) holds l = 4 g + s; the accumulator's register r of lane (w, g) is row
//       4 g + r, so  v'[r'] = sum_qq f[qq] U_qq[r']  lands in state layout: the chain never leaves registers.
//       Adjoint: W_qq[l, w] = sum_r' core[o, l, r', qq] G[r', w] from a second pack with the bond legs exchanged;
//       dv[l] = sum_qq f[qq] W_qq[l] (state layout again), df[qq] = sum_l v[l] W_qq[l].
//   * per-workgroup dCore records (the accumulator layout as it stands, 16-byte stores), summed in a fixed order and put
//     into the cores' layouts by the tail kernel: bit-reproducible gradients.
// Measured at BASELINE cfg4 r = 16 (115 200 windows): 108 us + 7 us tail (round 3: 161 us + 17 us of helpers; forward 50 -> 41 us
// without the state stores).  In-kernel stamps (tools/stamp_band.py): a 16-window tile takes a chain wave ~25 k cycles of
// which its MFMAs are 8.4 k and the gradient wave's 4.4 k; what was tried on that gap is in NOTEBOOK.md.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float bd_f4;
typedef __attribute__((ext_vector_type(2))) float bd_f2;
typedef __attribute__((ext_vector_type(2))) int bd_i2;

// diagnostic build only (make EXTRA=-DDCTN_STAMPS, tools/stamp_band.py): wave 0 (chain) and wave 4 (gradient) of every
// workgroup leave the shader-clock value of each phase boundary behind the side buffers in the workspace
#ifdef DCTN_STAMPS
#define BD_STAMP(k)                                                                                              \
  do {                                                                                                           \
    if (lane == 0 && (wv & 3) == 0 && p.stamps)                                                                  \
      p.stamps[((size_t)blockIdx.x * 2 + (wv >> 2)) * 32 + (k)] = (long long)__builtin_readcyclecounter();       \
  } while (0)
#else
#define BD_STAMP(k) do {} while (0)
#endif

namespace {

constexpr int BD_NC = 9;          // cores per string (compile-time unrolled; shorter strings run too)
constexpr int BD_NPK = BD_NC - 1; // packs / accumulator slots: middle core c -> slot c - 1, the second output value -> slot 7
constexpr int BD_TS = 20;         // floats per row of a hand-over tile [window][16]: the chain wave's b128 row writes and the
                                  // gradient wave's transposed b32 reads (rows 4 apart: 80 floats = 16 banks) are conflict-free
constexpr int BD_TILE = 16 * BD_TS;      // one hand-over tile (G or v)
constexpr int BD_FS = (BD_NC + 1) * 16 * 4;   // feature products of a 16-window tile: [core][window][4]; row 9: the windows' dY
constexpr int BD_LDS_BUDGET = 160 * 1024;      // gfx950: the whole LDS of a CU (one workgroup per CU); the run-time value: bd_dev()

// LDS per CU and CU count of the current device, asked once (the family is written for gfx950: 160 KiB, 256 CUs; on a
// device with less LDS the plans below then decline - DCTN_ERR_UNSUPPORTED, the caller takes the matrix-core sweep -
// instead of reporting `covers` and failing at the launch)
struct BdDev { int lds, cus; };
static BdDev bd_dev() {
  static const BdDev d = [] {
    BdDev r{BD_LDS_BUDGET, 256};
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && v >= 64 * 1024) r.lds = v;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) r.cus = v;
    }
    return r;
  }();
  return d;
}
constexpr int BD_THREADS = 512;
constexpr int BD_NFS = 3;         // feature buffers per chain wave: the tile on its way back, the next one, the one the gradient
                                  // wave may still be reading right behind the last barrier of the previous tile

struct BdP {
  const float* x;
  const float* dY;
  float* dX;            // (C, B, H, W, q) contiguous, or NULL
  float* records;       // [workgroup][core_off[n]] partial core gradients, or NULL (no core gradient wanted)
  float* side;          // partial sums of the pixel rows two bands share: [image][boundary][part][max_h][W][C q]
  const float* core[BD_NC];
  long long xs[5];
  int n, C, q, qc, B, H, W, Ho, Wo, Otot, max_h;
  int o[BD_NC], bl[BD_NC], br[BD_NC], ph[BD_NC], pw[BD_NC];
  int nin[BD_NC];       // input states of core c (1, or 2 behind the two-valued core)
  int c2;               // the core with two output values, or -1
  int nb, band_rows;    // bands per image, window rows per band (the last band of an image may be shorter)
  int iters;            // tiles per chain wave
  int core_off[BD_NC + 1];
  int packF_off, packA_off, first_off, last_off, fs_off, raw_off, gv_off, rows_off;   // LDS plan (float offsets)
  long long* stamps;    // diagnostic build only
};

struct BdTailP {
  float* dcore[BD_NC];
  int core_off[BD_NC + 1];
  int n, nrec, total, rec_len;
  int bl[BD_NC], br[BD_NC], o[BD_NC];
  int qc, c2;
  const float* records;
  const float* side;
  float* dX;
  int B, H, W, Cq, C, q, nb, band_rows, max_h;
  long long nshared;    // shared pixel-row elements: B (nb - 1) max_h W C q
};

__device__ __forceinline__ float bd_group_sum(float v) {   // sum over the four k groups (lanes w, w+16, w+32, w+48), in every lane
  const int iv = __float_as_int(v);
  const bd_i2 r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
  const float t = __int_as_float(r[0]) + __int_as_float(r[1]);
  const int it = __float_as_int(t);
  const bd_i2 r2 = __builtin_amdgcn_permlane32_swap(it, it, false, false);
  return __int_as_float(r2[0]) + __int_as_float(r2[1]);
}
template <int CTRL>
__device__ __forceinline__ float bd_dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float bd_row_sum16(float v) {   // sum over the 16 lanes of a row (the tile's windows), in every lane
  v = bd_dpp_add<0x128>(v);
  v = bd_dpp_add<0x124>(v);
  v = bd_dpp_add<0x122>(v);
  return bd_dpp_add<0x121>(v);
}
__device__ __forceinline__ void bd_wave_lds_sync() {   // LDS written by some lanes of this wave, read by others
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// bond index <-> (matrix row, k group, k step) of the packs for NS state values per lane (see bd_bwd_body)
template <int NS> __device__ __forceinline__ int bd_row(int r) { return NS == 4 ? r : 4 * (r >> 1) + (r & 1); }
template <int NS> __device__ __forceinline__ int bd_kg(int l) { return NS == 4 ? l >> 2 : l >> 1; }
template <int NS> __device__ __forceinline__ int bd_ks(int l) { return NS == 4 ? l & 3 : l & 1; }
// NS consecutive floats at a (NS * 4)-byte aligned LDS address
template <int NS> __device__ __forceinline__ void bd_store_state(float* dst, const float (&v)[NS]) {
  if constexpr (NS == 4) *reinterpret_cast<bd_f4*>(dst) = bd_f4{v[0], v[1], v[2], v[3]};
  else *reinterpret_cast<bd_f2*>(dst) = bd_f2{v[0], v[1]};
}

__device__ __forceinline__ void bd_barrier() {   // workgroup barrier that orders LDS only: vector-memory loads stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// QT: feature values per core (q^C, 2..4) = 16x16 tiles per product.  CH: 1 = one channel (the products ARE the pixel's
// values), 2 = two channels of two values (QT = 4; the deeper layers of the reference's classifier).
// NS: state values per lane and core.  The state layout is l = NS g + s (lane group g, register s); 4 covers bonds up
// to 16 with four k-steps per product, 2 covers bonds up to 8 with TWO (the packs put core row r' = 2 g + s at matrix
// row 4 g + s, so the accumulator's registers 0, 1 of lane group g are the new state): half the chain's matrix
// instructions and half its epilogues; the hand-over tiles, the gradient waves and the records keep their layouts
// (natural order [window][bond index]; entries 8..15 stay zero).
template <int QT, int CH, int NS>
__device__ __forceinline__ void bd_bwd_body(const BdP& p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wl = lane & 15, g = lane >> 4;
  constexpr int RK = CH == 2 ? 4 : QT;   // gradient values per core and window in the band's LDS rows
  constexpr int BD_CPT = BD_NPK * QT * 256;   // the gradient tiles of a workgroup, in accumulator layout (one copy of the join area)
  // the LDS plan as compile-time offsets (floats; bd_plan lays out the same): 16-byte accesses need provable alignment
  constexpr int PACKF = 0, PACKA = BD_CPT, FIRST = 2 * BD_CPT, LAST = FIRST + 64, FSO = LAST + 64;
  constexpr int RAWO = FSO + 4 * BD_NFS * BD_FS, GVO = RAWO + (CH == 2 ? 4 * BD_NFS * BD_FS : 0), ROWSO = GVO + 4 * 4 * BD_TILE;
  const int img = (int)blockIdx.x / p.nb, kb = (int)blockIdx.x - img * p.nb;
  const int r0 = kb * p.band_rows, r1 = r0 + p.band_rows < p.Ho ? r0 + p.band_rows : p.Ho;
  const int nwin = (r1 - r0) * p.Wo;
  BD_STAMP(0);
  if constexpr (NS != 4) {   // the chain waves write entries 0 .. 4 NS - 1 of a hand-over row: the rest stays zero
    for (int e = tid; e < 4 * 4 * BD_TILE; e += BD_THREADS) lds[GVO + e] = 0.f;
  }

  // ---- the packs.  Both are permutations of the middle cores' 6-8 K elements: every thread takes elements of the cores'
  // own (contiguous) layout - coalesced loads, all in flight before the first LDS store - and scatters each to its place
  // in the forward and in the adjoint pack (the gather form, 2 x 24 four-byte loads at computed addresses per thread, took
  // 21 k cycles of a 273 k-cycle kernel).  The packs are zero beyond a core's own bonds: zero-filled first.
  {
    constexpr int CHUNKS = (2 * BD_NPK * QT * 256) / 4 / BD_THREADS;   // 16-byte chunks per thread (the packs are contiguous)
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j)
      *reinterpret_cast<bd_f4*>(lds + PACKF + (tid + BD_THREADS * j) * 4) = bd_f4{0.f, 0.f, 0.f, 0.f};
    float va[BD_NC - 2][3];   // a core has at most 2 * 16 * 16 * 4 = 2048 elements: three rounds of 512 threads (q^C <= 3) or four
    float vb[BD_NC - 2];      // (fourth round: only two-valued cores with four feature values)
#pragma unroll
    for (int c = 1; c < BD_NC - 1; ++c) {
      const int cnt = c + 1 < p.n ? p.o[c] * p.bl[c] * p.br[c] * QT : 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) va[c - 1][j] = tid + BD_THREADS * j < cnt ? p.core[c][tid + BD_THREADS * j] : 0.f;
      vb[c - 1] = (QT == 4 && tid + BD_THREADS * 3 < cnt) ? p.core[c][tid + BD_THREADS * 3] : 0.f;
    }
    bd_barrier();   // the zero fill is complete
#pragma unroll
    for (int c = 1; c < BD_NC - 1; ++c) {
      if (c + 1 < p.n) {
        const int bl = p.bl[c], br = p.br[c];
        const int cnt = p.o[c] * bl * br * QT;
        const float ibr = 1.0f / (float)br, ibl = 1.0f / (float)bl;
#pragma unroll
        for (int j = 0; j < (QT == 4 ? 4 : 3); ++j) {
          const int e = tid + BD_THREADS * j;
          if (e < cnt) {
            const int qq = e % QT, t = e / QT;                      // element (o, l, r, qq) of the core's [o][l][r][qq] layout
            const int t1 = (int)(((float)t + 0.5f) * ibr);           // t / br  (exact: t < 2048)
            const int r = t - t1 * br;
            const int o = (int)(((float)t1 + 0.5f) * ibl);           // t1 / bl
            const int l = t1 - o * bl;
            const int pk = o == 0 ? c - 1 : BD_NPK - 1;
            const float val = j < 3 ? va[c - 1][j < 3 ? j : 0] : vb[c - 1];
            lds[PACKF + ((pk * QT + qq) * 64 + bd_row<NS>(r) + 16 * bd_kg<NS>(l)) * 4 + bd_ks<NS>(l)] = val;   // A[row r'][k: l = NS g + s]
            lds[PACKA + ((pk * QT + qq) * 64 + bd_row<NS>(l) + 16 * bd_kg<NS>(r)) * 4 + bd_ks<NS>(r)] = val;   // A[row l][k: r' = NS g + s]
          }
        }
      }
    }
    // first core (1, 1, r', qc) as [r'][4]; last core (1, l, 1, qc) as [l][4]; zero beyond the real extents
    if (tid < 64) {
      const int rr = tid >> 2, qq = tid & 3;
      lds[FIRST + tid] = (qq < p.qc && rr < p.br[0]) ? p.core[0][rr * p.qc + qq] : 0.f;
      lds[LAST + tid] = (qq < p.qc && rr < p.bl[p.n - 1]) ? p.core[p.n - 1][rr * p.qc + qq] : 0.f;
    }
  }
  bd_barrier();
  BD_STAMP(1);

  // ---- dX of the band (run behind barrier (A)): every pixel value sums the windows of this band that cover it, in core
  // order; rows shared with the band above / below go to the side buffer as this band's partial sum
  auto write_dx = [&](int t0, int nt) {
    if (p.dX == nullptr) return;
    const float* rows = lds + ROWSO;
    const int Cq = p.C * p.q;
    const int y0 = r0, y1 = (r1 + p.max_h < p.H) ? r1 + p.max_h : p.H;
    const int npix = (y1 - y0) * p.W * p.C;
    for (int e = t0; e < npix; e += nt) {
      const int ch = e / ((y1 - y0) * p.W), r2 = e - ch * (y1 - y0) * p.W;
      const int yr = r2 / p.W, xc = r2 - yr * p.W;
      const int y = y0 + yr;
      float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < BD_NC; ++c) {
        if (c < p.n) {
          const int ho = y - p.ph[c], wo = xc - p.pw[c];
          if (ho >= r0 && ho < r1 && wo >= 0 && wo < p.Wo) {
            const float* d = rows + ((size_t)((ho - r0) * p.Wo + wo) * p.n + c) * RK + (CH == 2 ? ch * 2 : 0);
#pragma unroll
            for (int k = 0; k < (CH == 2 ? 2 : QT); ++k) a[k] += d[k];
          }
        }
      }
      const bool with_above = kb > 0 && y < r0 + p.max_h;
      const bool with_below = kb + 1 < p.nb && y >= r1;
      float* dst;
      if (with_above || with_below) {
        const int bnd = with_above ? kb - 1 : kb, part = with_above ? 1 : 0;
        const int yb = y - (bnd + 1) * p.band_rows;
        dst = p.side + ((((size_t)img * (p.nb - 1) + bnd) * 2 + part) * p.max_h + yb) * p.W * Cq + (size_t)xc * Cq + ch * p.q;
      } else {
        dst = p.dX + ((((size_t)ch * p.B + img) * p.H + y) * p.W + xc) * p.q;
      }
#pragma unroll
      for (int k = 0; k < (CH == 2 ? 2 : QT); ++k) dst[k] = a[k];
    }
  };

  if (wv < 4) {
    // =============================================================================== chain wave
    // (s_setprio for either role changed nothing: 110.4 / 110.5 / 114.8 us for priorities chain:gradient 0:0 / 2:0 / 0:2)
    const float* fsb = lds + FSO + wv * BD_NFS * BD_FS;     // feature products (and dY) of three tiles: [tile % 3][row][window][4],
    const float* rawb = lds + RAWO + wv * BD_NFS * BD_FS;   // CH == 2: the pixels' raw values - both staged by the gradient wave
    float* gvb = lds + GVO + wv * 4 * BD_TILE;              // two buffers of (G tile, v tile)
    float* rows = lds + ROWSO;
    int gs = 0;

    // U_qq = A_qq x vin (K = l or r'): the whole A operand of a pack is QT b128 reads, all in flight before the first MFMA
    // waits for a part of it; k-step outermost so that the QT accumulators form independent chains
    auto product = [&](int off, int pk, const float (&vin)[NS], bd_f4 (&D)[QT]) {
      bd_f4 a[QT];
#pragma unroll
      for (int qq = 0; qq < QT; ++qq) {
        a[qq] = *reinterpret_cast<const bd_f4*>(lds + off + ((pk * QT + qq) * 64 + lane) * 4);
        D[qq] = bd_f4{0.f, 0.f, 0.f, 0.f};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) D[qq] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[qq][s], vin[s], D[qq], 0, 0, 0);
    };
    auto fwd_epi = [&](const bd_f4 (&D)[QT], const bd_f4& f, float (&out)[NS]) {   // v'[r'] = sum_qq f[qq] U_qq[r']
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        float a = f[0] * D[0][r] + f[1] * D[1][r];
        if (QT > 2) a += f[2] * D[2 % QT][r];
        if (QT > 3) a += f[3] * D[3 % QT][r];
        out[r] = a;
      }
    };
    auto bwd_epi = [&](const bd_f4 (&Wt)[QT], const bd_f4& f, const float (&vin)[NS], float (&dv)[NS], float (&df)[QT]) {
#pragma unroll
      for (int r = 0; r < NS; ++r) {   // dv[l] = sum_qq f[qq] W_qq[l];  df[qq] += sum_l v[l] W_qq[l]
        float a = f[0] * Wt[0][r] + f[1] * Wt[1][r];
        if (QT > 2) a += f[2] * Wt[2 % QT][r];
        if (QT > 3) a += f[3] * Wt[3 % QT][r];
        dv[r] = a;
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) df[qq] += vin[r] * Wt[qq][r];
      }
    };
    // G and v of the tile for the gradient wave (rows = windows, buffer gs & 1); meet(): the hand-over's barrier
    auto store_tiles = [&](const float (&Gs)[NS], const float (&vin)[NS]) {
      float* gt = gvb + (gs & 1) * 2 * BD_TILE;
      bd_store_state<NS>(gt + wl * BD_TS + NS * g, Gs);              // G[w][r' = NS g ..]
      bd_store_state<NS>(gt + BD_TILE + wl * BD_TS + NS * g, vin);   // v[w][l = NS g ..]
    };
    auto meet = [&]() {
      bd_barrier();
      ++gs;
    };
    auto handover_g = [&](const float (&Gs)[NS]) {   // the first / last core's gradient needs no second operand
      float* gt = gvb + (gs & 1) * 2 * BD_TILE;
      bd_store_state<NS>(gt + wl * BD_TS + NS * g, Gs);
      meet();
    };
    auto first_core = [&](const float* fs, float (&w0)[NS]) {   // v[r'] = sum_qq core0[r'][qq] f[qq]
      const bd_f4 f = *reinterpret_cast<const bd_f4*>(fs + (0 * 16 + wl) * 4);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float a = 0.f;
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) a += lds[FIRST + (NS * g + s) * 4 + qq] * f[qq];   // (4-byte reads: 16 lanes read one entry)
        w0[s] = a;
      }
    };
    // forward through middle core c: (w0, w1) -> (w0, w1)
    auto fwd_core = [&](int c, const float* fs, float (&w0)[NS], float (&w1)[NS]) {
      const bd_f4 f = *reinterpret_cast<const bd_f4*>(fs + (c * 16 + wl) * 4);
      float n0[NS], n1[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) n1[s] = 0.f;
      bd_f4 D[QT];
      product(PACKF, c - 1, w0, D);
      fwd_epi(D, f, n0);
      if (p.o[c] > 1 || p.nin[c] > 1) {
        const bool second_out = p.o[c] > 1;   // (o = 1 from state 0) or (o = 0 from state 1): wave-uniform
        float vin[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) vin[s] = second_out ? w0[s] : w1[s];
        product(PACKF, second_out ? BD_NPK - 1 : c - 1, vin, D);
        fwd_epi(D, f, n1);
      }
#pragma unroll
      for (int r = 0; r < NS; ++r) { w0[r] = n0[r]; w1[r] = n1[r]; }
    };

    float Sin0[BD_NC][NS], Sin1[BD_NC][NS];   // input states of the middle cores of the tile on its way back (slot c)
    float v0[NS], v1[NS];                     // input states of its last core
    // ---- barrier (P): the gradient wave has staged the first tile's features
    bd_barrier();
    auto forward_sweep = [&](const float* fs) {
      first_core(fs, v0);
#pragma unroll
      for (int s = 0; s < NS; ++s) v1[s] = 0.f;
#pragma unroll
      for (int c = 1; c < BD_NC - 1; ++c) {
        if (c + 1 < p.n) {
#pragma unroll
          for (int s = 0; s < NS; ++s) { Sin0[c][s] = v0[s]; Sin1[c][s] = v1[s]; }
          fwd_core(c, fs, v0, v1);
        }
      }
    };
    bd_barrier();   // (P2): the second tile's features are staged

    for (int it = 0; it < p.iters; ++it) {
      const float* fs = fsb + (it % BD_NFS) * BD_FS;
      if (it == 1) BD_STAMP(2);
      const float* raw = rawb + (it % BD_NFS) * BD_FS;
      (void)raw;
      forward_sweep(fs);
      if (it == 1) BD_STAMP(3);
      const float dy0 = fs[(BD_NC * 16 + wl) * 4], dy1 = fs[(BD_NC * 16 + wl) * 4 + 1];
      const int wb = (it * 4 + wv) * 16 + wl;
      const bool valid = wb < nwin;
      if (it == 1) BD_STAMP(4);

      // per-window feature gradient of core c -> the band's LDS row (one lane per window: the k groups hold the same sum)
      auto put_row = [&](int c, const float (&df)[QT]) {
        if (p.dX == nullptr) return;
        float t[QT];
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) t[qq] = bd_group_sum(df[qq]);
        if (g == 0 && valid) {
          float* dst = rows + ((size_t)wb * p.n + c) * RK;
          if constexpr (CH == 2) {   // d/d(products) -> d/d(pixel values): raw = (x0[0], x0[1], x1[0], x1[1])
            const bd_f4 xr = *reinterpret_cast<const bd_f4*>(raw + (c * 16 + wl) * 4);
            dst[0] = t[0] * xr[2] + t[1] * xr[3];   // channel 0, value 0
            dst[1] = t[2] * xr[2] + t[3] * xr[3];   // channel 0, value 1
            dst[2] = t[0] * xr[0] + t[2] * xr[1];   // channel 1, value 0
            dst[3] = t[1] * xr[0] + t[3] * xr[1];   // channel 1, value 1
          } else {
#pragma unroll
            for (int qq = 0; qq < QT; ++qq) dst[qq] = t[qq];
          }
        }
      };

      // ---------------------------------------------------------------- way back
      float G0[NS], G1[NS];
      {   // last core: out[a] = sum_l v_a[l] tl[l],  tl[l] = sum_qq coreL[l][qq] f[qq]
        const bd_f4 f = *reinterpret_cast<const bd_f4*>(fs + ((p.n - 1) * 16 + wl) * 4);
        float df[QT], u[NS];
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) df[qq] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          float cp[QT], tl = 0.f;
#pragma unroll
          for (int qq = 0; qq < QT; ++qq) {
            cp[qq] = lds[LAST + (NS * g + s) * 4 + qq];
            tl += cp[qq] * f[qq];
          }
          G0[s] = dy0 * tl;
          G1[s] = dy1 * tl;
          u[s] = dy0 * v0[s] + dy1 * v1[s];
#pragma unroll
          for (int qq = 0; qq < QT; ++qq) df[qq] += u[s] * cp[qq];
        }
        put_row(p.n - 1, df);
        handover_g(u);   // dCoreL[l][qq] = sum_w u[l, w] f[qq, w]: four MFMAs of the gradient wave
      }
      if (it == 1) BD_STAMP(5);
      // One (input state, output value) pair of a middle core.  What the gradient wave needs - the adjoint G the pair starts
      // from and the input state v - is known BEFORE the pair's product: it is stored first, the product's MFMAs are issued,
      // and only then the workgroup meets at the barrier - the wait runs while the matrix pipe works on this wave's product,
      // the gradient wave's MFMAs take the pipe during this wave's epilogue.  (Barrier behind the epilogue: 1 250 cycles a pair.)
#pragma unroll
      for (int c = BD_NC - 2; c >= 1; --c) {
        if (c + 1 < p.n) {
          const bd_f4 f = *reinterpret_cast<const bd_f4*>(fs + (c * 16 + wl) * 4);
          float d0[NS], d1[NS], df[QT];
#pragma unroll
          for (int s = 0; s < NS; ++s) d1[s] = 0.f;
#pragma unroll
          for (int qq = 0; qq < QT; ++qq) df[qq] = 0.f;
          bd_f4 Wt[QT];
          if (it == 1 && c == 3) BD_STAMP(19);
          store_tiles(G0, Sin0[c]);
          product(PACKA, c - 1, G0, Wt);
          if (it == 1 && c == 3) BD_STAMP(20);
          meet();
          if (it == 1 && c == 3) BD_STAMP(21);
          bwd_epi(Wt, f, Sin0[c], d0, df);
          if (it == 1 && c == 3) BD_STAMP(22);
          if (p.o[c] > 1 || p.nin[c] > 1) {   // a second pair: (o = 1, state 0) or (o = 0, state 1)
            const bool b_out = p.o[c] > 1;
            float vin2[NS], dt[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) vin2[s] = b_out ? Sin0[c][s] : Sin1[c][s];
            store_tiles(G1, vin2);
            product(PACKA, b_out ? BD_NPK - 1 : c - 1, G1, Wt);
            meet();
            bwd_epi(Wt, f, vin2, dt, df);
#pragma unroll
            for (int r = 0; r < NS; ++r) {
              d0[r] += b_out ? dt[r] : 0.f;
              d1[r] = b_out ? 0.f : dt[r];
            }
          }
          put_row(c, df);
          if (it == 1 && c == 3) BD_STAMP(23);
#pragma unroll
          for (int s = 0; s < NS; ++s) { G0[s] = d0[s]; G1[s] = d1[s]; }
          if (it == 1) BD_STAMP(5 + (BD_NC - 1 - c));   // 6 (core 7) .. 12 (core 1)
        }
      }
      {   // first core: v[r'] = sum_qq core0[r'][qq] f[qq]
        float df[QT];
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) df[qq] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
          for (int qq = 0; qq < QT; ++qq) df[qq] += G0[s] * lds[FIRST + (NS * g + s) * 4 + qq];
        }
        put_row(0, df);
        handover_g(G0);   // dCore0[r'][qq] = sum_w G0[r', w] f[qq, w]
      }
      if (it == 1) BD_STAMP(13);
    }
    BD_STAMP(14);

    // ---- barrier (A): every chain step is done and the gradient waves have consumed the last hand-over.  The chain waves
    // then write the band's dX (they hold nothing else; the gradient waves meanwhile join their accumulators in the part of
    // the LDS in front of the band's rows), and behind (B1) leave the partial sums of their first / last core gradients
    bd_barrier();   // (A)
    BD_STAMP(15);
    write_dx(tid, BD_THREADS / 2);
    bd_barrier();   // (B1)
    BD_STAMP(16);
  } else {
    // =============================================================================== gradient wave
    const int wp = wv - 4;
    float* fsb = lds + FSO + wp * BD_NFS * BD_FS;
    float* rawb = lds + RAWO + wp * BD_NFS * BD_FS;
    const float* gvb = lds + GVO + wp * 4 * BD_TILE;
    (void)rawb;
    // ---- it also stages its chain wave's tiles, two tiles ahead: the pixels of cores g, g + 4, g + 8 of the lane's window
    // (lane group 1, which has no third core: the window's dY) travel during one tile and go to LDS as feature products
    float pre[3][4];
    // (the lane's window advances by 64 per tile: its (row, column) and the pixel addresses are carried along instead of
    // being divided out of the window index every tile - the wave yields the issue port to its chain wave)
    int ld_wb = wp * 16 + wl, ld_hr = ld_wb / p.Wo, ld_wo = ld_wb - ld_hr * p.Wo;
    long long poff[3];   // element offsets of the lane's three pixels inside a window
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = g + 4 * k, cc = c < p.n ? c : 0;
      poff[k] = (long long)p.ph[cc] * p.xs[2] + (long long)p.pw[cc] * p.xs[3];
    }
    const float* ximg = p.x + (long long)img * p.xs[1] + (long long)r0 * p.xs[2];
    const float* dyimg = p.dY + ((long long)img * p.Ho + r0) * p.Wo * p.Otot;
    // Loads are UNCONDITIONAL (a lane without a window reads window 0 of the band) and nothing is done with their values
    // here: a select on a just-loaded value makes the wave wait for the load on the spot (the staging then took 2 600
    // cycles of every tile, at a barrier its chain wave waits at); validity is applied when the values go to LDS.
    auto issue_loads = [&](int it) {   // tiles in order: it = 0, 1, 2, ...
      (void)it;
      const bool valid = ld_wb < nwin;
      const float* win = ximg + (valid ? (long long)ld_hr * p.xs[2] + (long long)ld_wo * p.xs[3] : 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float* px = win + poff[k];
        if constexpr (CH == 2) {
          pre[k][0] = px[0];
          pre[k][1] = px[p.xs[4]];
          pre[k][2] = px[p.xs[0]];
          pre[k][3] = px[p.xs[0] + p.xs[4]];
        } else {
#pragma unroll
          for (int d = 0; d < 4; ++d) pre[k][d] = px[(d < QT ? d : 0) * p.xs[4]];
        }
      }
      if (g == 1) {   // (cores 1 and 5 only: the third slot carries dY)
        const float* dyp = dyimg + (long long)(valid ? ld_wb : 0) * p.Otot;
        pre[2][0] = dyp[0];
        pre[2][1] = dyp[p.Otot > 1 ? 1 : 0];
      }
      ld_wb += 64;
      ld_wo += 64;
      while (ld_wo >= p.Wo) { ld_wo -= p.Wo; ++ld_hr; }
    };
    auto commit = [&](int it) {
      float* fs = fsb + (it % BD_NFS) * BD_FS;
      const bool valid = (it * 4 + wp) * 16 + wl < nwin;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = g + 4 * k;
        if (c < p.n && !(k == 2 && g == 1)) {
          float v[4];
#pragma unroll
          for (int d = 0; d < 4; ++d) v[d] = (valid && (CH == 2 || d < QT)) ? pre[k][d] : 0.f;
          if constexpr (CH == 2) {   // f[2 d + e] = x0[d] x1[e]  (channel 0 most significant); the raw values beside them
            *reinterpret_cast<bd_f4*>(fs + (c * 16 + wl) * 4) = bd_f4{v[0] * v[2], v[0] * v[3], v[1] * v[2], v[1] * v[3]};
            *reinterpret_cast<bd_f4*>(rawb + (it % BD_NFS) * BD_FS + (c * 16 + wl) * 4) = bd_f4{v[0], v[1], v[2], v[3]};
          } else {
            *reinterpret_cast<bd_f4*>(fs + (c * 16 + wl) * 4) = bd_f4{v[0], v[1], v[2], v[3]};
          }
        }
      }
      if (g == 1)
        *reinterpret_cast<bd_f4*>(fs + (BD_NC * 16 + wl) * 4) =
            bd_f4{valid ? pre[2][0] : 0.f, (valid && p.Otot > 1) ? pre[2][1] : 0.f, 0.f, 0.f};
    };
    issue_loads(0);
    commit(0);
    if (p.iters > 1) issue_loads(1);
    bd_barrier();   // (P)
    if (p.iters > 1) commit(1);
    if (p.iters > 2) issue_loads(2);
    bd_barrier();   // (P2)

    bd_f4 acc[BD_NPK][QT], accL = bd_f4{0.f, 0.f, 0.f, 0.f}, accF = bd_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < BD_NPK; ++i)
#pragma unroll
      for (int qq = 0; qq < QT; ++qq) acc[i][qq] = bd_f4{0.f, 0.f, 0.f, 0.f};
    int gs = 0;
    // The wave runs ONE hand-over behind: behind barrier k it fetches the operands of hand-over k (LDS -> registers) and,
    // while those reads fly, multiplies hand-over k - 1 from the registers it fetched a step ago.
    // Middle cores: T_qq[r'][l] += sum_w G[r', w] f_qq[w] v[l, w]; k-step ks of lane group g is window 4 g + ks.
    // First / last core: T[r'][qq] += sum_w G[r', w] f[qq, w] - ONE tile, its columns the feature values (lanes wl < 4).
    float ga[4], vb[4], fw[4][QT], b4[4];   // the pending hand-over
    bool have = false, pend7 = false;       // (wave-uniform) one is pending; it belongs to the two-valued core's second output
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ga[ks] = 0.f; vb[ks] = 0.f; b4[ks] = 0.f;
#pragma unroll
      for (int qq = 0; qq < QT; ++qq) fw[ks][qq] = 0.f;
    }
    auto flush_mid = [&](bd_f4 (&A)[QT]) {
      float bq[4][QT];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) bq[ks][qq] = vb[ks] * fw[ks][qq];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) A[qq] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks], bq[ks][qq], A[qq], 0, 0, 0);
    };
    auto flush_fl = [&](bd_f4& A) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) A = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks], b4[ks], A, 0, 0, 0);
    };
    // fetch hand-over gs (behind its barrier) into the NEW registers; mid: G, v and the core's features; fl: G and the features as columns
    auto fetch_mid = [&](int c, const float* fs, float (&ga2)[4], float (&vb2)[4], float (&fw2)[4][QT]) {
      bd_barrier();
      const float* gt = gvb + (gs & 1) * 2 * BD_TILE;
      ++gs;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        ga2[ks] = gt[(4 * g + ks) * BD_TS + wl];              // G[w][r' = wl]
        vb2[ks] = gt[BD_TILE + (4 * g + ks) * BD_TS + wl];    // v[w][l = wl]
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) fw2[ks][qq] = fs[(c * 16 + 4 * g + ks) * 4 + qq];   // (4-byte reads broadcast)
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto fetch_fl = [&](int c, const float* fs, float (&ga2)[4], float (&b2)[4]) {
      bd_barrier();
      const float* gt = gvb + (gs & 1) * 2 * BD_TILE;
      ++gs;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        ga2[ks] = gt[(4 * g + ks) * BD_TS + wl];
        const float fv = fs[(c * 16 + 4 * g + ks) * 4 + (wl & 3)];
        b2[ks] = wl < 4 ? fv : 0.f;                            // B[k = window][column = feature value wl]
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int it = 0; it < p.iters; ++it) {
      const float* fs = fsb + (it % BD_NFS) * BD_FS;
      if (it == 1) BD_STAMP(2);
      {   // ---- the last core's hand-over; pending: the first core's of the previous tile
        float ga2[4], b2[4];
        fetch_fl(p.n - 1, fs, ga2, b2);
        if (have) flush_fl(accF);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { ga[ks] = ga2[ks]; b4[ks] = b2[ks]; }
        have = true;
      }
#pragma unroll
      for (int c = BD_NC - 2; c >= 1; --c) {
        if (c + 1 < p.n) {
          {   // ---- first pair of core c; pending: the last core's hand-over (top core) or the last pair of core c + 1
            float ga2[4], vb2[4], fw2[4][QT];
            if (it == 1 && c == 3) BD_STAMP(19);
            fetch_mid(c, fs, ga2, vb2, fw2);
            if (it == 1 && c == 3) BD_STAMP(20);
            if (c + 2 == p.n) flush_fl(accL);
            else if (pend7) flush_mid(acc[BD_NPK - 1]);
            else flush_mid(acc[c < BD_NC - 2 ? c : 0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              ga[ks] = ga2[ks]; vb[ks] = vb2[ks];
#pragma unroll
              for (int qq = 0; qq < QT; ++qq) fw[ks][qq] = fw2[ks][qq];
            }
            pend7 = false;
            if (it == 1 && c == 3) BD_STAMP(21);
          }
          if (p.o[c] > 1 || p.nin[c] > 1) {   // ---- second pair of core c; pending: its first pair (slot c - 1)
            float ga2[4], vb2[4], fw2[4][QT];
            fetch_mid(c, fs, ga2, vb2, fw2);
            flush_mid(acc[c - 1]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              ga[ks] = ga2[ks]; vb[ks] = vb2[ks];
#pragma unroll
              for (int qq = 0; qq < QT; ++qq) fw[ks][qq] = fw2[ks][qq];
            }
            pend7 = p.o[c] > 1;
          }
          if (c == 4) {   // the tile after next: its pixels have arrived during this tile; the one after that sets out
            if (it + 2 < p.iters) commit(it + 2);
            if (it + 3 < p.iters) issue_loads(it + 3);
          }
          if (it == 1) BD_STAMP(5 + (BD_NC - 1 - c));
        }
      }
      {   // ---- the first core's hand-over; pending: the last pair of core 1 (slot 0, or 7)
        float ga2[4], b2[4];
        fetch_fl(0, fs, ga2, b2);
        if (pend7) flush_mid(acc[BD_NPK - 1]); else flush_mid(acc[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { ga[ks] = ga2[ks]; b4[ks] = b2[ks]; }
        pend7 = false;
      }
    }
    if (have) flush_fl(accF);   // the last tile's first-core hand-over
    BD_STAMP(14);
    bd_barrier();   // (A)
    BD_STAMP(15);
    // ---- the four waves' tiles join in two copies, in accumulator layout [slot][qq][register][lane]: waves 0, 1 store,
    // behind (B1) waves 2, 3 add theirs (fixed pairing: bit-reproducible); the first / last core tiles (columns = feature
    // values, lanes wl < 4) go to the wave's own [16][4] tables
    {
      float* cp = lds + (wp & 1) * BD_CPT + lane;
      if (wp < 2) {
#pragma unroll
        for (int i = 0; i < BD_NPK; ++i)
#pragma unroll
          for (int qq = 0; qq < QT; ++qq)
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[((i * QT + qq) * 4 + r) * 64] = acc[i][qq][r];
      }
      if (wl < 4) {
        float* fl = lds + 2 * BD_CPT + wp * 128;   // first [16][4], last [16][4] of this wave
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          fl[(4 * g + r) * 4 + wl] = accF[r];
          fl[64 + (4 * g + r) * 4 + wl] = accL[r];
        }
      }
      bd_barrier();   // (B1)
      BD_STAMP(16);
      if (wp >= 2) {
#pragma unroll
        for (int i = 0; i < BD_NPK; ++i)
#pragma unroll
          for (int qq = 0; qq < QT; ++qq)
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[((i * QT + qq) * 4 + r) * 64] += acc[i][qq][r];
      }
    }
  }
  bd_barrier();   // (D)
  BD_STAMP(17);
  // ---- the workgroup's record = the sum of the two copies (and of the four first / last partial sums), in the join area's
  // own layout (the tail kernel, which reads every record anyway, puts the entries into the cores' layouts)
  if (p.records != nullptr) {
    bd_f4* rec = reinterpret_cast<bd_f4*>(p.records + (size_t)blockIdx.x * (BD_CPT + 128));
    for (int e = tid; e < BD_CPT / 4; e += BD_THREADS)
      rec[e] = *reinterpret_cast<const bd_f4*>(lds + e * 4) + *reinterpret_cast<const bd_f4*>(lds + BD_CPT + e * 4);
    if (tid < 32) {
      const float* fl = lds + 2 * BD_CPT + tid * 4;
      rec[BD_CPT / 4 + tid] = (*reinterpret_cast<const bd_f4*>(fl) + *reinterpret_cast<const bd_f4*>(fl + 128)) +
                              (*reinterpret_cast<const bd_f4*>(fl + 256) + *reinterpret_cast<const bd_f4*>(fl + 384));
    }
  }
  BD_STAMP(18);
}

template <int QT, int CH, int NS>
__global__ __launch_bounds__(BD_THREADS, 2) void convsbs_bwd_band_k(const BdP p) { bd_bwd_body<QT, CH, NS>(p); }

// The strings of one ManyConvSBS layer (dctn/conv_sbs.py:367-370) in one launch: blockIdx.y = string.  String s > 0 writes
// its share of dX into a buffer of its own; the tail kernel adds the strings' shares (and every string's shared rows).
constexpr int BD_MANY = 2;
struct BdPMany { BdP s[BD_MANY]; };
template <int QT, int CH, int NS>
__global__ __launch_bounds__(BD_THREADS, 2) void convsbs_bwd_band_many_k(const BdPMany pp) { bd_bwd_body<QT, CH, NS>(pp.s[blockIdx.y]); }

// ------------------------------------------------------------------------------------------------ forward
// The same chain, forwards only: no accumulators, no states to keep - 8 identical waves per workgroup, each with its own
// 16-window tiles (flat over all windows of the batch), at most 128 registers so that FOUR waves share a SIMD and cover
// each other's dependent product -> epilogue -> product chain (the round-2 forward, convsbs_fwd_mfma_k: two waves per
// SIMD, rows padded to four feature values, one LDS round trip per k-step: 41 us at the cfg4 shape, matrix pipe 29 % busy).
// The next tile's pixels travel while the current one is swept.
struct BdFwdP {
  const float* x;
  float* out;
  const float* core[BD_NC];
  long long xs[5];
  int n, C, q, qc, B, H, W, Ho, Wo, Otot;
  int o[BD_NC], bl[BD_NC], br[BD_NC], ph[BD_NC], pw[BD_NC], nin[BD_NC];
  long long Wn, ntiles;
};

template <int QT, int CH, int NS>
__device__ __forceinline__ void bd_fwd_body(const BdFwdP& p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wl = lane & 15, g = lane >> 4;
  constexpr int CPT = BD_NPK * QT * 256;
  constexpr int PACKF = 0, FIRST = CPT, LAST = FIRST + 64, FSO = LAST + 64;   // then per wave two feature buffers [core][window][4]
  constexpr int FS = BD_NC * 16 * 4;
  // ---- the forward pack (as in the backward: coalesced loads of the cores' own layouts, scattered into fragment order)
  {
    constexpr int CHUNKS = CPT / 4 / BD_THREADS;
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) *reinterpret_cast<bd_f4*>(lds + PACKF + (tid + BD_THREADS * j) * 4) = bd_f4{0.f, 0.f, 0.f, 0.f};
    float va[BD_NC - 2][4];
#pragma unroll
    for (int c = 1; c < BD_NC - 1; ++c) {
      const int cnt = c + 1 < p.n ? p.o[c] * p.bl[c] * p.br[c] * QT : 0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        va[c - 1][j] = ((QT == 4 || j < 3) && tid + BD_THREADS * j < cnt) ? p.core[c][tid + BD_THREADS * j] : 0.f;
    }
    bd_barrier();
#pragma unroll
    for (int c = 1; c < BD_NC - 1; ++c) {
      if (c + 1 < p.n) {
        const int bl = p.bl[c], br = p.br[c];
        const int cnt = p.o[c] * bl * br * QT;
        const float ibr = 1.0f / (float)br, ibl = 1.0f / (float)bl;
#pragma unroll
        for (int j = 0; j < (QT == 4 ? 4 : 3); ++j) {
          const int e = tid + BD_THREADS * j;
          if (e < cnt) {
            const int qq = e % QT, t = e / QT;
            const int t1 = (int)(((float)t + 0.5f) * ibr);
            const int r = t - t1 * br;
            const int o = (int)(((float)t1 + 0.5f) * ibl);
            const int l = t1 - o * bl;
            const int pk = o == 0 ? c - 1 : BD_NPK - 1;
            lds[PACKF + ((pk * QT + qq) * 64 + bd_row<NS>(r) + 16 * bd_kg<NS>(l)) * 4 + bd_ks<NS>(l)] = va[c - 1][j];
          }
        }
      }
    }
    if (tid < 64) {
      const int rr = tid >> 2, qq = tid & 3;
      lds[FIRST + tid] = (qq < p.qc && rr < p.br[0]) ? p.core[0][rr * p.qc + qq] : 0.f;
      lds[LAST + tid] = (qq < p.qc && rr < p.bl[p.n - 1]) ? p.core[p.n - 1][rr * p.qc + qq] : 0.f;
    }
  }
  bd_barrier();

  float* fsb = lds + FSO + wv * 2 * FS;
  const long long wave = (long long)blockIdx.x * 8 + wv, nwaves = (long long)gridDim.x * 8;
  const int hw = p.Ho * p.Wo;
  // prefetch registers: the pixels of cores g, g + 4, g + 8 of the lane's window (unconditional loads; validity at commit)
  float pre[3][4];
  auto issue_loads = [&](long long tile) {
    const long long w = tile * 16 + wl;
    const long long ww = w < p.Wn ? w : 0;
    const long long b = ww / hw;
    const int rem = (int)(ww - b * hw);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const float* win = p.x + b * p.xs[1] + (long long)ho * p.xs[2] + (long long)wo * p.xs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = g + 4 * k, cc = c < p.n ? c : 0;
      const float* px = win + (long long)p.ph[cc] * p.xs[2] + (long long)p.pw[cc] * p.xs[3];
      if constexpr (CH == 2) {
        pre[k][0] = px[0];
        pre[k][1] = px[p.xs[4]];
        pre[k][2] = px[p.xs[0]];
        pre[k][3] = px[p.xs[0] + p.xs[4]];
      } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) pre[k][d] = px[(d < QT ? d : 0) * p.xs[4]];
      }
    }
  };
  auto commit = [&](long long tile, float* fs) {
    const bool valid = tile * 16 + wl < p.Wn;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = g + 4 * k;
      if (c < p.n) {
        float v[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] = (valid && (CH == 2 || d < QT)) ? pre[k][d] : 0.f;
        if constexpr (CH == 2) *reinterpret_cast<bd_f4*>(fs + (c * 16 + wl) * 4) = bd_f4{v[0] * v[2], v[0] * v[3], v[1] * v[2], v[1] * v[3]};
        else *reinterpret_cast<bd_f4*>(fs + (c * 16 + wl) * 4) = bd_f4{v[0], v[1], v[2], v[3]};
      }
    }
  };
  auto product = [&](int pk, const float (&vin)[NS], bd_f4 (&D)[QT]) {
    bd_f4 a[QT];
#pragma unroll
    for (int qq = 0; qq < QT; ++qq) {
      a[qq] = *reinterpret_cast<const bd_f4*>(lds + PACKF + ((pk * QT + qq) * 64 + lane) * 4);
      D[qq] = bd_f4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int qq = 0; qq < QT; ++qq) D[qq] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[qq][s], vin[s], D[qq], 0, 0, 0);
  };
  auto epi = [&](const bd_f4 (&D)[QT], const bd_f4& f, float (&out)[NS]) {
#pragma unroll
    for (int r = 0; r < NS; ++r) {
      float a = f[0] * D[0][r] + f[1] * D[1][r];
      if (QT > 2) a += f[2] * D[2 % QT][r];
      if (QT > 3) a += f[3] * D[3 % QT][r];
      out[r] = a;
    }
  };

  long long tile = wave;
  if (tile < p.ntiles) issue_loads(tile);
  int buf = 0;
  for (; tile < p.ntiles; tile += nwaves) {
    float* fs = fsb + buf * FS;
    buf ^= 1;
    commit(tile, fs);
    if (tile + nwaves < p.ntiles) issue_loads(tile + nwaves);
    bd_wave_lds_sync();
    float w0[NS], w1[NS];
    {
      const bd_f4 f = *reinterpret_cast<const bd_f4*>(fs + (0 * 16 + wl) * 4);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float a = 0.f;
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) a += lds[FIRST + (NS * g + s) * 4 + qq] * f[qq];
        w0[s] = a;
        w1[s] = 0.f;
      }
    }
#pragma unroll
    for (int c = 1; c < BD_NC - 1; ++c) {
      if (c + 1 < p.n) {
        const bd_f4 f = *reinterpret_cast<const bd_f4*>(fs + (c * 16 + wl) * 4);
        float n0[NS], n1[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) n1[s] = 0.f;
        bd_f4 D[QT];
        product(c - 1, w0, D);
        epi(D, f, n0);
        if (p.o[c] > 1 || p.nin[c] > 1) {
          const bool second_out = p.o[c] > 1;
          float vin[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) vin[s] = second_out ? w0[s] : w1[s];
          product(second_out ? BD_NPK - 1 : c - 1, vin, D);
          epi(D, f, n1);
        }
#pragma unroll
        for (int r = 0; r < NS; ++r) { w0[r] = n0[r]; w1[r] = n1[r]; }
      }
    }
    {   // last core: out[a] = sum_l v_a[l] sum_qq coreL[l][qq] f[qq]
      const bd_f4 f = *reinterpret_cast<const bd_f4*>(fs + ((p.n - 1) * 16 + wl) * 4);
      float r0 = 0.f, r1 = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float tl = 0.f;
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) tl += lds[LAST + (NS * g + s) * 4 + qq] * f[qq];
        r0 += w0[s] * tl;
        r1 += w1[s] * tl;
      }
      r0 = bd_group_sum(r0);
      r1 = bd_group_sum(r1);
      const long long w = tile * 16 + wl;
      if (g == 0 && w < p.Wn) {
        p.out[w * p.Otot] = r0;
        if (p.Otot > 1) p.out[w * p.Otot + 1] = r1;
      }
    }
  }
}

template <int QT, int CH, int NS>
__global__ __launch_bounds__(BD_THREADS, 4) void convsbs_fwd_band_k(const BdFwdP p) { bd_fwd_body<QT, CH, NS>(p); }

struct BdFwdPMany { BdFwdP s[BD_MANY]; };
template <int QT, int CH, int NS>
__global__ __launch_bounds__(BD_THREADS, 4) void convsbs_fwd_band_many_k(const BdFwdPMany pp) { bd_fwd_body<QT, CH, NS>(pp.s[blockIdx.y]); }

// dCore_c[e] = sum over the workgroups' records in a fixed order; the pixel rows two bands share = the sum of their two
// partial sums.  64 elements per workgroup, 4 record subsets, LDS join (as convsbs_dcore_reduce_k).
__device__ __forceinline__ void bd_tail_records(const BdTailP& p, int blk) {
  __shared__ float red[4][64];
  {
    // 64 consecutive POSITIONS of the records per workgroup (coalesced: a record is the join area as it stands - middle
    // cores as accumulator tiles [slot][qq][register = r' & 3][lane = l + 16 (r' >> 2)], then the first and the last core
    // [16][4]); the sum goes to the element of the cores' layouts that position stands for.  (Element-major reads
    // gathered one 4-byte word per 64-byte line: 51 MB of traffic for 6.4 MB of records.)
    const int pos = blk * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (pos < p.rec_len) {
      int r = sub;
      for (; r + 12 < p.nrec; r += 16) {
        a0 += p.records[(size_t)r * p.rec_len + pos];
        a1 += p.records[(size_t)(r + 4) * p.rec_len + pos];
        a2 += p.records[(size_t)(r + 8) * p.rec_len + pos];
        a3 += p.records[(size_t)(r + 12) * p.rec_len + pos];
      }
      for (; r < p.nrec; r += 4) a0 += p.records[(size_t)r * p.rec_len + pos];
    }
    red[sub][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sub == 0 && pos < p.rec_len) {
      const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
      const int cpt = BD_NPK * p.qc * 256;
      int c, o = 0, l, r, qq;
      if (pos >= cpt) {   // first core [r'][4] / last core [l][4]
        const int e = pos - cpt, last = e >= 64;
        c = last ? p.n - 1 : 0;
        const int rr = (e & 63) >> 2;
        qq = e & 3;
        l = last ? rr : 0;
        r = last ? 0 : rr;
      } else {
        const int ln = pos & 63, reg = (pos >> 6) & 3, rest = pos >> 8;
        qq = rest % p.qc;
        const int slot = rest / p.qc;
        c = slot < BD_NPK - 1 ? slot + 1 : p.c2;
        o = slot < BD_NPK - 1 ? 0 : 1;
        l = ln & 15;
        r = 4 * (ln >> 4) + reg;
        if (slot < BD_NPK - 1 && c + 1 >= p.n) c = -1;   // a slot of a core the string does not have
      }
      if (c >= 0 && c < p.n && p.dcore[c] != nullptr && qq < p.qc && l < p.bl[c] && r < p.br[c] && o < p.o[c])
        p.dcore[c][((o * p.bl[c] + l) * p.br[c] + r) * p.qc + qq] = v;
    }
  }
}

__global__ __launch_bounds__(256) void convsbs_band_tail_k(const BdTailP p) {
  const int nblk_core = p.records ? (p.rec_len + 63) / 64 : 0;
  if ((int)blockIdx.x < nblk_core) {
    bd_tail_records(p, (int)blockIdx.x);
    return;
  }
  // shared rows: element = (image, boundary, row, column, channel, value)
  const long long e = ((long long)blockIdx.x - nblk_core) * 256 + threadIdx.x;
  if (e >= p.nshared) return;
  const int rowlen = p.W * p.Cq;
  const long long per_b = (long long)p.max_h * rowlen;
  const long long bi = e / per_b;                 // image * (nb - 1) + boundary
  const int rem = (int)(e - bi * per_b);
  const int yb = rem / rowlen, r2 = rem - yb * rowlen;
  const int xc = r2 / p.Cq, k = r2 - xc * p.Cq;
  const int img = (int)(bi / (p.nb - 1)), bnd = (int)(bi - (long long)img * (p.nb - 1));
  const int y = (bnd + 1) * p.band_rows + yb;
  if (y >= p.H) return;
  const int ch = k / p.q, d = k - ch * p.q;
  const float* s0 = p.side + ((size_t)bi * 2 * p.max_h + yb) * rowlen + r2;
  const float v = s0[0] + s0[(size_t)p.max_h * rowlen];
  p.dX[((((size_t)ch * p.B + img) * p.H + y) * p.W + xc) * p.q + d] = v;
}

// Several strings: the records of every string, then ONE pass over all of dX - an element is the sum over the strings of
// the string's own value (its dX buffer) or, on a pixel row two bands share, of the two partial sums of its side buffer.
// (One thread per element and all strings: string 0's buffer IS dX, nobody else touches the element.)  The strings share
// the band geometry (checked by the host).
struct BdTailMany { BdTailP s[BD_MANY]; int ns; };
__global__ __launch_bounds__(256) void convsbs_band_tail_many_k(const BdTailMany pp) {
  const BdTailP& p0 = pp.s[0];
  const int nblk_core = p0.records ? (p0.rec_len + 63) / 64 : 0;
  if ((int)blockIdx.x < nblk_core * pp.ns) {
    const int sidx = (int)blockIdx.x / nblk_core;
    bd_tail_records(pp.s[sidx], (int)blockIdx.x - sidx * nblk_core);
    return;
  }
  if (!p0.dX) return;
  const long long e = ((long long)blockIdx.x - (long long)nblk_core * pp.ns) * 256 + threadIdx.x;
  const long long total = (long long)p0.C * p0.B * p0.H * p0.W * p0.q;
  if (e >= total) return;
  const int d = (int)(e % p0.q);
  long long rest = e / p0.q;
  const int xc = (int)(rest % p0.W); rest /= p0.W;
  const int y = (int)(rest % p0.H); rest /= p0.H;
  const int img = (int)(rest % p0.B);
  const int ch = (int)(rest / p0.B);
  const int kq = y / p0.band_rows, yb = y - kq * p0.band_rows;
  const bool shared = kq >= 1 && kq <= p0.nb - 1 && yb < p0.max_h;
  const int rowlen = p0.W * p0.Cq;
  float v = 0.f;
  for (int sidx = 0; sidx < pp.ns; ++sidx) {
    const BdTailP& p = pp.s[sidx];
    if (shared) {
      const long long bi = (long long)img * (p.nb - 1) + (kq - 1);
      const float* s0 = p.side + ((size_t)bi * 2 * p.max_h + yb) * rowlen + (xc * p.Cq + ch * p.q + d);
      v += s0[0] + s0[(size_t)p.max_h * rowlen];
    } else {
      v += p.dX[e];
    }
  }
  p0.dX[e] = v;
}

// family check + the launch plan; DCTN_ERR_UNSUPPORTED outside the family
struct BdPlan {
  BdP p;
  int lds_bytes, nwg, QT, CH, NS, rec_len;
  size_t records_bytes, side_bytes;
};

int bd_plan(BdPlan& pl, const int64_t xs[5], const void* const* cores, int n, const int* out_sizes, const int* bond_sizes,
            const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q, int dtype) {
  if (dtype != DCTN_F32 || n < 3 || n > BD_NC || bond_sizes[0] != 1) return DCTN_ERR_UNSUPPORTED;
  int Ra = 0;
  for (int c = 1; c < n; ++c) {
    if (bond_sizes[c] < 1) return DCTN_ERR_UNSUPPORTED;
    Ra = bond_sizes[c] > Ra ? bond_sizes[c] : Ra;
  }
  if (Ra <= 4 || Ra > 16) return DCTN_ERR_UNSUPPORTED;   // (bonds <= 4: convsbs_reg takes them, lane = window)
  pl.NS = Ra <= 8 ? 2 : 4;
  int qc = 1;
  for (int c = 0; c < C; ++c) qc *= q;
  if (!((C == 1 && q >= 2 && q <= 4) || (C == 2 && q == 2))) return DCTN_ERR_UNSUPPORTED;
  BdP& p = pl.p;
  p.c2 = -1;
  int otot = 1;
  for (int c = 0; c < n; ++c) {
    if (out_sizes[c] < 1 || out_sizes[c] > 2) return DCTN_ERR_UNSUPPORTED;
    if (out_sizes[c] == 2) {
      if (p.c2 >= 0 || c == 0 || c == n - 1) return DCTN_ERR_UNSUPPORTED;
      p.c2 = c;
    }
    otot *= out_sizes[c];
  }
  pl.QT = qc;
  pl.CH = C == 2 ? 2 : 1;
  p.n = n; p.C = C; p.q = q; p.qc = qc; p.B = B; p.H = H; p.W = W; p.Otot = otot;
  int max_h = 0, max_w = 0, nin = 1;
  p.core_off[0] = 0;
  for (int c = 0; c < BD_NC; ++c) {
    const int cc = c < n ? c : n - 1;
    p.o[c] = out_sizes[cc];
    p.bl[c] = cc == 0 ? 1 : bond_sizes[cc];
    p.br[c] = cc == n - 1 ? 1 : bond_sizes[cc + 1];
    p.ph[c] = pos_h[cc];
    p.pw[c] = pos_w[cc];
    p.core[c] = cores ? (const float*)cores[cc] : nullptr;
    if (c < n) {
      p.nin[c] = nin;
      nin *= out_sizes[c];
      p.core_off[c + 1] = p.core_off[c] + p.o[c] * p.bl[c] * p.br[c] * qc;
      max_h = pos_h[c] > max_h ? pos_h[c] : max_h;
      max_w = pos_w[c] > max_w ? pos_w[c] : max_w;
    } else {
      p.nin[c] = 1;
      p.core_off[c + 1] = p.core_off[c];
    }
  }
  p.max_h = max_h;
  p.Ho = H - max_h; p.Wo = W - max_w;
  if (p.Ho < 1 || p.Wo < 1) return DCTN_ERR_BAD_SHAPE;
  for (int i = 0; i < 5; ++i) p.xs[i] = xs ? xs[i] : 0;
  // LDS plan: packs, tables, per chain wave two feature buffers (+ raw values), two hand-over buffers; the band's rows
  int off = 0;
  p.packF_off = off; off += BD_NPK * pl.QT * 256;
  p.packA_off = off; off += BD_NPK * pl.QT * 256;
  p.first_off = off; off += 64;
  p.last_off = off; off += 64;
  p.fs_off = off; off += 4 * BD_NFS * BD_FS;
  p.raw_off = off; off += pl.CH == 2 ? 4 * BD_NFS * BD_FS : 0;
  p.gv_off = off; off += 4 * 4 * BD_TILE;
  p.rows_off = off;   // (the kernel computes the same offsets at compile time)
  const int RK = pl.CH == 2 ? 4 : pl.QT;
  const int min_rows = max_h > 1 ? max_h : 1;
  // bands per image: enough workgroups for the chip, every band at least max_h rows, the band's gradient rows in LDS
  int nb = (bd_dev().cus + B - 1) / B;
  if (nb < 1) nb = 1;
  if (nb > p.Ho / min_rows) nb = p.Ho / min_rows;
  if (nb < 1) return DCTN_ERR_UNSUPPORTED;
  for (;;) {
    const int rows = (p.Ho + nb - 1) / nb;
    const long long rows_floats = (long long)rows * p.Wo * n * RK;
    if ((long long)(off + rows_floats) * 4 <= bd_dev().lds) {
      p.band_rows = rows;
      break;
    }
    ++nb;
    if ((p.Ho + nb - 1) / nb < min_rows || nb > p.Ho) return DCTN_ERR_UNSUPPORTED;
  }
  p.nb = (p.Ho + p.band_rows - 1) / p.band_rows;   // (the rounding can leave fewer bands than asked for)
  const int tiles = (p.band_rows * p.Wo + 15) / 16;
  p.iters = (tiles + 3) / 4;
  pl.lds_bytes = (off + p.band_rows * p.Wo * n * RK) * 4;
  pl.rec_len = BD_NPK * pl.QT * 256 + 128;
  // (the join area - two copies of the tiles + four first / last partial sums - lies in front of the band's rows: the packs
  // alone are as large as the two copies)
  pl.nwg = B * p.nb;
  pl.records_bytes = (size_t)pl.nwg * pl.rec_len * sizeof(float);
  pl.side_bytes = (size_t)B * (p.nb - 1) * 2 * max_h * W * C * q * sizeof(float);
  return DCTN_OK;
}

size_t bd_align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

bool convsbs_band_covers(int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B,
                         int H, int W, int q, int dtype) {
  BdPlan pl;
  return bd_plan(pl, nullptr, nullptr, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype) == DCTN_OK;
}

size_t convsbs_band_bwd_workspace(int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w, int C,
                                  int B, int H, int W, int q, int dtype) {
  BdPlan pl;
  if (bd_plan(pl, nullptr, nullptr, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype) != DCTN_OK) return 0;
  size_t extra = 0;
#ifdef DCTN_STAMPS
  extra = (size_t)pl.nwg * 2 * 32 * sizeof(long long);
#endif
  return bd_align256(pl.records_bytes) + bd_align256(pl.side_bytes) + 256 + extra;
}

int convsbs_bwd_band(const void* x, const int64_t xs[5], const void* const* cores, const void* dY, void* dX,
                     float* const* dcores, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h,
                     const int* pos_w, int C, int B, int H, int W, int q, int dtype, hipStream_t st, void* ws, size_t ws_bytes) {
  BdPlan pl;
  const int rc = bd_plan(pl, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  if (rc != DCTN_OK) return rc;
  if (!dX && !dcores) return DCTN_OK;
  const size_t need = bd_align256(pl.records_bytes) + bd_align256(pl.side_bytes);
  if (!ws || ws_bytes < need) return DCTN_ERR_WORKSPACE;
  BdP& p = pl.p;
  p.x = (const float*)x;
  p.dY = (const float*)dY;
  p.dX = (float*)dX;
  p.records = dcores ? (float*)ws : nullptr;
  p.side = (float*)((unsigned char*)ws + bd_align256(pl.records_bytes));
  p.stamps = nullptr;
#ifdef DCTN_STAMPS
  if (ws_bytes >= need + (size_t)pl.nwg * 2 * 32 * sizeof(long long)) p.stamps = (long long*)((unsigned char*)ws + need);
#endif
#define BD_LAUNCH(QTV, CHV, NSV)                                                                                     \
  do {                                                                                                          \
    if (hipFuncSetAttribute((const void*)convsbs_bwd_band_k<QTV, CHV, NSV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              pl.lds_bytes) != hipSuccess)                                                                    \
      return DCTN_ERR_UNSUPPORTED;   /* less LDS than the plan assumed */                                        \
    hipLaunchKernelGGL((convsbs_bwd_band_k<QTV, CHV, NSV>), dim3((unsigned)pl.nwg), dim3(BD_THREADS), pl.lds_bytes, st, p); \
  } while (0)
  if (pl.NS == 2) {
    if (pl.CH == 2) BD_LAUNCH(4, 2, 2);
    else if (pl.QT == 2) BD_LAUNCH(2, 1, 2);
    else if (pl.QT == 3) BD_LAUNCH(3, 1, 2);
    else BD_LAUNCH(4, 1, 2);
  } else {
    if (pl.CH == 2) BD_LAUNCH(4, 2, 4);
    else if (pl.QT == 2) BD_LAUNCH(2, 1, 4);
    else if (pl.QT == 3) BD_LAUNCH(3, 1, 4);
    else BD_LAUNCH(4, 1, 4);
  }
#undef BD_LAUNCH
  DCTN_CHECK_LAUNCH();
  BdTailP t;
  t.n = n; t.nrec = pl.nwg; t.total = p.core_off[n]; t.qc = p.qc; t.rec_len = pl.rec_len;
  for (int c = 0; c <= BD_NC; ++c) t.core_off[c] = p.core_off[c];
  for (int c = 0; c < BD_NC; ++c) { t.bl[c] = p.bl[c]; t.br[c] = p.br[c]; t.o[c] = p.o[c]; }
  t.c2 = p.c2;
  for (int c = 0; c < BD_NC; ++c) t.dcore[c] = (dcores && c < n) ? dcores[c] : nullptr;
  t.records = p.records; t.side = p.side; t.dX = p.dX;
  t.B = B; t.H = H; t.W = W; t.C = C; t.q = q; t.Cq = C * q; t.nb = p.nb; t.band_rows = p.band_rows; t.max_h = p.max_h;
  t.nshared = p.dX ? (long long)B * (p.nb - 1) * p.max_h * W * C * q : 0;
  const long long blocks = (p.records ? (t.rec_len + 63) / 64 : 0) + (t.nshared + 255) / 256;
  if (blocks > 0) {
    hipLaunchKernelGGL(convsbs_band_tail_k, dim3((unsigned)blocks), dim3(256), 0, st, t);
    DCTN_CHECK_LAUNCH();
  }
  dctn_set_last_kernel("convsbs_bwd_band_f32");
  return DCTN_OK;
}

int convsbs_fwd_band(const void* x, const int64_t xs[5], const void* const* cores, void* out, int n, const int* out_sizes,
                     const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q, int dtype,
                     hipStream_t st) {
  BdPlan pl;
  const int rc = bd_plan(pl, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  if (rc != DCTN_OK) return rc;
  const BdP& b = pl.p;
  BdFwdP p;
  p.x = (const float*)x; p.out = (float*)out;
  for (int c = 0; c < BD_NC; ++c) {
    p.core[c] = b.core[c]; p.o[c] = b.o[c]; p.bl[c] = b.bl[c]; p.br[c] = b.br[c]; p.ph[c] = b.ph[c]; p.pw[c] = b.pw[c]; p.nin[c] = b.nin[c];
  }
  for (int i = 0; i < 5; ++i) p.xs[i] = b.xs[i];
  p.n = n; p.C = C; p.q = q; p.qc = b.qc; p.B = B; p.H = H; p.W = W; p.Ho = b.Ho; p.Wo = b.Wo; p.Otot = b.Otot;
  p.Wn = (long long)B * b.Ho * b.Wo;
  p.ntiles = (p.Wn + 15) / 16;
  const int lds_bytes = (BD_NPK * pl.QT * 256 + 128 + 8 * 2 * BD_NC * 16 * 4) * 4;   // pack + tables + 8 waves x 2 feature buffers
  long long blocks = (p.ntiles + 7) / 8;
  const long long per_cu = (160 * 1024) / lds_bytes >= 2 ? 2 : 1;   // 128 registers: two workgroups (four waves per SIMD) per CU
  if (blocks > (long long)bd_dev().cus * per_cu) blocks = (long long)bd_dev().cus * per_cu;
#define BD_FLAUNCH(QTV, CHV, NSV)                                                                                    \
  do {                                                                                                          \
    if (hipFuncSetAttribute((const void*)convsbs_fwd_band_k<QTV, CHV, NSV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              lds_bytes) != hipSuccess)                                                                       \
      return DCTN_ERR_UNSUPPORTED;   /* less LDS than the plan assumed */                                        \
    hipLaunchKernelGGL((convsbs_fwd_band_k<QTV, CHV, NSV>), dim3((unsigned)blocks), dim3(BD_THREADS), lds_bytes, st, p); \
  } while (0)
  if (pl.NS == 2) {
    if (pl.CH == 2) BD_FLAUNCH(4, 2, 2);
    else if (pl.QT == 2) BD_FLAUNCH(2, 1, 2);
    else if (pl.QT == 3) BD_FLAUNCH(3, 1, 2);
    else BD_FLAUNCH(4, 1, 2);
  } else {
    if (pl.CH == 2) BD_FLAUNCH(4, 2, 4);
    else if (pl.QT == 2) BD_FLAUNCH(2, 1, 4);
    else if (pl.QT == 3) BD_FLAUNCH(3, 1, 4);
    else BD_FLAUNCH(4, 1, 4);
  }
#undef BD_FLAUNCH
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("convsbs_fwd_band_f32");
  return DCTN_OK;
}

// ---------------------------------------------------------------------------------- several strings per launch
namespace {
// the plans of the strings of one layer; DCTN_ERR_UNSUPPORTED unless every string is in the family with the same tile
// shape and band geometry (the tail kernel's one pass over dX relies on it)
int bd_plan_many(BdPlan (&pl)[BD_MANY], int ns, const int64_t xs[5], const void* const* cores, int n, const int* out_sizes,
                 const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q, int dtype) {
  if (ns != BD_MANY) return DCTN_ERR_UNSUPPORTED;
  for (int s2 = 0; s2 < ns; ++s2) {
    const int rc = bd_plan(pl[s2], xs, cores ? cores + s2 * n : nullptr, n, out_sizes + s2 * n, bond_sizes + s2 * n, pos_h + s2 * n,
                           pos_w + s2 * n, C, B, H, W, q, dtype);
    if (rc != DCTN_OK) return rc;
    if (s2 > 0) {
      const BdP &a = pl[0].p, &b = pl[s2].p;
      if (pl[s2].QT != pl[0].QT || pl[s2].CH != pl[0].CH || pl[s2].NS != pl[0].NS || pl[s2].lds_bytes != pl[0].lds_bytes || pl[s2].nwg != pl[0].nwg ||
          a.nb != b.nb || a.band_rows != b.band_rows || a.max_h != b.max_h || a.Ho != b.Ho || a.Wo != b.Wo)
        return DCTN_ERR_UNSUPPORTED;
    }
  }
  return DCTN_OK;
}
}  // namespace

size_t convsbs_many_band_bwd_workspace(int ns, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                                       int C, int B, int H, int W, int q, int dtype) {
  BdPlan pl[BD_MANY];
  if (bd_plan_many(pl, ns, nullptr, nullptr, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype) != DCTN_OK) return 0;
  size_t tot = 256;
  for (int s2 = 0; s2 < ns; ++s2) tot += bd_align256(pl[s2].records_bytes) + bd_align256(pl[s2].side_bytes);
  tot += (size_t)(ns - 1) * bd_align256((size_t)C * B * H * W * q * sizeof(float));   // the other strings' shares of dX
  return tot;
}

int convsbs_many_fwd_band(const void* x, const int64_t xs[5], const void* const* cores, void* const* outs, int ns, int n,
                          const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W,
                          int q, int dtype, hipStream_t st) {
  BdPlan pl[BD_MANY];
  const int rc = bd_plan_many(pl, ns, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  if (rc != DCTN_OK) return rc;
  BdFwdPMany pp;
  for (int s2 = 0; s2 < ns; ++s2) {
    if (!outs[s2]) return DCTN_ERR_NULL;
    const BdP& b = pl[s2].p;
    BdFwdP& p = pp.s[s2];
    p.x = (const float*)x; p.out = (float*)outs[s2];
    for (int c = 0; c < BD_NC; ++c) {
      p.core[c] = b.core[c]; p.o[c] = b.o[c]; p.bl[c] = b.bl[c]; p.br[c] = b.br[c]; p.ph[c] = b.ph[c]; p.pw[c] = b.pw[c]; p.nin[c] = b.nin[c];
    }
    for (int i = 0; i < 5; ++i) p.xs[i] = b.xs[i];
    p.n = n; p.C = C; p.q = q; p.qc = b.qc; p.B = B; p.H = H; p.W = W; p.Ho = b.Ho; p.Wo = b.Wo; p.Otot = b.Otot;
    p.Wn = (long long)B * b.Ho * b.Wo;
    p.ntiles = (p.Wn + 15) / 16;
  }
  const int lds_bytes = (BD_NPK * pl[0].QT * 256 + 128 + 8 * 2 * BD_NC * 16 * 4) * 4;
  long long blocks = (pp.s[0].ntiles + 7) / 8;
  const long long per_cu = (160 * 1024) / lds_bytes >= 2 ? 2 : 1;
  if (blocks > (long long)bd_dev().cus * per_cu / ns) blocks = (long long)bd_dev().cus * per_cu / ns;   // the strings' persistent waves are resident together
#define BD_FLAUNCH(QTV, CHV, NSV)                                                                                         \
  do {                                                                                                               \
    if (hipFuncSetAttribute((const void*)convsbs_fwd_band_many_k<QTV, CHV, NSV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              lds_bytes) != hipSuccess)                                                                            \
      return DCTN_ERR_UNSUPPORTED;   /* less LDS than the plan assumed */                                        \
    hipLaunchKernelGGL((convsbs_fwd_band_many_k<QTV, CHV, NSV>), dim3((unsigned)blocks, (unsigned)ns), dim3(BD_THREADS), lds_bytes, st, pp); \
  } while (0)
  if (pl[0].NS == 2) {
    if (pl[0].CH == 2) BD_FLAUNCH(4, 2, 2);
    else if (pl[0].QT == 2) BD_FLAUNCH(2, 1, 2);
    else if (pl[0].QT == 3) BD_FLAUNCH(3, 1, 2);
    else BD_FLAUNCH(4, 1, 2);
  } else {
    if (pl[0].CH == 2) BD_FLAUNCH(4, 2, 4);
    else if (pl[0].QT == 2) BD_FLAUNCH(2, 1, 4);
    else if (pl[0].QT == 3) BD_FLAUNCH(3, 1, 4);
    else BD_FLAUNCH(4, 1, 4);
  }
#undef BD_FLAUNCH
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("convsbs_many_fwd_band_f32");
  return DCTN_OK;
}

int convsbs_many_bwd_band(const void* x, const int64_t xs[5], const void* const* cores, const void* const* dYs, void* dX,
                          float* const* dcores, int ns, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h,
                          const int* pos_w, int C, int B, int H, int W, int q, int dtype, hipStream_t st, void* ws, size_t ws_bytes) {
  BdPlan pl[BD_MANY];
  const int rc = bd_plan_many(pl, ns, xs, cores, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  if (rc != DCTN_OK) return rc;
  if (!dX && !dcores) return DCTN_OK;
  const size_t need = convsbs_many_band_bwd_workspace(ns, n, out_sizes, bond_sizes, pos_h, pos_w, C, B, H, W, q, dtype);
  if (!ws || ws_bytes < need || ((uintptr_t)ws % 16)) return DCTN_ERR_WORKSPACE;
  BdPMany pp;
  BdTailMany tt;
  tt.ns = ns;
  unsigned char* cur = (unsigned char*)ws;
  for (int s2 = 0; s2 < ns; ++s2) {
    if (!dYs[s2]) return DCTN_ERR_NULL;
    BdP& p = pl[s2].p;
    p.x = (const float*)x;
    p.dY = (const float*)dYs[s2];
    p.records = dcores ? (float*)cur : nullptr;
    cur += bd_align256(pl[s2].records_bytes);
    p.side = (float*)cur;
    cur += bd_align256(pl[s2].side_bytes);
    if (!dX) p.dX = nullptr;
    else if (s2 == 0) p.dX = (float*)dX;
    else {
      p.dX = (float*)cur;
      cur += bd_align256((size_t)C * B * H * W * q * sizeof(float));
    }
    p.stamps = nullptr;
    pp.s[s2] = p;
    BdTailP& t = tt.s[s2];
    t.n = n; t.nrec = pl[s2].nwg; t.total = p.core_off[n]; t.qc = p.qc; t.rec_len = pl[s2].rec_len;
    for (int c = 0; c <= BD_NC; ++c) t.core_off[c] = p.core_off[c];
    for (int c = 0; c < BD_NC; ++c) { t.bl[c] = p.bl[c]; t.br[c] = p.br[c]; t.o[c] = p.o[c]; }
    t.c2 = p.c2;
    for (int c = 0; c < BD_NC; ++c) t.dcore[c] = (dcores && c < n) ? dcores[s2 * n + c] : nullptr;
    t.records = p.records; t.side = p.side; t.dX = p.dX;
    t.B = B; t.H = H; t.W = W; t.C = C; t.q = q; t.Cq = C * q; t.nb = p.nb; t.band_rows = p.band_rows; t.max_h = p.max_h;
    t.nshared = 0;
  }
#define BD_LAUNCH(QTV, CHV, NSV)                                                                                          \
  do {                                                                                                               \
    if (hipFuncSetAttribute((const void*)convsbs_bwd_band_many_k<QTV, CHV, NSV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              pl[0].lds_bytes) != hipSuccess)                                                                      \
      return DCTN_ERR_UNSUPPORTED;   /* less LDS than the plan assumed */                                        \
    hipLaunchKernelGGL((convsbs_bwd_band_many_k<QTV, CHV, NSV>), dim3((unsigned)pl[0].nwg, (unsigned)ns), dim3(BD_THREADS), pl[0].lds_bytes, \
                       st, pp);                                                                                      \
  } while (0)
  if (pl[0].NS == 2) {
    if (pl[0].CH == 2) BD_LAUNCH(4, 2, 2);
    else if (pl[0].QT == 2) BD_LAUNCH(2, 1, 2);
    else if (pl[0].QT == 3) BD_LAUNCH(3, 1, 2);
    else BD_LAUNCH(4, 1, 2);
  } else {
    if (pl[0].CH == 2) BD_LAUNCH(4, 2, 4);
    else if (pl[0].QT == 2) BD_LAUNCH(2, 1, 4);
    else if (pl[0].QT == 3) BD_LAUNCH(3, 1, 4);
    else BD_LAUNCH(4, 1, 4);
  }
#undef BD_LAUNCH
  DCTN_CHECK_LAUNCH();
  const long long rec_blocks = dcores ? (long long)ns * ((pl[0].rec_len + 63) / 64) : 0;
  const long long dx_blocks = dX ? ((long long)C * B * H * W * q + 255) / 256 : 0;
  if (rec_blocks + dx_blocks > 0) {
    hipLaunchKernelGGL(convsbs_band_tail_many_k, dim3((unsigned)(rec_blocks + dx_blocks)), dim3(256), 0, st, tt);
    DCTN_CHECK_LAUNCH();
  }
  dctn_set_last_kernel("convsbs_many_bwd_band_f32");
  return DCTN_OK;
}
