// EPS kernels, family "q2-f32": Q == 2 and a core small enough to live in registers (N = K*K*C in {8, 9}, out
// size <= 4: the 3x3 single-channel MNIST layer of BASELINE config 2 in the reference's OWN arithmetic -
// new_runner.py:417 runs float32 and nothing else).  Operands and accumulation are float32 on
// v_mfma_f32_32x32x2_f32, which is bit for bit an fmaf chain: the reference's numerics at the matrix rate
// (64 flop per clock and SIMD = the f32 vector peak, MI355X_MICROARCH.md).
//
// Forward (replaces dctn/eps.py:19-40).  With a = index over the first n0 factors (A = 2^n0), b over the last n1:
//     T[(b,o), w] = sum_a core[a,b,o] * P0[w,a]      <- MFMA: M = (b,o) rows, N = 32 windows, K = a (2 per instruction)
//     out[w,o]    = sum_b P1[w,b] * T[(b,o), w]      <- lane-local epilogue
// A lane owns ONE window (64 per wave step): it builds the P0 row of its window in registers as pairs
// (P0[2j], P0[2j+1]) - one v_pk_mul_f32 each - and one v_permlane32_swap per pair turns the pair into the lane's
// B operand for the step's two 32-window tiles (k = lane half).  The core (A operand) stays in registers for the
// whole kernel (A*BN*OP / 64 values per lane).  Per 64 windows: 64 matrix instructions of 64 cycles against
// ~130 vector instructions, so the kernel is paced by the matrix pipe; two waves per SIMD, one building its
// products while the other multiplies.
//
// Work decomposition of the forward: a workgroup owns ALL window positions of a few samples (so that a sample's
// logits can meet inside it when the head is fused) and deals the (sample, 64-position group) steps round-robin
// over its 8 waves; addressing is "per-lane constant + per-sample scalar" through raw buffer descriptors
// (lanes without a position read zeros and store nothing: the hardware range check).
//
// Backward dCore[a,b,o] = sum_w P0[w,a] P1[w,b] dY[w,o]: the windows must become the MFMA k index while lanes
// own windows.  Every lane writes P0 (A values), P1 (BN values) and dY (OP values) of its window into a
// per-wave LDS tile T[feature][window] (conflict-free 4-byte stores) and reads back, as lane (row r, k-half h),
// 16-byte pieces of feature rows: 4 consecutive windows = 4 k-steps per read (rows padded to 68 floats: the 16
// lanes a ds_read_b128 services together hit 16 different bank quads).  Z[w,(b,o)] = P1[w,b] dY[w,o] is formed
// AFTER the transposition (one packed multiply per two k-steps), so the tile holds 52 instead of 96 floats per
// window and 8 waves fit the CU's LDS.  With the classifier head fused, dY = dLogits x W is formed per lane from
// the lane's slice of the head weight (registers, loaded once: a workgroup works at ONE position group) and the
// sample's dLogits (scalar loads) - the feature gradient never exists in memory.  Per-wave sums stay in
// registers over all of the wave's samples, meet in LDS, leave as one tile per workgroup and are summed by the
// finishing kernel in a fixed order (no float atomics: bit-reproducible), which also forms the head's own
// gradients: dW = dLogits^T x features on v_mfma_f32_16x16x4_f32 (k = samples) and dBias.
#include "common.h"
#include "q2_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

namespace {

constexpr int QF_WAVES = 8;        // waves per workgroup, both kernels: two per SIMD
constexpr int QF_MAXN = 9;
constexpr int QF_NUM_CU = 256;
constexpr int QF_PITCH = 68;       // floats per row of the backward's transposition tile (64 windows + 4)

struct Q2fP {
  int B, O, P, Wo, npg;
  int spc;                       // forward: samples per workgroup; backward: samples per wave
  int ncb;                       // backward: chunk blocks (8 sample chunks each)
  unsigned foffb[2 * QF_MAXN];   // generic window mode: BYTE offset of (factor n, feature q) from the window's top-left pixel
  unsigned rowoffb[4];           // row modes: BYTE offset of window row (dh, ch), index dh*C + ch
  unsigned s1b, s2b, s3b;        // byte strides of x (batch, row, column)
  unsigned x_bytes;              // extent of x in bytes (< 2^31: buffer range check, 32-bit offsets)
  unsigned o_bytes, o_s1b;       // extent of out / dY in bytes (< 2^31), bytes per sample
  Q2FastDiv div_wo, div_npg;
  int ovec;                      // out / dY rows: O == OP and aligned -> one 16- / 8-byte access per window
  int Cout;                      // fused head: classes
  unsigned hw_rowb, hw_bytes;    //   bytes per row of the head weight (P * O * 4) and in total
  int opts;
};

// -------------------------------------------------------------------------------------------- window loads
// WIN = 1: K = 3, one channel, pixels contiguous: per window row one 16-byte + one 8-byte load (3 pixels x 2 features)
// WIN = 2: K = 2, two channels, pixels contiguous: per (row, channel) one 16-byte load (2 pixels x 2 features)
// WIN = 0: any strides: one 4-byte load per (factor, feature)
template <int N, int WIN>
struct RawWinF {
  u32x4 a[WIN == 1 ? 3 : (WIN == 2 ? 4 : 1)];
  u32x2 b[WIN == 1 ? 3 : 1];
  unsigned e[WIN == 0 ? 2 * N : 1];
};

template <int N, int WIN>
__device__ __forceinline__ void issue_win(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, const Q2fP& p,
                                          RawWinF<N, WIN>& raw) {
  if constexpr (WIN == 1) {
#pragma unroll
    for (int rw = 0; rw < 3; ++rw) {
      raw.a[rw] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + p.rowoffb[rw], 0);
      raw.b[rw] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff + p.rowoffb[rw] + 16u, 0);
    }
  } else if constexpr (WIN == 2) {
#pragma unroll
    for (int rw = 0; rw < 4; ++rw) raw.a[rw] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + p.rowoffb[rw], 0);
  } else {
#pragma unroll
    for (int i = 0; i < 2 * N; ++i) raw.e[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff + p.foffb[i], 0);
  }
}

template <int N, int WIN>
__device__ __forceinline__ void unpack_win(const RawWinF<N, WIN>& raw, float (&xv)[N][2]) {
  if constexpr (WIN == 1) {
    static_assert(WIN != 1 || N == 9, "3x3 single channel");
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {   // factor n = dh*3 + dw
      xv[3 * dh][0] = __uint_as_float(raw.a[dh].x);
      xv[3 * dh][1] = __uint_as_float(raw.a[dh].y);
      xv[3 * dh + 1][0] = __uint_as_float(raw.a[dh].z);
      xv[3 * dh + 1][1] = __uint_as_float(raw.a[dh].w);
      xv[3 * dh + 2][0] = __uint_as_float(raw.b[dh].x);
      xv[3 * dh + 2][1] = __uint_as_float(raw.b[dh].y);
    }
  } else if constexpr (WIN == 2) {
    static_assert(WIN != 2 || N == 8, "2x2 two channels");
#pragma unroll
    for (int rw = 0; rw < 4; ++rw) {   // row rw = dh*2 + ch; factor n = (dh*2 + dw)*2 + ch
      const int dh = rw >> 1, ch = rw & 1;
      xv[(dh * 2 + 0) * 2 + ch][0] = __uint_as_float(raw.a[rw].x);
      xv[(dh * 2 + 0) * 2 + ch][1] = __uint_as_float(raw.a[rw].y);
      xv[(dh * 2 + 1) * 2 + ch][0] = __uint_as_float(raw.a[rw].z);
      xv[(dh * 2 + 1) * 2 + ch][1] = __uint_as_float(raw.a[rw].w);
    }
  } else {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      xv[n][0] = __uint_as_float(raw.e[2 * n]);
      xv[n][1] = __uint_as_float(raw.e[2 * n + 1]);
    }
  }
}

// The whole P0 row of the lane's window as pairs pp[j] = (P0[2j], P0[2j+1]); a's MSB = factor 0.
// a = (hi: factors 0 .. N0-4 | lo: factors N0-3, N0-2 | last: factor N0-1): 2^(N0-1) + 6 + (hi table) packed multiplies.
template <int N0>
__device__ __forceinline__ void build_p0_pairs(const float (*xv)[2], f32x2 (&pp)[(1 << N0) / 2]) {
  static_assert(N0 >= 4, "at least one hi factor");
  constexpr int NH = N0 - 3, HI = 1 << NH;
  float u[HI];
  u[0] = xv[0][0];
  u[1] = xv[0][1];
#pragma unroll
  for (int n = 1; n < NH; ++n)
#pragma unroll
    for (int j = (1 << n) - 1; j >= 0; --j) {
      const f32x2 pr = q2_bmul2(u[j], f32x2{xv[n][0], xv[n][1]});
      u[2 * j] = pr[0];
      u[2 * j + 1] = pr[1];
    }
  const f32x2 xl = {xv[N0 - 2][0], xv[N0 - 2][1]}, xe = {xv[N0 - 1][0], xv[N0 - 1][1]};
  const f32x2 w01 = q2_bmul2(xv[N0 - 3][0], xl), w23 = q2_bmul2(xv[N0 - 3][1], xl);
  const f32x2 vp[4] = {q2_bmul2(w01[0], xe), q2_bmul2(w01[1], xe), q2_bmul2(w23[0], xe), q2_bmul2(w23[1], xe)};
#pragma unroll
  for (int i = 0; i < HI; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) pp[4 * i + jj] = q2_bmul2(u[i], vp[jj]);
}

#ifdef DCTN_STAMPS
// diagnostic build only (tools/stamp_q2f32.py): per wave 16 slots of s_memtime (shader clock) + the wave's HW_ID
__device__ unsigned long long qf_stamps[2048 * 16];
#define QF_STAMP(SLOT)                                                                                          \
  do {                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                 \
    if ((threadIdx.x & 63) == 0 && (SLOT) < 16 && blockIdx.x * QF_WAVES + (threadIdx.x >> 6) < 2048)            \
      qf_stamps[(blockIdx.x * QF_WAVES + (threadIdx.x >> 6)) * 16 + (SLOT)] = t_;                               \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
  } while (0)
#else
#define QF_STAMP(SLOT) do { } while (0)
#endif

// position of a lane in the sample: byte offsets of its window in x and of its row in out / dY
struct LanePos {
  unsigned voff_x, voff_o;
  bool valid;
};
__device__ __forceinline__ LanePos lane_pos(const Q2fP& p, int pg, int lane) {
  LanePos l;
  const int pos = pg * 64 + lane;
  l.valid = pos < p.P;
  const unsigned pu = l.valid ? (unsigned)pos : 0u;
  const unsigned ho = q2_fdiv(pu, p.div_wo), wo = pu - ho * (unsigned)p.Wo;
  l.voff_x = l.valid ? ho * p.s2b + wo * p.s3b : p.x_bytes;
  l.voff_o = l.valid ? pu * (unsigned)(p.O * 4) : p.o_bytes;
  return l;
}

// ------------------------------------------------------------------------------------------------ forward
// Row code of accumulator register v of M-tile t (lane-half bit h excluded), as in eps_mfma.hip:
//   code = (t << 4) | v;   o = code & (OP-1);   b = ((code >> LOGO) << 1) | h
//
// HEADC > 0: the linear head fused (EPSesPlusLinear's tail, dctn/eps_plus_linear.py:144-147: features = eps(core, x),
// logits = Linear(flatten(features))).  A workgroup holds ALL positions of its samples, in groups of at most
// FWD_GS = 4: every step also leaves its features in an LDS tile [sample][feature], and when the group's steps are
// done the workgroup forms logits[s][c] = bias[c] + sum_f W[c][f] * feat[s][f] as a tail phase on
// v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 blocks: rows = 4 classes, columns = the 4 samples, one feature per
// block and instruction): lane (blk, q) = (lane / 4, lane % 4) multiplies W[4 cg + q][f] (16-byte loads straight
// from memory in operand order, issued BEFORE the barrier that closes the group: they depend on nothing) with
// feat[q][f] (16-byte LDS reads) for f = 64 bs + 4 blk + e; the waves split the 64-feature steps round-robin, the 16
// blocks' and 8 waves' partial tiles meet in LDS and are summed in a fixed order.  Exact float32 (fma chains).
constexpr int FWD_GS = 4;          // samples per group = columns of the head product
constexpr int FWD_MAXBS = 6;       // 64-feature steps per wave of the head product: at most 8 * 6 * 64 = 3072 features
constexpr int fwd_tile_pitch(int F) { return (F + 63) / 64 * 64 + 16; }   // floats per sample row (pitch = 16 mod 64: the 16
                                                                          // lanes of a ds_read_b128 hit 16 bank quads)
template <int N0, int N1, int OP, int WIN, int HEADC>
__global__ __launch_bounds__(64 * QF_WAVES) void eps_fwd_q2f32_k(const float* __restrict__ x, const float* __restrict__ core,
                                                                 float* __restrict__ out, const float* __restrict__ hw,
                                                                 const float* __restrict__ bias, float* __restrict__ logits,
                                                                 Q2fP p) {
  constexpr int N = N0 + N1, A = 1 << N0, BN = 1 << N1, KS = A / 2, MT = BN * OP / 32, LOGO = q2_ilog2(OP);
  constexpr int CG = (HEADC + 3) / 4;   // class groups of the head product
  static_assert(MT >= 1 && OP >= 2 && OP <= 4, "tile shape");
  extern __shared__ __attribute__((aligned(16))) float fsm[];   // head: FWD_GS feature rows, then the partial logit tiles
  __shared__ unsigned step_ctr_mem;
  unsigned* step_ctr = &step_ctr_mem;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = (int)blockIdx.x * p.spc;
  const int nb = b0 + p.spc <= p.B ? p.spc : p.B - b0;   // samples of this workgroup (>= 1 by the launch)
  const __amdgpu_buffer_rsrc_t rs_x = q2_make_rsrc(x, p.x_bytes), rs_o = q2_make_rsrc(out, p.o_bytes);
  const int F = p.P * OP, LP = fwd_tile_pitch(F);
  float* hsum = fsm + FWD_GS * LP;   // [wave][cg][v][lane]
  if constexpr (HEADC > 0) {   // features past F of a row meet zero weights: they must be finite
    for (int e = tid; e < FWD_GS * (LP - F); e += 64 * QF_WAVES) fsm[(e / (LP - F)) * LP + F + e % (LP - F)] = 0.f;
  }

  // A operand: cf[t][ks] = core[a = 2 ks + h][b][o] for the lane's accumulator row r of M-tile t
  // (row r = (v & 3) + 8 (v >> 2) + 4 hh  <->  code = (t << 4) | v, b = ((code >> LOGO) << 1) | hh)
  float cf[MT][KS];
  {
    const int v = (r & 3) | ((r >> 3) << 2), hh = (r >> 2) & 1;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int code = (t << 4) | v, o = code & (OP - 1), b = ((code >> LOGO) << 1) | hh;
      const int oc = o < p.O ? o : 0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float val = core[(long long)((2 * ks + h) * BN + b) * p.O + oc];
        cf[t][ks] = o < p.O ? val : 0.f;
      }
    }
  }

  QF_STAMP(0);
#ifdef DCTN_STAMPS
  if (lane == 0 && blockIdx.x * QF_WAVES + wv < 2048) qf_stamps[(blockIdx.x * QF_WAVES + wv) * 16 + 15] = __builtin_amdgcn_s_getreg(63492);
  int stamp_i = 1;
#endif
  const int gs = HEADC > 0 ? FWD_GS : nb;   // no head: one group
  for (int g0 = 0; g0 < nb; g0 += gs) {
    const int ng = g0 + gs <= nb ? gs : nb - g0;
    // The group's (sample, position group) steps, i = s * npg + pg, are PULLED from a counter in LDS: the two waves of
    // a SIMD do not advance alike (the older one wins the issue arbitration: with a fixed 6 / 5 split it sat at the
    // group's barrier for 10 000 of 67 000 cycles while its partner finished alone - tools/stamp_q2f32.py), so each
    // takes a new step when it is done with one.  Tickets are drawn two steps ahead: the step whose window loads are in
    // flight is already chosen when the current one starts.
    const int nsteps = ng * p.npg;
    if (tid == 0) *step_ctr = 2 * QF_WAVES;   // steps 0 .. 15 are dealt: wave w starts with w and w + 8
    __syncthreads();
    auto draw = [&]() {
      unsigned v = 0;
      if (lane == 0) v = __hip_atomic_fetch_add(step_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return (int)__builtin_amdgcn_readfirstlane(v);
    };
    int cur = wv, nxt = wv + QF_WAVES;
    int s = (int)q2_fdiv((unsigned)cur, p.div_npg), pg = cur - s * p.npg;
    RawWinF<N, WIN> raw;
    LanePos lp = lane_pos(p, pg, lane);
    issue_win<N, WIN>(rs_x, cur < nsteps ? lp.voff_x : p.x_bytes, (unsigned)(b0 + g0 + (cur < nsteps ? s : 0)) * p.s1b, p, raw);
    while (cur < nsteps) {
      float xv[N][2];
      unpack_win<N, WIN>(raw, xv);
      const unsigned voff_o = lp.voff_o, soff_o = (unsigned)(b0 + g0 + s) * p.o_s1b;
      const bool valid = lp.valid;
      float* frow = fsm + s * LP + (pg * 64 + lane) * OP;
      {  // the wave's next step: its window loads go out before this step's arithmetic (past the last step every lane
         // is out of range and reads zeros: no control flow around the loads); and the ticket of the step after it
        cur = nxt;
        s = (int)q2_fdiv((unsigned)cur, p.div_npg);
        pg = cur - s * p.npg;
        lp = lane_pos(p, pg, lane);
        issue_win<N, WIN>(rs_x, cur < nsteps ? lp.voff_x : p.x_bytes, (unsigned)(b0 + g0 + (cur < nsteps ? s : 0)) * p.s1b, p, raw);
        nxt = draw();
        __builtin_amdgcn_sched_barrier(0);
      }
      f32x2 pp[KS];
      build_p0_pairs<N0>(xv, pp);
      float op0[KS], op1[KS];   // B operands of the windows of lanes 0-31 / 32-63
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        op0[j] = pp[j][0];
        op1[j] = pp[j][1];
        q2_swap_halves_inplace(op0[j], op1[j]);
      }
      // P1 over the last n1 factors: b bit u <-> factor N-1-u; bit 0 (factor N-1) is the lane half of the accumulator
      // row.  m0 / m1: multipliers this lane applies in tile 0 / tile 1.
      float m0[BN / 2], m1[BN / 2];
      {
        float ph[BN / 2];
        ph[0] = xv[N - 2][0];
        ph[1] = xv[N - 2][1];
#pragma unroll
        for (int u = 2; u < N1; ++u)
#pragma unroll
          for (int bh = (1 << (u - 1)) - 1; bh >= 0; --bh) {
            const f32x2 pr = q2_bmul2(ph[bh], f32x2{xv[N - 1 - u][0], xv[N - 1 - u][1]});
            ph[bh] = pr[0];
            ph[bh | (1 << (u - 1))] = pr[1];
          }
#pragma unroll
        for (int bh = 0; bh < BN / 2; ++bh) {
          const f32x2 mm = q2_bmul2(ph[bh], f32x2{xv[N - 1][0], xv[N - 1][1]});
          m0[bh] = mm[0];
          m1[bh] = mm[1];
          q2_swap_halves_inplace(m0[bh], m1[bh]);
        }
      }
      float res0[OP], res1[OP];
#pragma unroll
      for (int o = 0; o < OP; ++o) { res0[o] = 0.f; res1[o] = 0.f; }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          f32x16 acc;
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cf[t][ks], nt ? op1[ks] : op0[ks], acc, 0, 0, 0);
          // registers v, v + 1 (v even) are the outputs o, o + 1 of one b: one packed fused multiply-add per pair (the
          // f32 matrix instructions and the vector ALU share a pipe - tools/stamp_q2f32.py: a wave's time is the SUM of
          // its matrix and vector instruction times - so what counts is the number of vector instructions)
#pragma unroll
          for (int v = 0; v < 16; v += 2) {
            const int code = (t << 4) | v, o = code & (OP - 1);
            const float mm = nt ? m1[code >> LOGO] : m0[code >> LOGO];
            float* dst = nt ? res1 : res0;
            const f32x2 rr = __builtin_elementwise_fma(f32x2{mm, mm}, f32x2{acc[v], acc[v + 1]}, f32x2{dst[o], dst[o + 1]});
            dst[o] = rr[0];
            dst[o + 1] = rr[1];
          }
        }
      }
      // join the two row halves: lanes 0-31 end up with tile 0 (their own windows), lanes 32-63 with tile 1
      float res[OP];
#pragma unroll
      for (int o = 0; o < OP; ++o) {
        float a0 = res0[o], a1 = res1[o];
        q2_swap_halves_inplace(a0, a1);
        res[o] = a0 + a1;
      }
      if (p.ovec) {   // (lanes without a position: out of range, nothing stored)
        if constexpr (OP == 4)
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]),
                                                       __float_as_uint(res[3])}, rs_o, voff_o, soff_o, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(res[0]), __float_as_uint(res[1])}, rs_o, voff_o, soff_o, 0);
      } else {
#pragma unroll
        for (int o = 0; o < OP; ++o)
          if (o < p.O) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[o]), rs_o, voff_o, soff_o + 4u * o, 0);
      }
      if constexpr (HEADC > 0) {
        if (valid) {
          if constexpr (OP == 4) *reinterpret_cast<f32x4*>(frow) = f32x4{res[0], res[1], res[2], res[3]};
          else *reinterpret_cast<f32x2*>(frow) = f32x2{res[0], res[1]};
        }
      }
#ifdef DCTN_STAMPS
      QF_STAMP(stamp_i);
      if (stamp_i < 9) ++stamp_i;
#endif
    }
    if constexpr (HEADC > 0) {
      // ---- the group's head product.  Weight fragments first (nothing of the group is needed for them).
      const int blk = lane >> 2, q = lane & 3;
      const __amdgpu_buffer_rsrc_t rs_hw = q2_make_rsrc(hw, p.hw_bytes);
      const int nbs = (F + 63) / 64;
      u32x4 wf[FWD_MAXBS][CG];
#pragma unroll
      for (int i = 0; i < FWD_MAXBS; ++i) {
        const int bs = wv + QF_WAVES * i, f = 64 * bs + 4 * blk;
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
          const int c = 4 * cg + q;
          const unsigned vo = (bs < nbs && c < p.Cout && f + 3 < F) ? (unsigned)c * p.hw_rowb + (unsigned)f * 4u : p.hw_bytes;
          wf[i][cg] = __builtin_amdgcn_raw_buffer_load_b128(rs_hw, vo, 0, 0);
        }
      }
      QF_STAMP(10);
      __syncthreads();   // every wave's feature rows of the group are in the tile
      QF_STAMP(11);
      f32x4 hd[CG];
#pragma unroll
      for (int cg = 0; cg < CG; ++cg) hd[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < FWD_MAXBS; ++i) {
        const int bs = wv + QF_WAVES * i;
        if (bs < nbs) {   // wave-uniform
          const f32x4 fb = *reinterpret_cast<const f32x4*>(fsm + q * LP + 64 * bs + 4 * blk);
#pragma unroll
          for (int cg = 0; cg < CG; ++cg) {
            const float w4[4] = {__uint_as_float(wf[i][cg].x), __uint_as_float(wf[i][cg].y), __uint_as_float(wf[i][cg].z),
                                 __uint_as_float(wf[i][cg].w)};
#pragma unroll
            for (int e = 0; e < 4; ++e) hd[cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4[e], fb[e], hd[cg], 0, 0, 0);
          }
        }
      }
      // hd[cg][v] of lane (blk, q) = block blk's share of logits[sample q][class 4 cg + v]
#pragma unroll
      for (int cg = 0; cg < CG; ++cg)
#pragma unroll
        for (int v = 0; v < 4; ++v) hsum[((wv * CG + cg) * 4 + v) * 64 + lane] = hd[cg][v];
      __syncthreads();
      {  // thread (oi, w) = (tid / 8, tid % 8): output oi = (cg, v, s) of wave w's tiles, its 16 blocks in order; then the
         // 8 waves by three lane exchanges
        const int oi = tid >> 3, w = tid & 7, cg = oi >> 4, v = (oi >> 2) & 3, sl = oi & 3;
        float t = 0.f;
        if (cg < CG) {
#pragma unroll
          for (int bk = 0; bk < 16; ++bk) t += hsum[((w * CG + cg) * 4 + v) * 64 + 4 * bk + sl];
        }
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        const int c = 4 * cg + v;
        if (w == 0 && cg < CG && c < p.Cout && sl < ng) logits[(long long)(b0 + g0 + sl) * p.Cout + c] = t + bias[c];
      }
      QF_STAMP(12);
      // (the next group's steps write the tile only after every thread has passed the barrier above, and its partial
      //  tiles are written behind its own first barrier: the sums just read are safe)
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
constexpr size_t bwd_dyn_lds_bytes(int a, int bn, int op) { return (size_t)QF_WAVES * (a + bn + op) * QF_PITCH * sizeof(float); }

// HEADC > 0: `dY` points at dLogits (B, Cout) and `hw` at the head weight (Cout, P*O); the lane forms
// dY[w, o] = sum_c dLogits[b, c] * hw[c, pos*O + o] itself (HEADC = Cout padded to the instantiated bound).
template <int N0, int N1, int OP, int WIN, int HEADC>
__global__ __launch_bounds__(64 * QF_WAVES) void eps_bwd_q2f32_k(const float* __restrict__ x, const float* __restrict__ dY,
                                                                 const float* __restrict__ hw, float* __restrict__ partial,
                                                                 Q2fP p) {
  constexpr int N = N0 + N1, A = 1 << N0, BN = 1 << N1, MT = BN * OP / 32, LOGO = q2_ilog2(OP);
  constexpr int TROWS = A + BN + OP;
  static_assert(MT >= 1 && OP >= 2 && OP <= 4 && TROWS >= 32, "tile shape");
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* T = dsm + wv * (TROWS * QF_PITCH);

  // workgroup = (chunk block cb, position group pg), pg fastest; its 8 waves = the 8 sample chunks of cb
  // the batch is split evenly: chunk block cb owns samples [cb B / ncb, (cb + 1) B / ncb), its waves take n / 8 of them
  // each and the first n % 8 waves one more
  const int cb = (int)blockIdx.x / p.npg, pg = (int)blockIdx.x - cb * p.npg;
  const int c0 = (int)((long long)cb * p.B / p.ncb), c1 = (int)((long long)(cb + 1) * p.B / p.ncb);
  const int per = (c1 - c0) / QF_WAVES, rem = (c1 - c0) - per * QF_WAVES;
  const int sb0 = c0 + wv * per + (wv < rem ? wv : rem);
  const int sb1 = sb0 + per + (wv < rem ? 1 : 0);
  const LanePos lp = lane_pos(p, pg, lane);
  const __amdgpu_buffer_rsrc_t rs_x = q2_make_rsrc(x, p.x_bytes);

  RawWinF<N, WIN> raw;
  if (sb0 < sb1) issue_win<N, WIN>(rs_x, lp.voff_x, (unsigned)sb0 * p.s1b, p, raw);

  // head: the lane's slice of the classifier weight, W[c, pos*OP .. pos*OP + OP - 1] (rows >= Cout and lanes without a
  // position: out of range -> zeros); no head: rows of dY through the same kind of descriptor
  float hwf[HEADC > 0 ? HEADC : 1][OP];
  const __amdgpu_buffer_rsrc_t rs_dy = q2_make_rsrc(HEADC > 0 ? (const float*)nullptr : dY, HEADC > 0 ? 0u : p.o_bytes);
  u32x4 rawdy = {0u, 0u, 0u, 0u};
  auto issue_dy = [&](int b) {
    if constexpr (HEADC == 0) {
      if (p.ovec) {
        if constexpr (OP == 4) {
          rawdy = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, lp.voff_o, (unsigned)b * p.o_s1b, 0);
        } else {
          const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs_dy, lp.voff_o, (unsigned)b * p.o_s1b, 0);
          rawdy.x = t.x;
          rawdy.y = t.y;
        }
      } else {
        unsigned e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int o = 0; o < OP; ++o)
          if (o < p.O) e[o] = __builtin_amdgcn_raw_buffer_load_b32(rs_dy, lp.voff_o, (unsigned)b * p.o_s1b + 4u * o, 0);
        rawdy = u32x4{e[0], e[1], e[2], e[3]};
      }
    }
  };
  if constexpr (HEADC > 0) {
    const __amdgpu_buffer_rsrc_t rs_hw = q2_make_rsrc(hw, p.hw_bytes);
    const unsigned voff_hw = lp.valid ? (unsigned)(pg * 64 + lane) * (unsigned)(OP * 4) : p.hw_bytes;
#pragma unroll
    for (int c = 0; c < HEADC; ++c) {
      const unsigned vo = c < p.Cout ? voff_hw : p.hw_bytes;
      if constexpr (OP == 4) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_hw, vo, (unsigned)c * p.hw_rowb, 0);
        hwf[c][0] = __uint_as_float(t.x); hwf[c][1] = __uint_as_float(t.y);
        hwf[c][2] = __uint_as_float(t.z); hwf[c][3] = __uint_as_float(t.w);
      } else {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs_hw, vo, (unsigned)c * p.hw_rowb, 0);
        hwf[c][0] = __uint_as_float(t.x); hwf[c][1] = __uint_as_float(t.y);
      }
    }
  } else {
    if (sb0 < sb1) issue_dy(sb0);
  }

  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

  // read side of the tile: lane (r, h) reads windows 32 h + 4 j .. + 3 (= k-steps 4 j .. 4 j + 3 of its k-half) of
  // row r (P0: the B operand's column a = r), of row A + ((32 t + r) >> LOGO) (P1 of the A operand's row m = 32 t + r)
  // and of row A + BN + (r & (OP - 1)) (dY of that row's o)
  const float* rd_p0 = T + r * QF_PITCH + 32 * h;
  const float* rd_dy = T + (A + BN + (r & (OP - 1))) * QF_PITCH + 32 * h;
  const float* rd_p1[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) rd_p1[t] = T + (A + ((32 * t + r) >> LOGO)) * QF_PITCH + 32 * h;

  for (int b = sb0; b < sb1; ++b) {
    float xv[N][2];
    unpack_win<N, WIN>(raw, xv);
    float dyl[OP];
    if constexpr (HEADC == 0) {
      dyl[0] = __uint_as_float(rawdy.x);
      dyl[1] = __uint_as_float(rawdy.y);
      if constexpr (OP == 4) {
        dyl[2] = __uint_as_float(rawdy.z);
        dyl[3] = __uint_as_float(rawdy.w);
      }
    }
    {  // prefetch the next sample of this wave (the last step re-reads its own: no control flow around the loads)
      const int bn = b + 1 < sb1 ? b + 1 : b;
      issue_win<N, WIN>(rs_x, lp.voff_x, (unsigned)bn * p.s1b, p, raw);
      issue_dy(bn);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (HEADC > 0) {   // dLogits of the sample: wave-uniform (scalar loads)
      const float* dl = dY + (long long)b * p.Cout;
#pragma unroll
      for (int o = 0; o < OP; ++o) dyl[o] = 0.f;
#pragma unroll
      for (int c = 0; c < HEADC; ++c) {
        // HEADC == 10 is instantiated for Cout == 10 exactly (the row loads as s_load_dwordx8 + x2, no control flow);
        // the padded bound 16 reads a clamped index and selects
        const float d = HEADC == 10 ? dl[c] : (c < p.Cout ? dl[c < p.Cout ? c : p.Cout - 1] : 0.f);
#pragma unroll
        for (int o = 0; o < OP; ++o) dyl[o] = __builtin_fmaf(d, hwf[c][o], dyl[o]);
      }
    }
    f32x2 pp[A / 2];
    build_p0_pairs<N0>(xv, pp);
    float p1[BN];   // by doubling: (p1[bb], p1[bb | 1 << u]) = p1[bb] * (x[0], x[1]) (b bit u <-> factor N-1-u)
    p1[0] = xv[N - 1][0];
    p1[1] = xv[N - 1][1];
#pragma unroll
    for (int u = 1; u < N1; ++u)
#pragma unroll
      for (int bb = (1 << u) - 1; bb >= 0; --bb) {
        const f32x2 pr = q2_bmul2(p1[bb], f32x2{xv[N - 1 - u][0], xv[N - 1 - u][1]});
        p1[bb] = pr[0];
        p1[bb | (1 << u)] = pr[1];
      }
    q2_wave_lds_sync();   // the previous step's reads are done
#pragma unroll
    for (int j = 0; j < A / 2; ++j) {
      T[(2 * j) * QF_PITCH + lane] = pp[j][0];
      T[(2 * j + 1) * QF_PITCH + lane] = pp[j][1];
    }
#pragma unroll
    for (int bb = 0; bb < BN; ++bb) T[(A + bb) * QF_PITCH + lane] = p1[bb];
#pragma unroll
    for (int o = 0; o < OP; ++o) T[(A + BN + o) * QF_PITCH + lane] = dyl[o];
    q2_wave_lds_sync();
    // 8 groups of 4 k-steps; the reads of group j + 1 are issued BEFORE the 8 matrix instructions of group j (issued
    // behind them they came back ~100 cycles after the pipe had drained: a bubble per group)
    f32x4 bq = *reinterpret_cast<const f32x4*>(rd_p0), dq = *reinterpret_cast<const f32x4*>(rd_dy), pq[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) pq[t] = *reinterpret_cast<const f32x4*>(rd_p1[t]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f32x4 bqn = bq, dqn = dq, pqn[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) pqn[t] = pq[t];
      if (j + 1 < 8) {
        bqn = *reinterpret_cast<const f32x4*>(rd_p0 + 4 * (j + 1));
        dqn = *reinterpret_cast<const f32x4*>(rd_dy + 4 * (j + 1));
#pragma unroll
        for (int t = 0; t < MT; ++t) pqn[t] = *reinterpret_cast<const f32x4*>(rd_p1[t] + 4 * (j + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      // Z = P1 (x) dY for 4 k-steps as two packed multiplies per tile (written as text: the compiler scalarises a
      // general pair x pair product; the 2 wait states between a vector write and the matrix instruction reading it are
      // part of the text, the hazard recogniser does not see into it)
      float z[MT][4];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        f32x2 z01, z23;
        asm volatile("v_pk_mul_f32 %0, %2, %4\n\tv_pk_mul_f32 %1, %3, %5\n\ts_nop 1"
                     : "=&v"(z01), "=&v"(z23)
                     : "v"(f32x2{pq[t][0], pq[t][1]}), "v"(f32x2{pq[t][2], pq[t][3]}), "v"(f32x2{dq[0], dq[1]}), "v"(f32x2{dq[2], dq[3]}));
        z[t][0] = z01[0]; z[t][1] = z01[1]; z[t][2] = z23[0]; z[t][3] = z23[1];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(z[t][e], bq[e], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      bq = bqn;
      dq = dqn;
#pragma unroll
      for (int t = 0; t < MT; ++t) pq[t] = pqn[t];
    }
  }

  // workgroup reduction of the per-wave dCoreT tiles ([m = b*OP + o][a]), one coalesced store per workgroup
  static_assert((size_t)QF_WAVES * MT * 1024 * 4 <= bwd_dyn_lds_bytes(A, BN, OP), "dCore tiles must fit the transposition tiles");
  float* dst = partial + (long long)blockIdx.x * (MT * 1024);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = (v & 3) + 8 * (v >> 2) + 4 * h;
      dsm[(wv * MT + t) * 1024 + row * 32 + r] = acc[t][v];
    }
  __syncthreads();
  for (int e = tid; e < MT * 1024; e += 64 * QF_WAVES) {
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < QF_WAVES; ++k) sum += dsm[k * MT * 1024 + e];
    dst[e] = sum;
  }
}

// ------------------------------------------------------------------------------------- finishing kernel
// dW[c][f] = sum_b dLogits[b][c] * feat[b][f] for the FW features of one workgroup's slice, on v_mfma_f32_16x16x4_f32
// (rows = classes, columns = features, k = samples).  Wave w takes the samples [w spw, (w + 1) spw) in blocks of 4;
// lane (n, kg) = (lane % 16, lane / 16) loads, for sample kb + kg of a block, its FL features of the row and one value
// of dLogits; tile j of the wave is the features FL n + j.  The 16 waves' tiles meet in LDS, thread (slot, lane) sums
// one element in wave order (deterministic) and stores it.
constexpr int FIN_WAVES = 16;
template <int FL>
__device__ __forceinline__ void head_dw_role_f32(const float* __restrict__ feat, const float* __restrict__ dL,
                                                 float* __restrict__ dW, int B, int Cout, long long F, int blk,
                                                 float* __restrict__ lds) {
  constexpr int FW = 16 * FL, UN = 16;   // features per workgroup; k-blocks (4 samples each) in flight per wave
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kg = lane >> 4;
  const unsigned f_bytes = (unsigned)((long long)B * F * 4), dl_bytes = (unsigned)B * (unsigned)Cout * 4u;
  const __amdgpu_buffer_rsrc_t rs_f = q2_make_rsrc(feat, f_bytes), rs_dl = q2_make_rsrc(dL, dl_bytes);
  const long long fcol = (long long)blk * FW + FL * n;
  const bool fok = fcol + FL - 1 < F;   // (F is a multiple of FL: O == OP >= FL)
  const int spw = (((B + FIN_WAVES - 1) / FIN_WAVES) + 3) / 4 * 4;
  const int b0 = wv * spw < B ? wv * spw : B, b1 = b0 + spw < B ? b0 + spw : B;
  f32x4 acc[FL];
#pragma unroll
  for (int j = 0; j < FL; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kb = b0; kb < b1; kb += 4 * UN) {
    unsigned fr[UN][FL], a[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int b = kb + 4 * u + kg;
      const bool in = b < b1;
      const unsigned vo = (in && fok) ? (unsigned)((long long)b * F * 4 + fcol * 4) : f_bytes;
      if constexpr (FL == 1) {
        fr[u][0] = __builtin_amdgcn_raw_buffer_load_b32(rs_f, vo, 0, 0);
      } else {
        const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rs_f, vo, 0, 0);
        fr[u][0] = q.x;
        fr[u][FL - 1] = q.y;
      }
      a[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_dl, (in && n < Cout) ? (unsigned)b * (unsigned)Cout * 4u + 4u * n : dl_bytes, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int j = 0; j < FL; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[u]), __uint_as_float(fr[u][j]), acc[j], 0, 0, 0);
  }
  // acc[j][i] = dW[class 4 kg + i][feature FW blk + FL n + j] of this wave's samples
#pragma unroll
  for (int j = 0; j < FL; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) lds[(wv * (4 * FL) + j * 4 + i) * 64 + lane] = acc[j][i];
  __syncthreads();
  if (tid < 4 * FL * 64) {
    const int slot = tid >> 6, j = slot >> 2, i = slot & 3;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < FIN_WAVES; ++w) t += lds[(w * (4 * FL) + slot) * 64 + lane];
    const int c = 4 * kg + i;
    const long long f = (long long)blk * FW + FL * n + j;
    if (c < Cout && f < F) dW[(long long)c * F + f] = t;
  }
}

// Workgroups [0, n_core): dCore[a][b][o] = sum_blocks partial[blk][m = b*OP + o][a] for 32 consecutive (m, a)
// entries (thread (k32, c) streams every 32nd block's 128-byte segment: one round of independent loads for up to 256
// blocks); the next n_dw: the head-weight gradient slices; the last one: dBias.
template <int FIN_FL>
__global__ __launch_bounds__(64 * FIN_WAVES) void eps_q2f32_finish_k(const float* __restrict__ partial, float* __restrict__ dCore,
                                                                     int nblk, int A, int BN, int O, int OP, int n_core,
                                                                     const float* __restrict__ feat, const float* __restrict__ dL,
                                                                     float* __restrict__ dW, float* __restrict__ dBias, int B,
                                                                     int Cout, long long F, int n_dw) {
  __shared__ float red[32][33];
  __shared__ float gemm_lds[FIN_WAVES * 4 * FIN_FL * 64];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < n_core) {
    const int c = tid & 31, k32 = tid >> 5;
    const long long stride = (long long)BN * OP * 32;
    const long long e = (long long)blockIdx.x * 32 + c;  // flat (m, a) index, 32 columns per row
    float acc = 0.f;
    for (int k0 = 0; k0 < nblk; k0 += 256) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = k0 + k32 + 32 * i;
        const float ld = partial[(k < nblk ? k : nblk - 1) * stride + e];   // (clamped index + select: no branch between the loads)
        v[i] = k < nblk ? ld : 0.f;
      }
      acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    red[k32][c] = acc;
    __syncthreads();
    if (k32 == 0) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) t += red[i][c];
      const int m = (int)(e >> 5), a = (int)(e & 31);
      const int b = m / OP, o = m % OP;
      if (a < A && o < O) dCore[((long long)a * BN + b) * O + o] = t;
    }
    return;
  }
  if ((int)blockIdx.x < n_core + n_dw) {
    if (dW) head_dw_role_f32<FIN_FL>(feat, dL, dW, B, Cout, F, (int)blockIdx.x - n_core, gemm_lds);
    return;
  }
  if (!dBias) return;
  {  // dBias[c] = sum_b dLogits[b, c]: thread (c = tid % 16, j = tid / 16) takes every 64th sample
    const int c = tid & 15, j = tid >> 4;
    float t = 0.f;
    if (c < Cout)
      for (int b = j; b < B; b += 64) t += dL[(long long)b * Cout + c];
    __shared__ float rb[64][17];
    rb[j][c] = t;
    __syncthreads();
    if (tid < 16 && tid < Cout) {
      float u = 0.f;
#pragma unroll 8
      for (int i = 0; i < 64; ++i) u += rb[i][tid];
      dBias[tid] = u;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
int next_pow2(int v) {
  int r = 1;
  while (r < v) r <<= 1;
  return r;
}

bool family_ok(const EpsP& p, int dtype, int precision) {
  if (dtype != DCTN_F32 || precision == DCTN_PREC_BF16) return false;
  if (p.Q != 2 || (p.N != 8 && p.N != 9) || p.O > 4) return false;
  {  // 32-bit byte offsets with the hardware range check: non-negative strides, extents below 2 GiB
    long long ext = 0;
    const long long dims[5] = {p.C, p.B, p.H, p.W, p.Q};
    for (int i = 0; i < 5; ++i) {
      if (p.s[i] < 0) return false;
      ext += (dims[i] - 1) * p.s[i];
    }
    if ((ext + 1) * 4 >= (1LL << 31)) return false;
    if (p.Wn * p.O * 4 >= (1LL << 31)) return false;
  }
  return true;
}

// window mode of a shape / layout (template parameter WIN)
int window_mode(const EpsP& p, const void* x) {
  const bool rows = p.s[4] == 1 && p.s[3] == 2 && ((uintptr_t)x % 8) == 0 && p.s[0] % 2 == 0 && p.s[1] % 2 == 0 && p.s[2] % 2 == 0;
  if (rows && p.N == 9 && p.C == 1 && p.K == 3) return 1;
  if (rows && p.N == 8 && p.C == 2 && p.K == 2) return 2;
  return 0;
}

void fill_qp(Q2fP& m, const EpsP& p) {
  for (int n = 0; n < p.N && n < QF_MAXN; ++n) {
    const int pos = n / p.C, ch = n - pos * p.C;
    const int dh = pos / p.K, dw = pos - dh * p.K;
    for (int q = 0; q < 2; ++q) m.foffb[2 * n + q] = (unsigned)((ch * p.s[0] + dh * p.s[2] + dw * p.s[3] + q * p.s[4]) * 4);
  }
  for (int rw = 0; rw < 4; ++rw) m.rowoffb[rw] = 0;
  for (int rw = 0; rw < p.K * p.C && rw < 4; ++rw) {
    const int dh = rw / p.C, ch = rw - dh * p.C;
    m.rowoffb[rw] = (unsigned)((ch * p.s[0] + dh * p.s[2]) * 4);
  }
  m.s1b = (unsigned)(p.s[1] * 4); m.s2b = (unsigned)(p.s[2] * 4); m.s3b = (unsigned)(p.s[3] * 4);
  long long ext = 0;
  const long long dims[5] = {p.C, p.B, p.H, p.W, p.Q};
  for (int i = 0; i < 5; ++i) ext += (dims[i] - 1) * p.s[i];
  m.x_bytes = (unsigned)((ext + 1) * 4);
  m.B = p.B; m.O = p.O; m.Wo = p.Wo;
  m.P = p.Ho * p.Wo;
  m.npg = (m.P + 63) / 64;
  m.spc = 1; m.ncb = 0;
  m.o_s1b = (unsigned)((long long)m.P * p.O * 4);
  m.o_bytes = (unsigned)(p.Wn * p.O * 4);
  m.div_wo = q2_make_fastdiv((unsigned)p.Wo);
  m.div_npg = q2_make_fastdiv((unsigned)m.npg);
  m.ovec = 0;
  m.Cout = 0; m.hw_rowb = 0; m.hw_bytes = 0;
  m.opts = p.opts;
}

bool row_vec_ok(const Q2fP& m, int OP, const void* ptr) { return m.O == OP && ((uintptr_t)ptr % (OP * 4)) == 0; }

template <int N0, int N1, int OP>
int fwd_launch_t(const void* x, const void* core, void* out, Q2fP m, int win, hipStream_t st) {
  // one or two workgroups per CU (the kernel holds ~120 registers: four waves per SIMD fit)
  int nwg = m.B < 2 * QF_NUM_CU ? m.B : 2 * QF_NUM_CU;
  m.spc = (m.B + nwg - 1) / nwg;
  nwg = (m.B + m.spc - 1) / m.spc;
  m.ovec = row_vec_ok(m, OP, out) ? 1 : 0;
  const dim3 g((unsigned)nwg), b(64 * QF_WAVES);
  constexpr int WROWS = (N0 + N1) == 9 ? 1 : 2;
  if (win != 0)
    hipLaunchKernelGGL((eps_fwd_q2f32_k<N0, N1, OP, WROWS, 0>), g, b, 0, st, (const float*)x, (const float*)core, (float*)out,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, m);
  else
    hipLaunchKernelGGL((eps_fwd_q2f32_k<N0, N1, OP, 0, 0>), g, b, 0, st, (const float*)x, (const float*)core, (float*)out,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, m);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("eps_fwd_q2f32");
  return DCTN_OK;
}

#define QF_FWD_HEAD_LAUNCH(WINV, HC)                                                                                       \
  do {                                                                                                                     \
    const size_t dyn = ((size_t)FWD_GS * fwd_tile_pitch(m.P * OP) + (size_t)QF_WAVES * ((HC + 3) / 4) * 4 * 64) * sizeof(float); \
    if (dyn > 150 * 1024) return DCTN_ERR_UNSUPPORTED;                                                                      \
    (void)hipFuncSetAttribute((const void*)eps_fwd_q2f32_k<N0, N1, OP, WINV, HC>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)dyn);                                                                                   \
    hipLaunchKernelGGL((eps_fwd_q2f32_k<N0, N1, OP, WINV, HC>), g, b, dyn, st, (const float*)x, (const float*)core,         \
                       (float*)out, (const float*)hw, (const float*)bias, (float*)logits, m);                              \
  } while (0)

// forward of (EPS layer -> flatten -> linear head) as one kernel
template <int N0, int N1, int OP>
int fwd_head_launch_t(const void* x, const void* core, const void* hw, const void* bias, void* out, void* logits, Q2fP m,
                      int win, hipStream_t st) {
  const long long F = (long long)m.P * OP;
  if (m.O != OP || F % 4 != 0 || F > (long long)QF_WAVES * FWD_MAXBS * 64) return DCTN_ERR_UNSUPPORTED;
  if (((uintptr_t)hw % 16) != 0 || ((uintptr_t)out % (OP * 4)) != 0) return DCTN_ERR_UNSUPPORTED;
  // one workgroup per CU (166 registers with the head's weight fragments: no second workgroup fits beside it)
  int nwg = m.B < QF_NUM_CU ? m.B : QF_NUM_CU;
  m.spc = (m.B + nwg - 1) / nwg;
  nwg = (m.B + m.spc - 1) / m.spc;
  m.ovec = 1;
  const dim3 g((unsigned)nwg), b(64 * QF_WAVES);
  constexpr int WROWS = (N0 + N1) == 9 ? 1 : 2;
  if (m.Cout <= 12) {
    if (win != 0) QF_FWD_HEAD_LAUNCH(WROWS, 12); else QF_FWD_HEAD_LAUNCH(0, 12);
  } else {
    if (win != 0) QF_FWD_HEAD_LAUNCH(WROWS, 16); else QF_FWD_HEAD_LAUNCH(0, 16);
  }
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("eps_head_fwd_q2f32");
  return DCTN_OK;
}
#undef QF_FWD_HEAD_LAUNCH

// the backward's grid: ncb chunk blocks x npg position groups workgroups (<= QF_NUM_CU: one partial tile each)
int plan_bwd(Q2fP& m) {
  if (m.npg > QF_NUM_CU) return 0;
  long long ncb = QF_NUM_CU / m.npg;
  const long long need = (m.B + QF_WAVES - 1) / QF_WAVES;   // at least one sample per wave where the batch allows
  if (ncb > need) ncb = need;
  m.ncb = (int)ncb;
  m.spc = 0;
  return m.ncb * m.npg;
}

#define QF_BWD_LAUNCH(WINV, HC)                                                                                          \
  do {                                                                                                                   \
    (void)hipFuncSetAttribute((const void*)eps_bwd_q2f32_k<N0, N1, OP, WINV, HC>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)DYN);                                                                                 \
    hipLaunchKernelGGL((eps_bwd_q2f32_k<N0, N1, OP, WINV, HC>), g, b, DYN, st, (const float*)x, (const float*)dY,         \
                       (const float*)hw, (float*)ws, m);                                                                 \
  } while (0)

// dCore (and, with a head, dW / dBias): main kernel + finishing kernel
template <int N0, int N1, int OP>
int bwd_launch_t(const void* x, const void* dY, const void* hw, const void* feat, void* dCore, void* dW, void* dBias,
                 void* ws, Q2fP m, int win, hipStream_t st) {
  constexpr int A = 1 << N0, BN = 1 << N1, MT = BN * OP / 32;
  constexpr size_t DYN = bwd_dyn_lds_bytes(A, BN, OP);
  constexpr int WROWS = (N0 + N1) == 9 ? 1 : 2;
  const bool head = hw != nullptr;
  const int grid = plan_bwd(m);
  if (grid == 0) return DCTN_ERR_UNSUPPORTED;
  if (head && m.O != OP) return DCTN_ERR_UNSUPPORTED;
  m.ovec = (!head && row_vec_ok(m, OP, dY)) ? 1 : 0;
  const dim3 g((unsigned)grid), b(64 * QF_WAVES);
  if (head) {
    if (m.Cout == 10) {
      if (win != 0) QF_BWD_LAUNCH(WROWS, 10); else QF_BWD_LAUNCH(0, 10);
    } else {
      if (win != 0) QF_BWD_LAUNCH(WROWS, 16); else QF_BWD_LAUNCH(0, 16);
    }
  } else {
    if (win != 0) QF_BWD_LAUNCH(WROWS, 0); else QF_BWD_LAUNCH(0, 0);
  }
  DCTN_CHECK_LAUNCH();
  if (m.opts & DCTN_OPT_MAIN_KERNEL_ONLY) return DCTN_PARTIAL;   // measurement option: partial sums only, gradients NOT written
  const int n_core = MT * 32;
  const long long F = (long long)m.P * OP;
  // head-weight gradient: 16-feature slices (64-byte pieces of the feature rows: 5.3 us for the finishing kernel at
  // B = 1024 against 6.7 with 32-feature slices - a sector is the least a piece costs, and 169 workgroups instead of 85)
  const int n_dw = (head && dW) ? (int)((F + 15) / 16) : 0;
  hipLaunchKernelGGL(eps_q2f32_finish_k<1>, dim3(n_core + n_dw + (head && dBias ? 1 : 0)), dim3(64 * FIN_WAVES), 0, st,
                     (const float*)ws, (float*)dCore, grid, A, BN, m.O, OP, n_core, (const float*)feat, (const float*)dY,
                     (float*)dW, (float*)dBias, m.B, m.Cout, F, n_dw);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel(head ? "eps_head_bwd_q2f32" : "eps_bwd_q2f32");
  return DCTN_OK;
}

#undef QF_BWD_LAUNCH

}  // namespace

#ifdef DCTN_STAMPS
extern "C" int dctn_debug_read_qf_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(qf_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

bool eps_q2f32_covers(const EpsP& p, int dtype, int precision) { return family_ok(p, dtype, precision); }

int eps_fwd_q2f32(const void* x, const void* core, void* out, const EpsP& p, int dtype, int precision, hipStream_t st) {
  if (!family_ok(p, dtype, precision)) return DCTN_ERR_UNSUPPORTED;
  Q2fP m;
  fill_qp(m, p);
  const int win = window_mode(p, x), op = p.O <= 2 ? 2 : 4;
  if (p.N == 9) return op == 2 ? fwd_launch_t<5, 4, 2>(x, core, out, m, win, st) : fwd_launch_t<5, 4, 4>(x, core, out, m, win, st);
  return op == 2 ? fwd_launch_t<4, 4, 2>(x, core, out, m, win, st) : fwd_launch_t<4, 4, 4>(x, core, out, m, win, st);
}

size_t eps_bwd_q2f32_workspace(const EpsP& p, int dtype, int precision) {
  if (!family_ok(p, dtype, precision)) return 0;
  return (size_t)QF_NUM_CU * 2 * 1024 * sizeof(float);   // one (MT <= 2) x 32 x 32 tile per workgroup
}

// dCore only; the caller (capi) sends dX to the large-core family.
int eps_bwd_q2f32(const void* x, const void* dY, void* dCore, void* ws, size_t ws_bytes, const EpsP& p, int dtype,
                  int precision, hipStream_t st) {
  if (!dCore) return DCTN_ERR_UNSUPPORTED;
  if (!family_ok(p, dtype, precision)) return DCTN_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < eps_bwd_q2f32_workspace(p, dtype, precision)) return DCTN_ERR_WORKSPACE;
  Q2fP m;
  fill_qp(m, p);
  const int win = window_mode(p, x), op = p.O <= 2 ? 2 : 4;
  if (p.N == 9)
    return op == 2 ? bwd_launch_t<5, 4, 2>(x, dY, nullptr, nullptr, dCore, nullptr, nullptr, ws, m, win, st)
                   : bwd_launch_t<5, 4, 4>(x, dY, nullptr, nullptr, dCore, nullptr, nullptr, ws, m, win, st);
  return op == 2 ? bwd_launch_t<4, 4, 2>(x, dY, nullptr, nullptr, dCore, nullptr, nullptr, ws, m, win, st)
                 : bwd_launch_t<4, 4, 4>(x, dY, nullptr, nullptr, dCore, nullptr, nullptr, ws, m, win, st);
}

static bool head_ok(const EpsP& p, int Cout, int dtype, int precision) {
  if (!family_ok(p, dtype, precision)) return false;
  if (p.O != 2 && p.O != 4) return false;   // the weight slice of a lane is one 8- / 16-byte piece
  if (Cout < 1 || Cout > 16) return false;
  const long long P = (long long)p.Ho * p.Wo;
  if ((P + 63) / 64 > QF_NUM_CU) return false;
  return P * p.O * 4 * Cout < (1LL << 31) && (long long)p.B * Cout * 4 < (1LL << 31);
}

// Forward of the same tail: features (stored, the backward needs them) and logits from one kernel.
int eps_head_fwd_q2f32(const void* x, const void* core, const void* head_w, const void* bias, void* feat, void* logits,
                       const EpsP& p, int Cout, int dtype, int precision, hipStream_t st) {
  if (!head_ok(p, Cout, dtype, precision)) return DCTN_ERR_UNSUPPORTED;
  Q2fP m;
  fill_qp(m, p);
  m.Cout = Cout;
  const long long rowb = (long long)m.P * p.O * 4;
  m.hw_rowb = (unsigned)rowb;
  m.hw_bytes = (unsigned)(rowb * Cout);
  const int win = window_mode(p, x);
  if (p.N == 9)
    return p.O == 2 ? fwd_head_launch_t<5, 4, 2>(x, core, head_w, bias, feat, logits, m, win, st)
                    : fwd_head_launch_t<5, 4, 4>(x, core, head_w, bias, feat, logits, m, win, st);
  return p.O == 2 ? fwd_head_launch_t<4, 4, 2>(x, core, head_w, bias, feat, logits, m, win, st)
                  : fwd_head_launch_t<4, 4, 4>(x, core, head_w, bias, feat, logits, m, win, st);
}

size_t eps_head_bwd_q2f32_workspace(const EpsP& p, int Cout, int dtype, int precision) {
  return head_ok(p, Cout, dtype, precision) ? eps_bwd_q2f32_workspace(p, dtype, precision) : 0;
}

// Backward of (EPS layer -> flatten -> linear head) in one pass over x: dCore, dW and dBias from dLogits, the head
// weight and the layer's forward output; dY is never materialised.
int eps_head_bwd_q2f32(const void* x, const void* feat, const void* dLogits, const void* head_w, void* dCore, void* dW,
                       void* dBias, void* ws, size_t ws_bytes, const EpsP& p, int Cout, int dtype, int precision,
                       hipStream_t st) {
  if (!dCore) return DCTN_ERR_UNSUPPORTED;
  if (!head_ok(p, Cout, dtype, precision)) return DCTN_ERR_UNSUPPORTED;
  if (((uintptr_t)head_w % (p.O * 4)) != 0 || ((uintptr_t)feat % 8) != 0 || ((uintptr_t)dLogits % 4) != 0) return DCTN_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < eps_head_bwd_q2f32_workspace(p, Cout, dtype, precision)) return DCTN_ERR_WORKSPACE;
  Q2fP m;
  fill_qp(m, p);
  m.Cout = Cout;
  const long long rowb = (long long)m.P * p.O * 4;
  m.hw_rowb = (unsigned)rowb;
  m.hw_bytes = (unsigned)(rowb * Cout);
  const int win = window_mode(p, x);
  if (p.N == 9)
    return p.O == 2 ? bwd_launch_t<5, 4, 2>(x, dLogits, head_w, feat, dCore, dW, dBias, ws, m, win, st)
                    : bwd_launch_t<5, 4, 4>(x, dLogits, head_w, feat, dCore, dW, dBias, ws, m, win, st);
  return p.O == 2 ? bwd_launch_t<4, 4, 2>(x, dLogits, head_w, feat, dCore, dW, dBias, ws, m, win, st)
                  : bwd_launch_t<4, 4, 4>(x, dLogits, head_w, feat, dCore, dW, dBias, ws, m, win, st);
}
