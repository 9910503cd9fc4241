// Linear classifier head of EPSesPlusLinear (reference: dctn/eps_plus_linear.py:147,
// `self.linear(rearrange(intermediate, "b h w q -> b (h w q)"))`, nn.Linear(H'W'Q, 10)).
//
// With 10 outputs the three GEMMs of the head (forward, dFeat, dWeight) are skinny: a library
// GEMM spends 8-12 us on each at batch 1024, more than the EPS contraction itself.  These
// kernels are bf16-in / f32-accumulate and bound by streaming `feat` (B x F) once:
//   forward : v_mfma_f32_16x16x32_bf16, 16 samples x 16 (padded) classes per workgroup, the F
//             dimension split over the 8 waves (every operand load issued up front: one memory
//             round trip), LDS reduction, bias add.
//   dFeat   : one lane per 8 consecutive features, the 16-byte weight chunks of all classes are
//             loaded once and reused for 8 samples.
//   dWeight : one lane per 4 features x all classes, samples split over grid.y into partial
//             slices; a second kernel sums the slices in a fixed order (deterministic) and emits
//             dBias as well.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int HEAD_MAXC = 16;

// ------------------------------------------------------------------------------------ forward
constexpr int HF_MAXK = 12;  // k-steps (of 32 features) one wave can hold in flight

// ROWS samples per workgroup (rows of the 16 x 16 tile that carry a sample): 16, or 4 when 16 would leave most of
// the 256 CUs without a workgroup (B = 1024: 64 -> 256 workgroups; the idle tile rows cost nothing, the kernel
// is one memory round trip long).  cfg2 step: 39.0 us with 16 rows, 38.4 (8), 37.5 (4), 39.2 (2), 42.1 (1); 16 waves
// per workgroup instead of 8: no change.
template <int ROWS>
__global__ __launch_bounds__(512) void head_fwd_k(const bf16_t* __restrict__ feat,
                                                  const bf16_t* __restrict__ W,
                                                  const bf16_t* __restrict__ bias,
                                                  bf16_t* __restrict__ out, long long B, int F,
                                                  int Cout) {
  __shared__ float red[8][16 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long long b = (long long)blockIdx.x * ROWS + r;    // sample of this lane's A rows
  const bool bok = r < ROWS && b < B, cok = r < Cout;
  const int ksteps = (F + 31) / 32;
  const int per = (ksteps + 7) / 8;                        // k-steps per wave
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const bf16_t* fa = feat + (bok ? b : 0) * (long long)F;
  const bf16_t* wb = W + (long long)(cok ? r : 0) * F;
  for (int s0 = wv * per; s0 < (wv + 1) * per && s0 < ksteps; s0 += HF_MAXK) {
    // every operand load of this wave is issued before the first MFMA: one memory round trip
    bf16x8 av[HF_MAXK], bv[HF_MAXK];
#pragma unroll
    for (int u = 0; u < HF_MAXK; ++u) {
      const int s = s0 + u;
      const int k0 = s * 32 + 8 * g;
      const bool in = s < (wv + 1) * per && k0 + 8 <= F;   // F % 8 == 0: a chunk is in or out
      const int kc = in ? k0 : 0;
      av[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      bv[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (in && bok) av[u] = *reinterpret_cast<const bf16x8*>(fa + kc);
      if (in && cok) bv[u] = *reinterpret_cast<const bf16x8*>(wb + kc);
    }
#pragma unroll
    for (int u = 0; u < HF_MAXK; ++u)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u], bv[u], acc, 0, 0, 0);
  }
  // D: col = lane & 15 (class), row = 4 * (lane >> 4) + reg (sample)
#pragma unroll
  for (int v = 0; v < 4; ++v) red[wv][(4 * g + v) * 16 + r] = acc[v];
  __syncthreads();
  if (tid < 256) {
    const int row = tid >> 4, c = tid & 15;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][tid];
    const long long bb = (long long)blockIdx.x * ROWS + row;
    if (row < ROWS && bb < B && c < Cout) out[bb * Cout + c] = (bf16_t)(s + (float)bias[c]);
  }
}

// ------------------------------------------------------------------------------------ dFeat
constexpr int DF_SPB = 8;  // samples per workgroup

__device__ __forceinline__ void head_bwd_dfeat_role(const bf16_t* __restrict__ W,
                                                    const bf16_t* __restrict__ dOut,
                                                    bf16_t* __restrict__ dFeat, long long B, int F,
                                                    int Cout, int bx, int by, float (*gs)[HEAD_MAXC]) {
  const int tid = threadIdx.x;
  const int chunk = bx * 256 + tid;                         // 8 consecutive features
  const long long b0 = (long long)by * DF_SPB;
  if (tid < DF_SPB * HEAD_MAXC) {
    const int s = tid / HEAD_MAXC, c = tid % HEAD_MAXC;
    gs[s][c] = (b0 + s < B && c < Cout) ? (float)dOut[(b0 + s) * Cout + c] : 0.f;
  }
  const bool ok = chunk < F / 8;
  bf16x8 wv[HEAD_MAXC];
#pragma unroll
  for (int c = 0; c < HEAD_MAXC; ++c)      // all weight chunks in flight together
    wv[c] = *reinterpret_cast<const bf16x8*>(W + (long long)(c < Cout ? c : 0) * F + (ok ? chunk : 0) * 8);
  __syncthreads();
  float acc[DF_SPB][8];
#pragma unroll
  for (int s = 0; s < DF_SPB; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[s][j] = 0.f;
#pragma unroll
  for (int c = 0; c < HEAD_MAXC; ++c) {
    if (c < Cout) {
      float wf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) wf[j] = (float)wv[c][j];
#pragma unroll
      for (int s = 0; s < DF_SPB; ++s) {
        const float gsc = gs[s][c];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[s][j] += gsc * wf[j];
      }
    }
  }
  if (ok) {
#pragma unroll
    for (int s = 0; s < DF_SPB; ++s) {
      if (b0 + s < B) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[s][j];
        *reinterpret_cast<bf16x8*>(dFeat + (b0 + s) * (long long)F + chunk * 8) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------ dWeight
// partial[split][c][f] = sum_{b in split} dOut[b][c] * feat[b][f]; thread = 4 consecutive f,
// DW_SPS samples per split (their feat loads are all issued before the arithmetic)
constexpr int DW_SPS = 16;

__device__ __forceinline__ void head_bwd_dw_role(const bf16_t* __restrict__ feat,
                                                 const bf16_t* __restrict__ dOut,
                                                 float* __restrict__ partial, long long B, int F,
                                                 int Cout, int bx, int by, float (*gs)[HEAD_MAXC]) {
  const int tid = threadIdx.x;
  const int qi = bx * 256 + tid;
  const bool ok = qi < F / 4;
  const long long b0 = (long long)by * DW_SPS;
  for (int e = tid; e < DW_SPS * HEAD_MAXC; e += 256) {
    const int s = e / HEAD_MAXC, c = e % HEAD_MAXC;
    gs[s][c] = (b0 + s < B && c < Cout) ? (float)dOut[(b0 + s) * Cout + c] : 0.f;
  }
  uint2 raw[DW_SPS];
#pragma unroll
  for (int s = 0; s < DW_SPS; ++s) {
    const long long b = b0 + s < B ? b0 + s : B - 1;
    raw[s] = *reinterpret_cast<const uint2*>(feat + b * (long long)F + (ok ? qi : 0) * 4);
  }
  __syncthreads();
  float acc[HEAD_MAXC][4];
#pragma unroll
  for (int c = 0; c < HEAD_MAXC; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
#pragma unroll
  for (int s = 0; s < DW_SPS; ++s) {
    const float f0 = __uint_as_float(raw[s].x << 16), f1 = __uint_as_float(raw[s].x & 0xffff0000u);
    const float f2 = __uint_as_float(raw[s].y << 16), f3 = __uint_as_float(raw[s].y & 0xffff0000u);
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c) {
      if (c < Cout) {
        const float gsc = gs[s][c];   // zero for samples past B
        acc[c][0] += gsc * f0; acc[c][1] += gsc * f1; acc[c][2] += gsc * f2; acc[c][3] += gsc * f3;
      }
    }
  }
  if (ok) {
    float* dst = partial + (long long)by * Cout * F;
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c)
      if (c < Cout) *reinterpret_cast<float4*>(dst + (long long)c * F + qi * 4) =
          make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
  }
}

// One launch for both streaming passes of the backward: workgroups [0, n_dfeat) compute dFeat,
// the rest the dWeight partial slices (independent work, no reason to serialise two launches).
__global__ __launch_bounds__(256) void head_bwd_k(const bf16_t* __restrict__ feat,
                                                  const bf16_t* __restrict__ W,
                                                  const bf16_t* __restrict__ dOut,
                                                  bf16_t* __restrict__ dFeat,
                                                  float* __restrict__ partial, long long B, int F,
                                                  int Cout, int n_dfeat, int dfeat_gx, int dw_gx) {
  __shared__ float gs[DW_SPS][HEAD_MAXC];
  static_assert(DW_SPS >= DF_SPB, "shared staging buffer");
  const int bid = blockIdx.x;
  if (bid < n_dfeat) {
    head_bwd_dfeat_role(W, dOut, dFeat, B, F, Cout, bid % dfeat_gx, bid / dfeat_gx, gs);
  } else {
    const int r = bid - n_dfeat;
    head_bwd_dw_role(feat, dOut, partial, B, F, Cout, r % dw_gx, r / dw_gx, gs);
  }
}

// dW[i] = sum_split partial[split][i] (fixed order); the last workgroup produces dBias
__global__ __launch_bounds__(256) void head_bwd_dw_reduce_k(const float* __restrict__ partial,
                                                            const bf16_t* __restrict__ dOut,
                                                            bf16_t* __restrict__ dW,
                                                            bf16_t* __restrict__ dBias, long long B,
                                                            int F, int Cout, int splits) {
  const long long n = (long long)Cout * F;
  if (blockIdx.x + 1 < gridDim.x) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = 0.f;
      int k = 0;
      for (; k + 8 <= splits; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += partial[(k + u) * n + i];
      }
      for (; k < splits; ++k) a[0] += partial[k * n + i];
      dW[i] = (bf16_t)(((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
    }
  } else if (dBias) {
    // thread t < stride handles class t % Cout over rows t / Cout, t / Cout + rows_per_pass, ...
    __shared__ float red[256];
    const int rpp = 256 / Cout, stride = rpp * Cout;
    float s = 0.f;
    if ((int)threadIdx.x < stride) {
      const int c = threadIdx.x % Cout;
      for (long long b = threadIdx.x / Cout; b < B; b += rpp) s += (float)dOut[b * Cout + c];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < Cout) {
      float t = 0.f;
      for (int k = 0; k < rpp; ++k) t += red[k * Cout + threadIdx.x];
      dBias[threadIdx.x] = (bf16_t)t;
    }
  }
}

int dw_splits(long long B) { return (int)((B + DW_SPS - 1) / DW_SPS); }

bool head_ok(long long B, int F, int Cout, int dtype) {
  return dtype == DCTN_BF16 && B >= 1 && F >= 8 && F % 8 == 0 && Cout >= 1 && Cout <= HEAD_MAXC;
}

// ------------------------------------------------------------------------------------ generic head (any dtype, any F)
// float32 / float64 models (the reference's own arithmetic: new_runner.py:417 float32, its tests float64) and bf16
// feature counts that are not multiples of 8 (cfg3a: 23 x 23 x 6 = 3174) used to go to the library (`F.linear`, three
// Tensile GEMM launches in the backward).  Same three streaming passes as above in scalar form: S = storage type,
// A = accumulator (float; double for float64).  The work is tiny (B x F x Cout <= a few MFLOP) and bound by launch
// latency and one pass over `feat`; no matrix cores needed.
constexpr int HG_SPB = 4;   // samples per workgroup, forward (large batches; one per workgroup below 1024 samples)

// (round 5: batches below 1024 samples put one sample into a workgroup - four left 32 workgroups on 256 CUs at B = 128 -
// and the feature loop is unrolled by four: its 11 loads per turn were one memory round trip per turn, 46 us for cfg3a's
// 128 x 3174 features.)
template <typename S, typename A, int SPB>
__global__ __launch_bounds__(256) void head_fwd_gen_k(const S* __restrict__ feat, const S* __restrict__ W,
                                                      const S* __restrict__ bias, S* __restrict__ out, long long B, int F,
                                                      int Cout) {
  __shared__ A red[4][SPB * HEAD_MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long b0 = (long long)blockIdx.x * SPB;
  A acc[SPB][HEAD_MAXC];
#pragma unroll
  for (int s2 = 0; s2 < SPB; ++s2)
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c) acc[s2][c] = A(0);
  // (every load of a turn is issued before the first product: with `if (c < Cout)` around a row's load the compiler
  // branched per row and waited for each load on the spot - 11 dependent round trips per turn, 58 us for 128 x 5400
  // features.  Rows beyond Cout re-read the last row into accumulators nobody reads.)
#pragma unroll 2
  for (int f = tid; f < F; f += 256) {
    A xv[SPB], wr[HEAD_MAXC];
#pragma unroll
    for (int s2 = 0; s2 < SPB; ++s2) xv[s2] = (A)feat[(b0 + s2 < B ? b0 + s2 : B - 1) * (long long)F + f];
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c) wr[c] = (A)W[(long long)(c < Cout ? c : Cout - 1) * F + f];
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c)
#pragma unroll
      for (int s2 = 0; s2 < SPB; ++s2) acc[s2][c] += xv[s2] * wr[c];
  }
#pragma unroll
  for (int s2 = 0; s2 < SPB; ++s2)
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c) {
      if (c < Cout) {
        const A r = wave_reduce_sum(acc[s2][c]);
        if (lane == 0) red[wv][s2 * HEAD_MAXC + c] = r;
      }
    }
  __syncthreads();
  if (tid < SPB * HEAD_MAXC) {
    const int s2 = tid / HEAD_MAXC, c = tid % HEAD_MAXC;
    if (b0 + s2 < B && c < Cout)
      out[(b0 + s2) * Cout + c] = (S)(((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + (A)bias[c]);
  }
}

// backward, both streaming passes in one launch (as head_bwd_k): workgroups [0, n_dfeat) -> dFeat, the rest -> dW slices
template <typename S, typename A>
__global__ __launch_bounds__(256) void head_bwd_gen_k(const S* __restrict__ feat, const S* __restrict__ W,
                                                      const S* __restrict__ dOut, S* __restrict__ dFeat,
                                                      A* __restrict__ partial, long long B, int F, int Cout, int n_dfeat,
                                                      int gx) {
  __shared__ A gs[DW_SPS][HEAD_MAXC];
  const int tid = threadIdx.x, bid = blockIdx.x;
  if (bid < n_dfeat) {
    const int f = (bid % gx) * 256 + tid;
    const long long b0 = (long long)(bid / gx) * DF_SPB;
    if (tid < DF_SPB * HEAD_MAXC) {
      const int s2 = tid / HEAD_MAXC, c = tid % HEAD_MAXC;
      gs[s2][c] = (b0 + s2 < B && c < Cout) ? (A)dOut[(b0 + s2) * Cout + c] : A(0);
    }
    A w[HEAD_MAXC];
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c) {   // clamped indices + select: no branch between the loads
      const A ld = (A)W[(long long)(c < Cout ? c : Cout - 1) * F + (f < F ? f : F - 1)];
      w[c] = (c < Cout && f < F) ? ld : A(0);
    }
    __syncthreads();
    if (f < F) {
#pragma unroll
      for (int s2 = 0; s2 < DF_SPB; ++s2) {
        if (b0 + s2 < B) {
          A a = A(0);
#pragma unroll
          for (int c = 0; c < HEAD_MAXC; ++c) a += gs[s2][c] * w[c];
          dFeat[(b0 + s2) * (long long)F + f] = (S)a;
        }
      }
    }
  } else {
    const int r = bid - n_dfeat;
    const int f = (r % gx) * 256 + tid, by = r / gx;
    const long long b0 = (long long)by * DW_SPS;
    for (int e = tid; e < DW_SPS * HEAD_MAXC; e += 256) {
      const int s2 = e / HEAD_MAXC, c = e % HEAD_MAXC;
      gs[s2][c] = (b0 + s2 < B && c < Cout) ? (A)dOut[(b0 + s2) * Cout + c] : A(0);
    }
    A raw[DW_SPS];
#pragma unroll
    for (int s2 = 0; s2 < DW_SPS; ++s2) {
      const long long b = b0 + s2 < B ? b0 + s2 : B - 1;
      const A ld = (A)feat[b * (long long)F + (f < F ? f : F - 1)];
      raw[s2] = f < F ? ld : A(0);
    }
    __syncthreads();
    A acc[HEAD_MAXC];
#pragma unroll
    for (int c = 0; c < HEAD_MAXC; ++c) acc[c] = A(0);
#pragma unroll
    for (int s2 = 0; s2 < DW_SPS; ++s2)
#pragma unroll
      for (int c = 0; c < HEAD_MAXC; ++c) acc[c] += gs[s2][c] * raw[s2];   // gs is zero for samples past B
    if (f < F) {
      A* dst = partial + (long long)by * Cout * F;
#pragma unroll
      for (int c = 0; c < HEAD_MAXC; ++c)
        if (c < Cout) dst[(long long)c * F + f] = acc[c];
    }
  }
}

template <typename S, typename A>
__global__ __launch_bounds__(256) void head_bwd_gen_reduce_k(const A* __restrict__ partial, const S* __restrict__ dOut,
                                                             S* __restrict__ dW, S* __restrict__ dBias, long long B, int F,
                                                             int Cout, int splits) {
  const long long n = (long long)Cout * F;
  if (blockIdx.x + 1 < gridDim.x) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && dW) {
      A a[4] = {A(0), A(0), A(0), A(0)};
      int k = 0;
      for (; k + 4 <= splits; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += partial[(k + u) * n + i];
      }
      for (; k < splits; ++k) a[0] += partial[k * n + i];
      dW[i] = (S)((a[0] + a[1]) + (a[2] + a[3]));
    }
  } else if (dBias) {
    __shared__ A red[256];
    const int rpp = 256 / Cout, stride = rpp * Cout;
    A s2 = A(0);
    if ((int)threadIdx.x < stride) {
      const int c = threadIdx.x % Cout;
      for (long long b = threadIdx.x / Cout; b < B; b += rpp) s2 += (A)dOut[b * Cout + c];
    }
    red[threadIdx.x] = s2;
    __syncthreads();
    if ((int)threadIdx.x < Cout) {
      A t = A(0);
      for (int k = 0; k < rpp; ++k) t += red[k * Cout + threadIdx.x];
      dBias[threadIdx.x] = (S)t;
    }
  }
}

bool head_gen_ok(long long B, int F, int Cout, int dtype) {
  return (dtype == DCTN_BF16 || dtype == DCTN_F32 || dtype == DCTN_F64) && B >= 1 && F >= 1 && Cout >= 1 && Cout <= HEAD_MAXC;
}

template <typename S, typename A>
int head_fwd_gen(const void* feat, const void* weight, const void* bias, void* out, long long B, int F, int Cout, hipStream_t st) {
  if (B >= 1024)
    hipLaunchKernelGGL((head_fwd_gen_k<S, A, HG_SPB>), dim3((unsigned)((B + HG_SPB - 1) / HG_SPB)), dim3(256), 0, st, (const S*)feat,
                       (const S*)weight, (const S*)bias, (S*)out, B, F, Cout);
  else
    hipLaunchKernelGGL((head_fwd_gen_k<S, A, 1>), dim3((unsigned)B), dim3(256), 0, st, (const S*)feat, (const S*)weight,
                       (const S*)bias, (S*)out, B, F, Cout);
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

template <typename S, typename A>
int head_bwd_gen(const void* feat, const void* weight, const void* dOut, void* dFeat, void* dWeight, void* dBias, void* ws,
                 long long B, int F, int Cout, hipStream_t st) {
  const int splits = dw_splits(B);
  const int gx = (F + 255) / 256;
  const int n_dfeat = dFeat ? gx * (int)((B + DF_SPB - 1) / DF_SPB) : 0;
  const int n_dw = dWeight ? gx * splits : 0;
  if (n_dfeat + n_dw > 0) {
    hipLaunchKernelGGL((head_bwd_gen_k<S, A>), dim3((unsigned)(n_dfeat + n_dw)), dim3(256), 0, st, (const S*)feat,
                       (const S*)weight, (const S*)dOut, (S*)dFeat, (A*)ws, B, F, Cout, n_dfeat, gx);
    DCTN_CHECK_LAUNCH();
  }
  if (dWeight || dBias) {
    const long long n = (long long)Cout * F;
    hipLaunchKernelGGL((head_bwd_gen_reduce_k<S, A>), dim3((unsigned)(dWeight ? (n + 255) / 256 : 0) + 1), dim3(256), 0, st,
                       (const A*)ws, (const S*)dOut, (S*)dWeight, (S*)dBias, B, F, Cout, splits);
    DCTN_CHECK_LAUNCH();
  }
  return DCTN_OK;
}

}  // namespace

extern "C" {

int dctn_linear_head_fwd(const void* feat, const void* weight, const void* bias, void* out,
                         int64_t B, int F, int Cout, int dtype, void* stream) {
  if (!feat || !weight || !bias || !out) return DCTN_ERR_NULL;
  if (B < 1 || F < 1 || Cout < 1) return DCTN_ERR_BAD_SHAPE;
  if (!head_ok(B, F, Cout, dtype) || ((uintptr_t)feat % 16) || ((uintptr_t)weight % 16)) {
    // float32 / float64, feature counts that are not multiples of 8, unaligned views: the scalar kernels
    if (!head_gen_ok(B, F, Cout, dtype)) return DCTN_ERR_UNSUPPORTED;
    int rc;
    if (dtype == DCTN_F64) rc = head_fwd_gen<double, double>(feat, weight, bias, out, B, F, Cout, (hipStream_t)stream);
    else if (dtype == DCTN_F32) rc = head_fwd_gen<float, float>(feat, weight, bias, out, B, F, Cout, (hipStream_t)stream);
    else rc = head_fwd_gen<bf16_t, float>(feat, weight, bias, out, B, F, Cout, (hipStream_t)stream);
    if (rc == DCTN_OK) dctn_set_last_kernel("linear_head_fwd_generic");
    return rc;
  }
  if ((B + 15) / 16 >= 256)
    hipLaunchKernelGGL(head_fwd_k<16>, dim3((unsigned)((B + 15) / 16)), dim3(512), 0, (hipStream_t)stream,
                       (const bf16_t*)feat, (const bf16_t*)weight, (const bf16_t*)bias, (bf16_t*)out,
                       (long long)B, F, Cout);
  else
    hipLaunchKernelGGL(head_fwd_k<4>, dim3((unsigned)((B + 3) / 4)), dim3(512), 0, (hipStream_t)stream,
                       (const bf16_t*)feat, (const bf16_t*)weight, (const bf16_t*)bias, (bf16_t*)out,
                       (long long)B, F, Cout);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("linear_head_fwd_mfma");
  return DCTN_OK;
}

size_t dctn_linear_head_bwd_workspace_bytes(int64_t B, int F, int Cout, int dtype) {
  if (!head_gen_ok(B, F, Cout, dtype)) return 0;
  return (size_t)dw_splits(B) * Cout * F * (dtype == DCTN_F64 ? sizeof(double) : sizeof(float)) + 256;
}

int dctn_linear_head_bwd(const void* feat, const void* weight, const void* dOut, void* dFeat,
                         void* dWeight, void* dBias, void* workspace, size_t workspace_bytes,
                         int64_t B, int F, int Cout, int dtype, void* stream) {
  if (!feat || !weight || !dOut) return DCTN_ERR_NULL;
  if (B < 1 || F < 1 || Cout < 1) return DCTN_ERR_BAD_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int splits = dw_splits(B);
  if (!head_ok(B, F, Cout, dtype) || ((uintptr_t)feat % 16) || ((uintptr_t)weight % 16) || (dFeat && ((uintptr_t)dFeat % 16))) {
    if (!head_gen_ok(B, F, Cout, dtype)) return DCTN_ERR_UNSUPPORTED;
    const size_t asz = dtype == DCTN_F64 ? sizeof(double) : sizeof(float);
    if (dWeight && (!workspace || workspace_bytes < (size_t)splits * Cout * F * asz || ((uintptr_t)workspace % 8)))
      return DCTN_ERR_WORKSPACE;
    int rc;
    if (dtype == DCTN_F64) rc = head_bwd_gen<double, double>(feat, weight, dOut, dFeat, dWeight, dBias, workspace, B, F, Cout, st);
    else if (dtype == DCTN_F32) rc = head_bwd_gen<float, float>(feat, weight, dOut, dFeat, dWeight, dBias, workspace, B, F, Cout, st);
    else rc = head_bwd_gen<bf16_t, float>(feat, weight, dOut, dFeat, dWeight, dBias, workspace, B, F, Cout, st);
    if (rc == DCTN_OK) dctn_set_last_kernel("linear_head_bwd_generic");
    return rc;
  }
  if (dWeight && (!workspace || workspace_bytes < (size_t)splits * Cout * F * sizeof(float)))
    return DCTN_ERR_WORKSPACE;
  const int dfeat_gx = (F / 8 + 255) / 256, dfeat_gy = (int)((B + DF_SPB - 1) / DF_SPB);
  const int dw_gx = (F / 4 + 255) / 256;
  const int n_dfeat = dFeat ? dfeat_gx * dfeat_gy : 0;
  const int n_dw = dWeight ? dw_gx * splits : 0;
  if (n_dfeat + n_dw > 0) {
    hipLaunchKernelGGL(head_bwd_k, dim3((unsigned)(n_dfeat + n_dw)), dim3(256), 0, st, (const bf16_t*)feat,
                       (const bf16_t*)weight, (const bf16_t*)dOut, (bf16_t*)dFeat, (float*)workspace,
                       (long long)B, F, Cout, n_dfeat, dfeat_gx, dw_gx);
    DCTN_CHECK_LAUNCH();
  }
  if (dWeight) {
    const long long n = (long long)Cout * F;
    hipLaunchKernelGGL(head_bwd_dw_reduce_k, dim3((unsigned)((n + 255) / 256) + 1), dim3(256), 0, st,
                       (const float*)workspace, (const bf16_t*)dOut, (bf16_t*)dWeight, (bf16_t*)dBias,
                       (long long)B, F, Cout, splits);
    DCTN_CHECK_LAUNCH();
  }
  dctn_set_last_kernel("linear_head_bwd");
  return DCTN_OK;
}

}  // extern "C"
