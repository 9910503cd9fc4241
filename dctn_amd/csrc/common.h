// Shared helpers for the dctn_amd HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dctn_amd.h"

typedef __bf16 bf16_t;

template <typename S> struct AccOf { typedef float type; };
template <> struct AccOf<double> { typedef double type; };

#define DCTN_WAVE 64
#define DCTN_LDS_BUDGET (150 * 1024)  // bytes of the 160 KiB LDS a single workgroup may claim here

#define DCTN_CHECK_LAUNCH()                                   \
  do {                                                        \
    if (hipGetLastError() != hipSuccess) return DCTN_ERR_LAUNCH; \
  } while (0)

// name of the kernel family the last call dispatched to
void dctn_set_last_kernel(const char* name);

// Zero `bytes` bytes on the stream with a kernel.  Not hipMemsetAsync: recorded into a
// torch HIP graph, a memset node zero-filled on the first replay and wrote garbage on the later ones
// (tools/memset_capture_check.py, ROCm 7.2 + torch 2.10; the same node replays correctly from a plain HIP program,
// tools/probes/memset_graph.hip) - accumulators that start from zero must not depend on it.
int dctn_zero_async(void* ptr, size_t bytes, hipStream_t st);

static inline long long ipow_ll(long long b, int e) {
  long long r = 1;
  for (int i = 0; i < e; ++i) r *= b;
  return r;
}

static inline size_t dtype_size(int dtype) {
  return dtype == DCTN_F64 ? 8 : (dtype == DCTN_F32 ? 4 : 2);
}

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// -------------------------------------------------------------------------------- EPS shapes
struct EpsP {
  int C, B, H, W, Q, K, O, N, Ho, Wo;
  long long Wn;    // B*Ho*Wo windows
  long long R;     // Q^N core rows
  long long s[5];  // element strides of x
  int m, LO, NH;   // low split: last m factors, LO = Q^m, NH = N-m high factors
  long long HI;    // Q^NH
  int bits;        // bits per packed digit in the dCore kernel
  int opts;        // DCTN_OPT_* flags of the call's policy argument (0 from the workspace queries' defaults)
};

int eps_fill_params(EpsP& p, const int64_t x_strides[5], int C, int B, int H, int W, int Q, int K,
                    int O, int policy = 0);

// generic (any shape, f32/f64/bf16-storage) kernels — eps_generic.hip
int eps_fwd_generic(const void* x, const void* core, void* out, EpsP p, int dtype, hipStream_t st);
size_t eps_bwd_generic_workspace(const EpsP& p, int dtype, int need_dx, int need_dcore);
int eps_bwd_generic(const void* x, const void* core, const void* dY, void* dX, void* dCore,
                    void* ws, size_t ws_bytes, EpsP p, int dtype, hipStream_t st);

int eps_gather_dx_launch(const void* gxw, void* dX, const EpsP& p, int dtype, hipStream_t st);

// MFMA family "bigcore" (exact f32, LDS-streamed core) — eps_bigcore.hip
size_t eps_fwd_bigcore_workspace(const EpsP& p, int dtype, int precision);
// zsave (optional): room of eps_bigcore_saved_bytes() for the forward GEMM result Z, which eps_bwd_dx_bigcore then
// reads instead of running its third GEMM (0 bytes: the shape keeps nothing)
size_t eps_bigcore_saved_bytes(const EpsP& p, int dtype, int precision);
int eps_fwd_bigcore(const void* x, const void* core, void* out, void* ws, size_t ws_bytes,
                    const EpsP& p, int dtype, int precision, hipStream_t st, void* zsave = nullptr);
size_t eps_bwd_dcore_bigcore_workspace(const EpsP& p, int dtype, int precision);
int eps_bwd_dcore_bigcore(const void* x, const void* dY, void* dCore, const EpsP& p, int dtype,
                          int precision, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0);
size_t eps_bwd_dfactor_bigcore_workspace(const EpsP& p, int dtype, int precision);
int eps_bwd_dx_bigcore(const void* x, const void* core, const void* dY, void* dX, void* ws,
                       size_t ws_bytes, const EpsP& p, int dtype, int precision, hipStream_t st,
                       const void* zsaved = nullptr, size_t zsaved_bytes = 0);

// register-resident exact-float32 family for Q = 2, N in {8, 9}, O <= 4 (cfg2 in the reference's own dtype) — eps_q2f32.hip
bool eps_q2f32_covers(const EpsP& p, int dtype, int precision);
int eps_fwd_q2f32(const void* x, const void* core, void* out, const EpsP& p, int dtype, int precision, hipStream_t st);
size_t eps_bwd_q2f32_workspace(const EpsP& p, int dtype, int precision);
int eps_bwd_q2f32(const void* x, const void* dY, void* dCore, void* ws, size_t ws_bytes, const EpsP& p, int dtype,
                  int precision, hipStream_t st);
int eps_head_fwd_q2f32(const void* x, const void* core, const void* head_w, const void* bias, void* feat, void* logits,
                       const EpsP& p, int Cout, int dtype, int precision, hipStream_t st);
size_t eps_head_bwd_q2f32_workspace(const EpsP& p, int Cout, int dtype, int precision);
int eps_head_bwd_q2f32(const void* x, const void* feat, const void* dLogits, const void* head_w, void* dCore, void* dW,
                       void* dBias, void* ws, size_t ws_bytes, const EpsP& p, int Cout, int dtype, int precision,
                       hipStream_t st);

// which family the forward of a shape dispatches to (dctn_eps_family)
bool eps_mfma_covers(const EpsP& p, int dtype, int precision);
bool eps_bigcore_covers(const EpsP& p, int dtype, int precision);

// two-halves GEMM path on the 16x16x4 matrix instructions — eps_halves.hip: float64 (v_mfma_f64_16x16x4_f64),
// and float32 (v_mfma_f32_16x16x4_f32) for shapes the other float32 families leave (odd Q, ...)
bool eps_halves_wanted(const EpsP& p, int dtype);
size_t eps_fwd_halves_workspace(const EpsP& p, int dtype);
// saved (optional): room of eps_halves_saved_bytes() for the forward's halves and GEMM result, which eps_bwd_halves
// then reads instead of rebuilding them
size_t eps_halves_saved_bytes(const EpsP& p, int dtype);
int eps_fwd_halves(const void* x, const void* core, void* out, void* ws, size_t ws_bytes, const EpsP& p, int dtype,
                   hipStream_t st, void* saved = nullptr);
size_t eps_bwd_halves_workspace(const EpsP& p, int dtype, int need_dx, int need_dcore);
int eps_bwd_halves(const void* x, const void* core, const void* dY, void* dX, void* dCore, void* ws, size_t ws_bytes,
                   const EpsP& p, int dtype, hipStream_t st, const void* saved = nullptr, size_t saved_bytes = 0);

// MFMA kernels for power-of-two Q — eps_mfma.hip.  Return DCTN_ERR_UNSUPPORTED when the shape
// is outside the family so that the dispatcher can take the generic kernels.
// stats (optional): float64 {sum y, sum y^2}, accumulated; out may then be NULL (nothing stored)
int eps_fwd_mfma(const void* x, const void* core, void* out, const EpsP& p, int dtype,
                 int precision, hipStream_t st, double* stats = nullptr);
// {sum y, sum y^2} of n stored values, accumulated into float64 stats[2] — window_stats.hip
int eps_out_stats(const void* y, long long n, int dtype, double* stats, hipStream_t st);
size_t eps_bwd_mfma_workspace(const EpsP& p, int dtype, int precision, int need_dx, int need_dcore);
int eps_bwd_mfma(const void* x, const void* core, const void* dY, void* dX, void* dCore, void* ws,
                 size_t ws_bytes, const EpsP& p, int dtype, int precision, hipStream_t st);
int eps_head_fwd_mfma(const void* x, const void* core, const void* head_w, const void* bias, void* feat, void* logits,
                      const EpsP& p, int Cout, int dtype, int precision, hipStream_t st);
size_t eps_head_bwd_mfma_workspace(const EpsP& p, int Cout, int dtype, int precision);
int eps_head_bwd_mfma(const void* x, const void* feat, const void* dLogits, const void* head_w, void* dCore,
                      void* dW, void* dBias, void* ws, size_t ws_bytes, const EpsP& p, int Cout, int dtype,
                      int precision, hipStream_t st);

// MFMA ConvSBS sweep (open chain, uniform bond) — convsbs_mfma.hip
// save_states (optional): room of convsbs_saved_states_bytes(...) for the forward states a following backward takes over
size_t convsbs_saved_states_bytes(int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                                  int C, int B, int H, int W, int q, int dtype);
int convsbs_fwd_mfma(const void* x, const int64_t xs[5], const void* const* cores, void* out, int n,
                     const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                     int C, int B, int H, int W, int q, int dtype, hipStream_t st, float* save_states = nullptr);
int convsbs_bwd_mfma(const void* x, const int64_t xs[5], const void* const* cores, const void* dY,
                     float* states, float* gxw, float* const* dcores, int n, const int* out_sizes,
                     const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W,
                     int q, int dtype, hipStream_t st, float* partials = nullptr, size_t partial_bytes = 0,
                     const float* saved_states = nullptr);
// Band-owning backward for bonds 5..16 (two roles per SIMD, nothing kept by the forward) - convsbs_band.hip.  Strings it
// covers keep no forward states (convsbs_saved_states_bytes returns 0 for them); `ws` holds the per-workgroup dCore
// records and the partial sums of the pixel rows two bands share (convsbs_band_bwd_workspace; 0 = outside the family).
bool convsbs_band_covers(int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B,
                         int H, int W, int q, int dtype);
size_t convsbs_band_bwd_workspace(int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w, int C,
                                  int B, int H, int W, int q, int dtype);
int convsbs_fwd_band(const void* x, const int64_t xs[5], const void* const* cores, void* out, int n, const int* out_sizes,
                     const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q, int dtype,
                     hipStream_t st);
int convsbs_bwd_band(const void* x, const int64_t xs[5], const void* const* cores, const void* dY, void* dX,
                     float* const* dcores, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h,
                     const int* pos_w, int C, int B, int H, int W, int q, int dtype, hipStream_t st, void* ws, size_t ws_bytes);
// room for the per-workgroup partial-gradient records of the MFMA backward (deterministic dCore)
constexpr int SBS_MAX_PARTIAL_RECORDS = 2048;

// Register-resident sweep for small bonds (every bond <= 4, float32 open chains, at most one two-valued core,
// q^C <= 4) - convsbs_reg.hip.  The backward writes dX itself (no per-window gradients, no gather launch) and needs
// `ws` only for its per-workgroup dCore records (convsbs_reg_bwd_workspace; 0 = the string is outside the family).
size_t convsbs_reg_bwd_workspace(int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                                 int C, int B, int H, int W, int q, int dtype);
int convsbs_fwd_reg(const void* x, const int64_t xs[5], const void* const* cores, void* out, int n, const int* out_sizes,
                    const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W, int q, int dtype,
                    hipStream_t st);
int convsbs_bwd_reg(const void* x, const int64_t xs[5], const void* const* cores, const void* dY, void* dX,
                    float* const* dcores, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h,
                    const int* pos_w, int C, int B, int H, int W, int q, int dtype, hipStream_t st, void* ws, size_t ws_bytes);
// several uniform strings of one layer in one launch each way (DCTN_ERR_UNSUPPORTED: run them one by one)
// The same for strings of the band family (bonds 5..16): blockIdx.y = string, one tail kernel for all strings.
size_t convsbs_many_band_bwd_workspace(int ns, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                                       int C, int B, int H, int W, int q, int dtype);
int convsbs_many_fwd_band(const void* x, const int64_t xs[5], const void* const* cores, void* const* outs, int ns, int n,
                          const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H, int W,
                          int q, int dtype, hipStream_t st);
int convsbs_many_bwd_band(const void* x, const int64_t xs[5], const void* const* cores, const void* const* dYs, void* dX,
                          float* const* dcores, int ns, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h,
                          const int* pos_w, int C, int B, int H, int W, int q, int dtype, hipStream_t st, void* ws, size_t ws_bytes);
size_t convsbs_many_reg_bwd_workspace(int ns, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w,
                                      int C, int B, int H, int W, int q, int dtype);
int convsbs_many_fwd_reg(const void* x, const int64_t xs[5], const void* const* cores, void* const* outs, int ns, int n,
                         const int* out_sizes, const int* bond_sizes, const int* pos_h, const int* pos_w, int C, int B, int H,
                         int W, int q, int dtype, hipStream_t st);
int convsbs_many_bwd_reg(const void* x, const int64_t xs[5], const void* const* cores, const void* const* dYs, void* dX,
                         float* const* dcores, int ns, int n, const int* out_sizes, const int* bond_sizes, const int* pos_h,
                         const int* pos_w, int C, int B, int H, int W, int q, int dtype, hipStream_t st, void* ws, size_t ws_bytes);
