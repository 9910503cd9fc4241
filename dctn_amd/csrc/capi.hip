// C-ABI entry points (include/dctn_amd.h): argument validation + dispatch to kernel families.
#include "common.h"

#include <stdlib.h>

#include <atomic>

// The library's only mutable global: a diagnostic pointer to a string literal naming the kernel family of the last
// successful call (process-wide, relaxed atomic, last writer wins: autograd runs backward on its own thread, so a
// thread-local would hide the backward's kernels from the caller).  No entry point reads it to decide anything.
static std::atomic<const char*> g_last_kernel{"none"};
void dctn_set_last_kernel(const char* name) { g_last_kernel.store(name, std::memory_order_relaxed); }

namespace {
__global__ __launch_bounds__(256) void dctn_zero_k(unsigned* __restrict__ p, size_t words) {
  const size_t n4 = words / 4;
  uint4* p4 = reinterpret_cast<uint4*>(p);
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p4[i] = uint4{0u, 0u, 0u, 0u};
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) p[i] = 0u;
}
__global__ __launch_bounds__(256) void dctn_zero_words_k(unsigned* __restrict__ p, size_t words) {   // 4-byte aligned only
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) p[i] = 0u;
}
__global__ __launch_bounds__(256) void dctn_zero_bytes_k(unsigned char* __restrict__ p, size_t bytes) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < bytes; i += stride) p[i] = 0;
}
}  // namespace

int dctn_zero_async(void* ptr, size_t bytes, hipStream_t st) {
  if (bytes == 0) return DCTN_OK;
  if (!ptr) return DCTN_ERR_NULL;
  if ((bytes & 3) || ((uintptr_t)ptr & 3)) {   // odd sizes (bf16 tails): byte by byte
    size_t bb = (bytes + 255) / 256;
    if (bb > 2048) bb = 2048;
    hipLaunchKernelGGL(dctn_zero_bytes_k, dim3((unsigned)bb), dim3(256), 0, st, (unsigned char*)ptr, bytes);
    return hipGetLastError() == hipSuccess ? DCTN_OK : DCTN_ERR_LAUNCH;
  }
  const size_t words = bytes / 4;
  size_t blocks = (words / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  if (((uintptr_t)ptr & 15) == 0)
    hipLaunchKernelGGL(dctn_zero_k, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned*)ptr, words);
  else
    hipLaunchKernelGGL(dctn_zero_words_k, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned*)ptr, words);
  return hipGetLastError() == hipSuccess ? DCTN_OK : DCTN_ERR_LAUNCH;
}

extern "C" {

int dctn_version(void) { return 500; }   // round 5: register-resident exact-f32 EPS family (eps_q2f32.hip), fused head both ways in float32

const char* dctn_last_kernel(void) { return g_last_kernel.load(std::memory_order_relaxed); }

const char* dctn_strerror(int code) {
  switch (code) {
    case DCTN_OK: return "ok";
    case DCTN_ERR_BAD_SHAPE: return "inconsistent shape";
    case DCTN_ERR_BAD_DTYPE: return "unknown dtype code";
    case DCTN_ERR_UNSUPPORTED: return "shape/dtype not covered by any kernel of this build";
    case DCTN_ERR_WORKSPACE: return "workspace missing or too small";
    case DCTN_ERR_LAUNCH: return "HIP launch error";
    case DCTN_ERR_NULL: return "required pointer is NULL";
  }
  return "unknown error";
}

// DCTN_OPT_F32_PREFER_HALVES sends float32 shapes that both exact families cover to the two-halves path
static bool f32_prefers_halves(const EpsP& p, int dtype) {
  return (p.opts & DCTN_OPT_F32_PREFER_HALVES) && dtype == DCTN_F32 && eps_halves_wanted(p, dtype);
}

static bool dtype_ok(int dtype) { return dtype == DCTN_F32 || dtype == DCTN_F64 || dtype == DCTN_BF16; }

size_t dctn_eps_fwd_workspace_bytes(int C, int B, int H, int W, int Q, int K, int O, int dtype,
                                    int policy) {
  EpsP p;
  const int precision = policy & DCTN_PREC_MASK;
  const int64_t dummy[5] = {0, 0, 0, 0, 1};
  if (eps_fill_params(p, dummy, C, B, H, W, Q, K, O, policy) != DCTN_OK) return 0;
  const size_t a = eps_fwd_bigcore_workspace(p, dtype, precision), b = eps_fwd_halves_workspace(p, dtype);
  return (a > b ? a : b) + 256;
}

int dctn_eps_family(int C, int B, int H, int W, int Q, int K, int O, int dtype, int policy) {
  EpsP p;
  const int precision = policy & DCTN_PREC_MASK;
  const int64_t dummy[5] = {0, 0, 0, 0, 1};
  if (!dtype_ok(dtype) || eps_fill_params(p, dummy, C, B, H, W, Q, K, O, policy) != DCTN_OK) return -1;
  if (p.opts & DCTN_OPT_GENERIC_KERNELS) return DCTN_EPS_FAMILY_GENERIC;
  if (eps_mfma_covers(p, dtype, precision)) return DCTN_EPS_FAMILY_Q2REG;
  if (eps_q2f32_covers(p, dtype, precision)) return DCTN_EPS_FAMILY_Q2REG_F32;
  if (eps_bigcore_covers(p, dtype, precision) && !f32_prefers_halves(p, dtype)) return DCTN_EPS_FAMILY_BIGCORE_F32;
  if (eps_halves_wanted(p, dtype)) return DCTN_EPS_FAMILY_HALVES;
  return DCTN_EPS_FAMILY_GENERIC;
}

// forward with an optional buffer for what the backward would otherwise recompute; *kept = 1 when it was written
static int eps_fwd_impl(const void* x, const int64_t x_strides[5], const void* core, void* out, void* saved,
                        size_t saved_bytes, int* kept, void* workspace, size_t workspace_bytes, int C, int B, int H, int W,
                        int Q, int K, int O, int dtype, int policy, void* stream) {
  if (kept) *kept = 0;
  if (!x || !core || !out || !x_strides) return DCTN_ERR_NULL;
  if (!dtype_ok(dtype)) return DCTN_ERR_BAD_DTYPE;
  EpsP p;
  const int precision = policy & DCTN_PREC_MASK;
  int rc = eps_fill_params(p, x_strides, C, B, H, W, Q, K, O, policy);
  if (rc != DCTN_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (p.opts & DCTN_OPT_GENERIC_KERNELS) return eps_fwd_generic(x, core, out, p, dtype, st);
  rc = eps_fwd_mfma(x, core, out, p, dtype, precision, st);
  if (rc != DCTN_ERR_UNSUPPORTED) return rc;
  // the register-resident exact-f32 family keeps nothing for a backward: a training forward whose input needs a gradient
  // (the caller brought a buffer for the GEMM result) stays on the large-core family, whose backward reads that buffer
  const bool wants_saved = saved && saved_bytes > 0 && saved_bytes >= eps_bigcore_saved_bytes(p, dtype, precision) &&
                           eps_bigcore_saved_bytes(p, dtype, precision) > 0 && !f32_prefers_halves(p, dtype);
  if (!wants_saved) {
    rc = eps_fwd_q2f32(x, core, out, p, dtype, precision, st);
    if (rc != DCTN_ERR_UNSUPPORTED) return rc;
  }
  if (!f32_prefers_halves(p, dtype)) {
    const size_t zb = saved ? eps_bigcore_saved_bytes(p, dtype, precision) : 0;
    const bool keep = zb > 0 && saved_bytes >= zb;
    rc = eps_fwd_bigcore(x, core, out, workspace, workspace_bytes, p, dtype, precision, st, keep ? saved : nullptr);
    if (rc == DCTN_OK && keep && kept) *kept = 1;
    if (rc != DCTN_ERR_UNSUPPORTED) return rc;
  }
  {
    // the two-halves family may keep the buffer only where the backward will read it as ITS layout (P0 | P1 | Z): the
    // same routing as dctn_eps_saved_bytes and eps_bwd_impl.  A shape the large-core family claims but then declines
    // falls through to here with nothing kept (its backward would read the buffer as row-quad-major Z).
    const bool halves_route = dtype == DCTN_F64 || !eps_bigcore_covers(p, dtype, precision) || f32_prefers_halves(p, dtype);
    const size_t hb = (saved && halves_route) ? eps_halves_saved_bytes(p, dtype) : 0;
    const bool keep = hb > 0 && saved_bytes >= hb;
    rc = eps_fwd_halves(x, core, out, workspace, workspace_bytes, p, dtype, st, keep ? saved : nullptr);
    if (rc == DCTN_OK && keep && kept) *kept = 1;
    if (rc != DCTN_ERR_UNSUPPORTED && rc != DCTN_ERR_WORKSPACE) return rc;
  }
  return eps_fwd_generic(x, core, out, p, dtype, st);
}

int dctn_eps_fwd(const void* x, const int64_t x_strides[5], const void* core, void* out,
                 void* workspace, size_t workspace_bytes, int C, int B, int H, int W, int Q, int K,
                 int O, int dtype, int policy, void* stream) {
  return eps_fwd_impl(x, x_strides, core, out, nullptr, 0, nullptr, workspace, workspace_bytes, C, B, H, W, Q, K, O, dtype,
                      policy, stream);
}

size_t dctn_eps_saved_bytes(int C, int B, int H, int W, int Q, int K, int O, int dtype, int policy) {
  EpsP p;
  const int precision = policy & DCTN_PREC_MASK;
  const int64_t dummy[5] = {0, 0, 0, 0, 1};
  if (!dtype_ok(dtype) || eps_fill_params(p, dummy, C, B, H, W, Q, K, O, policy) != DCTN_OK) return 0;
  if ((p.opts & DCTN_OPT_GENERIC_KERNELS) || eps_mfma_covers(p, dtype, precision)) return 0;
  if (eps_bigcore_covers(p, dtype, precision) && !f32_prefers_halves(p, dtype)) return eps_bigcore_saved_bytes(p, dtype, precision);
  return eps_halves_saved_bytes(p, dtype);
}

int dctn_eps_fwd_save(const void* x, const int64_t x_strides[5], const void* core, void* out, void* saved,
                      size_t saved_bytes, void* workspace, size_t workspace_bytes, int C, int B, int H, int W, int Q,
                      int K, int O, int dtype, int policy, void* stream) {
  int kept = 0;
  const int rc = eps_fwd_impl(x, x_strides, core, out, saved, saved_bytes, &kept, workspace, workspace_bytes, C, B, H, W, Q,
                              K, O, dtype, policy, stream);
  return rc != DCTN_OK ? rc : (kept ? DCTN_SAVED : DCTN_OK);
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t dctn_eps_fwd_stats_workspace_bytes(int C, int B, int H, int W, int Q, int K, int O, int dtype, int policy) {
  EpsP p;
  const int precision = policy & DCTN_PREC_MASK;
  const int64_t dummy[5] = {0, 0, 0, 0, 1};
  if (eps_fill_params(p, dummy, C, B, H, W, Q, K, O, policy) != DCTN_OK) return 0;
  if (eps_mfma_covers(p, dtype, precision)) return 256;   // in-kernel epilogue: no scratch
  return align256((size_t)p.Wn * O * dtype_size(dtype)) + dctn_eps_fwd_workspace_bytes(C, B, H, W, Q, K, O, dtype, policy) + 256;
}

int dctn_eps_fwd_stats(const void* x, const int64_t x_strides[5], const void* core, void* stats, void* workspace,
                       size_t workspace_bytes, int C, int B, int H, int W, int Q, int K, int O, int dtype, int policy,
                       void* stream) {
  if (!x || !core || !stats || !x_strides) return DCTN_ERR_NULL;
  if (!dtype_ok(dtype)) return DCTN_ERR_BAD_DTYPE;
  EpsP p;
  const int precision = policy & DCTN_PREC_MASK;
  int rc = eps_fill_params(p, x_strides, C, B, H, W, Q, K, O, policy);
  if (rc != DCTN_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  // register-resident family: the statistics are an epilogue of the forward kernel, nothing is stored
  rc = eps_fwd_mfma(x, core, nullptr, p, dtype, precision, st, (double*)stats);
  if (rc != DCTN_ERR_UNSUPPORTED) return rc;
  // other families: the slice's output goes to scratch (cache resident for the slice sizes of eps.py:126-137) and one
  // reduction pass folds it into the running sums
  const size_t out_bytes = align256((size_t)p.Wn * O * dtype_size(dtype));
  if (!workspace || workspace_bytes < out_bytes) return DCTN_ERR_WORKSPACE;
  unsigned char* ws = (unsigned char*)workspace;
  rc = dctn_eps_fwd(x, x_strides, core, ws, ws + out_bytes, workspace_bytes - out_bytes, C, B, H, W, Q, K, O, dtype, policy,
                    stream);
  if (rc != DCTN_OK) return rc;
  return eps_out_stats(ws, p.Wn * O, dtype, (double*)stats, st);
}

size_t dctn_eps_bwd_workspace_bytes(int C, int B, int H, int W, int Q, int K, int O, int dtype,
                                    int policy, int need_dx, int need_dcore) {
  EpsP p;
  const int precision = policy & DCTN_PREC_MASK;
  const int64_t dummy[5] = {0, 0, 0, 0, 1};
  if (eps_fill_params(p, dummy, C, B, H, W, Q, K, O, policy) != DCTN_OK) return 0;
  const size_t a = align256(eps_bwd_mfma_workspace(p, dtype, precision, need_dx, need_dcore)) +
                   align256(need_dcore ? eps_bwd_q2f32_workspace(p, dtype, precision) : 0);
  size_t b = eps_bwd_generic_workspace(p, dtype, need_dx, need_dcore);
  const size_t c = need_dx ? eps_bwd_dfactor_bigcore_workspace(p, dtype, precision) : 0;
  if (c > b) b = c;
  const size_t c2 = need_dcore ? eps_bwd_dcore_bigcore_workspace(p, dtype, precision) : 0;
  if (c2 > b) b = c2;
  const size_t d = eps_bwd_halves_workspace(p, dtype, need_dx, need_dcore);
  if (d > b) b = d;
  return a + b + 256;
}

int dctn_eps_head_fwd(const void* x, const int64_t x_strides[5], const void* core, const void* head_weight,
                      const void* head_bias, void* features, void* logits, int C, int B, int H, int W, int Q, int K, int O,
                      int Cout, int dtype, int policy, void* stream) {
  if (!x || !core || !head_weight || !head_bias || !features || !logits || !x_strides) return DCTN_ERR_NULL;
  if (!dtype_ok(dtype)) return DCTN_ERR_BAD_DTYPE;
  if (Cout < 1) return DCTN_ERR_BAD_SHAPE;
  EpsP p;
  const int precision = policy & DCTN_PREC_MASK;
  const int rc = eps_fill_params(p, x_strides, C, B, H, W, Q, K, O, policy);
  if (rc != DCTN_OK) return rc;
  const int rcm = eps_head_fwd_mfma(x, core, head_weight, head_bias, features, logits, p, Cout, dtype, precision, (hipStream_t)stream);
  if (rcm != DCTN_ERR_UNSUPPORTED) return rcm;
  return eps_head_fwd_q2f32(x, core, head_weight, head_bias, features, logits, p, Cout, dtype, precision, (hipStream_t)stream);
}

size_t dctn_eps_head_bwd_workspace_bytes(int C, int B, int H, int W, int Q, int K, int O, int Cout, int dtype,
                                         int policy) {
  EpsP p;
  const int precision = policy & DCTN_PREC_MASK;
  const int64_t dummy[5] = {0, 0, 0, 0, 1};
  if (eps_fill_params(p, dummy, C, B, H, W, Q, K, O, policy) != DCTN_OK) return 0;
  const size_t a = eps_head_bwd_mfma_workspace(p, Cout, dtype, precision), b = eps_head_bwd_q2f32_workspace(p, Cout, dtype, precision);
  return (a > b ? a : b) + 256;
}

int dctn_eps_head_bwd(const void* x, const int64_t x_strides[5], const void* features, const void* dLogits,
                      const void* head_weight, void* dCore, void* dWeight, void* dBias, void* workspace,
                      size_t workspace_bytes, int C, int B, int H, int W, int Q, int K, int O, int Cout,
                      int dtype, int policy, void* stream) {
  if (!x || !features || !dLogits || !head_weight || !dCore || !x_strides) return DCTN_ERR_NULL;
  if (!dtype_ok(dtype)) return DCTN_ERR_BAD_DTYPE;
  if (Cout < 1) return DCTN_ERR_BAD_SHAPE;
  EpsP p;
  const int precision = policy & DCTN_PREC_MASK;
  const int rc = eps_fill_params(p, x_strides, C, B, H, W, Q, K, O, policy);
  if (rc != DCTN_OK) return rc;
  const int rcm = eps_head_bwd_mfma(x, features, dLogits, head_weight, dCore, dWeight, dBias, workspace, workspace_bytes, p,
                                    Cout, dtype, precision, (hipStream_t)stream);
  if (rcm != DCTN_ERR_UNSUPPORTED) return rcm;
  return eps_head_bwd_q2f32(x, features, dLogits, head_weight, dCore, dWeight, dBias, workspace, workspace_bytes, p, Cout,
                            dtype, precision, (hipStream_t)stream);
}

static int eps_bwd_impl(const void* x, const int64_t x_strides[5], const void* core, const void* dY, const void* saved,
                        size_t saved_bytes, void* dX, void* dCore, void* workspace, size_t workspace_bytes, int C, int B,
                        int H, int W, int Q, int K, int O, int dtype, int policy, void* stream) {
  if (!x || !core || !dY || !x_strides) return DCTN_ERR_NULL;
  if (!dtype_ok(dtype)) return DCTN_ERR_BAD_DTYPE;
  if (!dX && !dCore) return DCTN_OK;
  EpsP p;
  const int precision = policy & DCTN_PREC_MASK;
  int rc = eps_fill_params(p, x_strides, C, B, H, W, Q, K, O, policy);
  if (rc != DCTN_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (p.opts & DCTN_OPT_GENERIC_KERNELS) return eps_bwd_generic(x, core, dY, dX, dCore, workspace, workspace_bytes, p, dtype, st);
  // dCore on the MFMA family when it covers the shape; whatever is left goes to the generic kernels
  unsigned char* ws = (unsigned char*)workspace;
  const size_t wa = align256(eps_bwd_mfma_workspace(p, dtype, precision, 0, dCore != nullptr));
  if (dCore && wa > 0) {
    if (!ws || workspace_bytes < wa) return DCTN_ERR_WORKSPACE;
    rc = eps_bwd_mfma(x, core, dY, nullptr, dCore, ws, wa, p, dtype, precision, st);
    if (rc == DCTN_OK) {
      dCore = nullptr;
    } else if (rc != DCTN_ERR_UNSUPPORTED) {
      return rc;
    }
  }
  size_t wq = 0;
  if (dCore && wa == 0 && !saved && (wq = align256(eps_bwd_q2f32_workspace(p, dtype, precision))) > 0) {
    if (!ws || workspace_bytes < wq) return DCTN_ERR_WORKSPACE;
    rc = eps_bwd_q2f32(x, dY, dCore, ws, wq, p, dtype, precision, st);
    if (rc == DCTN_OK) {
      dCore = nullptr;
    } else if (rc != DCTN_ERR_UNSUPPORTED) {
      return rc;
    }
  }
  if (!dX && !dCore) return DCTN_OK;
  const size_t off = (wa + wq) <= workspace_bytes ? (wa + wq) : workspace_bytes;
  // float64, and float32 shapes the bigcore family does not take: both gradients on the two-halves GEMM path
  if (dtype == DCTN_F64 || !eps_bigcore_covers(p, dtype, precision) || f32_prefers_halves(p, dtype)) {
    rc = eps_bwd_halves(x, core, dY, dX, dCore, ws ? ws + off : nullptr, workspace_bytes - off, p, dtype, st, saved,
                        saved_bytes);
    if (rc == DCTN_OK) return rc;
    if (rc != DCTN_ERR_UNSUPPORTED && rc != DCTN_ERR_WORKSPACE) return rc;
  }
  if (dCore) {
    // (its per-chunk slices use the tail of the workspace before dX does: the sum kernel is done with them by then)
    rc = eps_bwd_dcore_bigcore(x, dY, dCore, p, dtype, precision, st, ws ? ws + off : nullptr, workspace_bytes - off);
    if (rc == DCTN_OK) {
      dCore = nullptr;
    } else if (rc != DCTN_ERR_UNSUPPORTED) {
      return rc;
    }
  }
  if (dX) {
    // dX on the bigcore MFMA family (it shares the tail of the workspace with the generic kernels)
    rc = eps_bwd_dx_bigcore(x, core, dY, dX, ws ? ws + off : nullptr, workspace_bytes - off, p, dtype,
                            precision, st, saved, saved_bytes);
    if (rc == DCTN_OK) {
      dX = nullptr;
    } else if (rc != DCTN_ERR_UNSUPPORTED && rc != DCTN_ERR_WORKSPACE) {
      return rc;
    }
  }
  if (!dX && !dCore) return DCTN_OK;
  return eps_bwd_generic(x, core, dY, dX, dCore, ws ? ws + off : nullptr, workspace_bytes - off, p,
                         dtype, st);
}

int dctn_eps_bwd(const void* x, const int64_t x_strides[5], const void* core, const void* dY,
                 void* dX, void* dCore, void* workspace, size_t workspace_bytes, int C, int B,
                 int H, int W, int Q, int K, int O, int dtype, int policy, void* stream) {
  return eps_bwd_impl(x, x_strides, core, dY, nullptr, 0, dX, dCore, workspace, workspace_bytes, C, B, H, W, Q, K, O, dtype,
                      policy, stream);
}

int dctn_eps_bwd_saved(const void* x, const int64_t x_strides[5], const void* core, const void* dY, const void* saved,
                       size_t saved_bytes, void* dX, void* dCore, void* workspace, size_t workspace_bytes, int C, int B,
                       int H, int W, int Q, int K, int O, int dtype, int policy, void* stream) {
  if (!saved) return DCTN_ERR_NULL;
  return eps_bwd_impl(x, x_strides, core, dY, saved, saved_bytes, dX, dCore, workspace, workspace_bytes, C, B, H, W, Q, K,
                      O, dtype, policy, stream);
}

}  // extern "C"
