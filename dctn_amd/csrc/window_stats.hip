// Window statistics of the input feature map — the step just before the EPS path.
//
// Reference: dctn/dataset_loading.py:79-94 `calc_scaling_factor` builds, for 10 880 samples, the
// tensor of all K x K windows (`make_windows`, dctn/align.py:49-61: K*K shifted views stacked, i.e.
// K*K copies of the data set) and asks `RankOneTensorsBatch` (dctn/rank_one_tensor.py:53-100) for
// the mean and the variance of the rank-one tensors  T_w = (x)_n x_n[w, :]  over all windows w:
//     sum(T_w)   = prod_n sum_q x_n[w,q]          ||T_w||^2 = prod_n sum_q x_n[w,q]^2
// Here one lane owns a window, forms the two products from the window's N*Q features read in
// place (no window tensor is ever materialised) and the workgroup adds its float64 partial sums
// to the two global accumulators.  HBM: x once (plus the K*K-fold overlap served by the caches).
#include "common.h"

namespace {

template <typename S>
__global__ __launch_bounds__(256) void window_stats_k(const S* __restrict__ x, double* __restrict__ sums, EpsP p) {
  __shared__ double red[2][4];
  double s_sum = 0.0, s_sq = 0.0;
  const int hw = p.Ho * p.Wo;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < p.Wn;
       w += (long long)gridDim.x * blockDim.x) {
    const long long b = w / hw;
    const int rem = (int)(w - b * hw);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    double ps = 1.0, pq = 1.0;
    for (int n = 0; n < p.N; ++n) {
      const int pos = n / p.C, ch = n - pos * p.C;
      const int dh = pos / p.K, dw = pos - dh * p.K;
      const S* px = x + ch * p.s[0] + b * p.s[1] + (long long)(ho + dh) * p.s[2] + (long long)(wo + dw) * p.s[3];
      double t = 0.0, u = 0.0;
      for (int q = 0; q < p.Q; ++q) {
        const double v = (double)(float)px[q * p.s[4]];
        t += v;
        u += v * v;
      }
      ps *= t;
      pq *= u;
    }
    s_sum += ps;
    s_sq += pq;
  }
  // wave reduction, then one atomic pair per workgroup
  for (int off = 32; off > 0; off >>= 1) {
    s_sum += __shfl_down(s_sum, off, 64);
    s_sq += __shfl_down(s_sq, off, 64);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[0][wv] = s_sum; red[1][wv] = s_sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[0], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    atomicAdd(&sums[1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

template <>
__global__ __launch_bounds__(256) void window_stats_k<double>(const double* __restrict__ x, double* __restrict__ sums,
                                                               EpsP p) {
  __shared__ double red[2][4];
  double s_sum = 0.0, s_sq = 0.0;
  const int hw = p.Ho * p.Wo;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < p.Wn;
       w += (long long)gridDim.x * blockDim.x) {
    const long long b = w / hw;
    const int rem = (int)(w - b * hw);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    double ps = 1.0, pq = 1.0;
    for (int n = 0; n < p.N; ++n) {
      const int pos = n / p.C, ch = n - pos * p.C;
      const int dh = pos / p.K, dw = pos - dh * p.K;
      const double* px = x + ch * p.s[0] + b * p.s[1] + (long long)(ho + dh) * p.s[2] + (long long)(wo + dw) * p.s[3];
      double t = 0.0, u = 0.0;
      for (int q = 0; q < p.Q; ++q) {
        const double v = px[q * p.s[4]];
        t += v;
        u += v * v;
      }
      ps *= t;
      pq *= u;
    }
    s_sum += ps;
    s_sq += pq;
  }
  for (int off = 32; off > 0; off >>= 1) {
    s_sum += __shfl_down(s_sum, off, 64);
    s_sq += __shfl_down(s_sq, off, 64);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[0][wv] = s_sum; red[1][wv] = s_sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[0], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    atomicAdd(&sums[1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

}  // namespace

// {sum y, sum y^2} of n stored values, float64, accumulated into stats[2]: the reduction behind the
// empirical-output-std initialisation (dctn/eps.py:172-175 `output.std(unbiased=False)`) for the kernel families
// that have no in-kernel epilogue: the forward writes one slice into a scratch buffer that stays in the caches and
// this pass folds it into the two running sums, so the output of the whole data set never exists.
template <typename S>
__global__ __launch_bounds__(256) void eps_out_stats_k(const S* __restrict__ y, long long n, double* __restrict__ stats) {
  __shared__ double red[2][4];
  double s1 = 0.0, s2 = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double v = (double)(float)y[i];
    s1 += v;
    s2 += v * v;
  }
  s1 = wave_reduce_sum(s1);
  s2 = wave_reduce_sum(s2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&stats[0], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    atomicAdd(&stats[1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}
template <>
__global__ __launch_bounds__(256) void eps_out_stats_k<double>(const double* __restrict__ y, long long n, double* __restrict__ stats) {
  __shared__ double red[2][4];
  double s1 = 0.0, s2 = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double v = y[i];
    s1 += v;
    s2 += v * v;
  }
  s1 = wave_reduce_sum(s1);
  s2 = wave_reduce_sum(s2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&stats[0], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    atomicAdd(&stats[1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

int eps_out_stats(const void* y, long long n, int dtype, double* stats, hipStream_t st) {
  long long blocks = (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  const dim3 g((unsigned)blocks), b(256);
  switch (dtype) {
    case DCTN_F32: hipLaunchKernelGGL(eps_out_stats_k<float>, g, b, 0, st, (const float*)y, n, stats); break;
    case DCTN_F64: hipLaunchKernelGGL(eps_out_stats_k<double>, g, b, 0, st, (const double*)y, n, stats); break;
    case DCTN_BF16: hipLaunchKernelGGL(eps_out_stats_k<bf16_t>, g, b, 0, st, (const bf16_t*)y, n, stats); break;
    default: return DCTN_ERR_BAD_DTYPE;
  }
  DCTN_CHECK_LAUNCH();
  return DCTN_OK;
}

extern "C" int dctn_window_stats(const void* x, const int64_t x_strides[5], void* sums, int C, int B, int H, int W,
                                 int Q, int K, int dtype, void* stream) {
  if (!x || !x_strides || !sums) return DCTN_ERR_NULL;
  if (C < 1 || B < 1 || Q < 1 || K < 1 || H < K || W < K) return DCTN_ERR_BAD_SHAPE;
  EpsP p = {};   // only the window geometry is used (no core: any number of factors)
  p.C = C; p.B = B; p.H = H; p.W = W; p.Q = Q; p.K = K; p.O = 1;
  p.N = K * K * C;
  p.Ho = H - K + 1; p.Wo = W - K + 1;
  p.Wn = (long long)B * p.Ho * p.Wo;
  for (int i = 0; i < 5; ++i) p.s[i] = x_strides[i];
  hipStream_t st = (hipStream_t)stream;
  if (dctn_zero_async(sums, 2 * sizeof(double), st) != DCTN_OK) return DCTN_ERR_LAUNCH;
  long long blocks = (p.Wn + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  const dim3 g((unsigned)blocks), b(256);
  switch (dtype) {
    case DCTN_F32: hipLaunchKernelGGL(window_stats_k<float>, g, b, 0, st, (const float*)x, (double*)sums, p); break;
    case DCTN_F64: hipLaunchKernelGGL(window_stats_k<double>, g, b, 0, st, (const double*)x, (double*)sums, p); break;
    case DCTN_BF16: hipLaunchKernelGGL(window_stats_k<bf16_t>, g, b, 0, st, (const bf16_t*)x, (double*)sums, p); break;
    default: return DCTN_ERR_BAD_DTYPE;
  }
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("window_stats");
  return DCTN_OK;
}

// ------------------------------------------------------------------------------------------------ feature map on the device
// phi_cos_sin_squared_1 (dctn/dataset_loading.py:33-36): u -> (2 sin^2(pi u / 2), 2 cos^2(pi u / 2)), evaluated in
// float32 like the reference evaluates it on its float32 images.
__device__ __forceinline__ void phi_cs2(float u, float& a, float& b) {
  float sn, cs;
  sincosf(u * 1.57079632679489661923f, &sn, &cs);
  a = 2.f * sn * sn;
  b = 2.f * cs * cs;
}

// x[0, b, h, w, :] = scale * phi(images[b, h, w]): the (1, samples, h, w, 2) tensor of dataset_loading.py:63 written once,
// in the model's dtype, with the scaling factor of new_runner.py's autoscale folded in (no float32 copy, no second pass)
template <typename S>
__global__ __launch_bounds__(256) void phi_expand_k(const float* __restrict__ img, S* __restrict__ x, long long n, float scale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float a, b;
    phi_cs2(img[i], a, b);
    x[2 * i] = (S)(scale * a);
    x[2 * i + 1] = (S)(scale * b);
  }
}

// Window statistics straight from the raw images: one workgroup per image applies phi to every pixel ONCE (per-pixel
// sum and sum of squares of the two features, float64, in LDS) and forms the K x K window products from there - neither
// the expanded (1, N, H, W, 2) tensor nor the K*K stacked window copies of dataset_loading.py:79-94 exist.
__global__ __launch_bounds__(256) void phi_window_stats_k(const float* __restrict__ img, double* __restrict__ sums, int H,
                                                          int W, int K) {
  extern __shared__ double pix[];   // [H*W][2]: t = a + b, u = a^2 + b^2
  __shared__ double red[2][4];
  const float* im = img + (long long)blockIdx.x * H * W;
  for (int i = threadIdx.x; i < H * W; i += 256) {
    float a, b;
    phi_cs2(im[i], a, b);
    pix[2 * i] = (double)a + (double)b;
    pix[2 * i + 1] = (double)a * (double)a + (double)b * (double)b;
  }
  __syncthreads();
  const int Ho = H - K + 1, Wo = W - K + 1;
  double s_sum = 0.0, s_sq = 0.0;
  for (int w = threadIdx.x; w < Ho * Wo; w += 256) {
    const int ho = w / Wo, wo = w - ho * Wo;
    double ps = 1.0, pq = 1.0;
    for (int dh = 0; dh < K; ++dh)
      for (int dw = 0; dw < K; ++dw) {
        const int i = (ho + dh) * W + wo + dw;
        ps *= pix[2 * i];
        pq *= pix[2 * i + 1];
      }
    s_sum += ps;
    s_sq += pq;
  }
  s_sum = wave_reduce_sum(s_sum);
  s_sq = wave_reduce_sum(s_sq);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[0][wv] = s_sum; red[1][wv] = s_sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[0], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    atomicAdd(&sums[1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

extern "C" int dctn_phi_expand(const void* images, void* x, int64_t n_pixels, float scale, int dtype, void* stream) {
  if (!images || !x) return DCTN_ERR_NULL;
  if (n_pixels < 1) return DCTN_ERR_BAD_SHAPE;
  long long blocks = (n_pixels + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const dim3 g((unsigned)blocks), b(256);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case DCTN_F32: hipLaunchKernelGGL(phi_expand_k<float>, g, b, 0, st, (const float*)images, (float*)x, (long long)n_pixels, scale); break;
    case DCTN_F64: hipLaunchKernelGGL(phi_expand_k<double>, g, b, 0, st, (const float*)images, (double*)x, (long long)n_pixels, scale); break;
    case DCTN_BF16: hipLaunchKernelGGL(phi_expand_k<bf16_t>, g, b, 0, st, (const float*)images, (bf16_t*)x, (long long)n_pixels, scale); break;
    default: return DCTN_ERR_BAD_DTYPE;
  }
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("phi_expand");
  return DCTN_OK;
}

extern "C" int dctn_phi_window_stats(const void* images, void* sums, int B, int H, int W, int K, void* stream) {
  if (!images || !sums) return DCTN_ERR_NULL;
  if (B < 1 || K < 1 || H < K || W < K) return DCTN_ERR_BAD_SHAPE;
  const size_t lds = (size_t)H * W * 2 * sizeof(double);
  if (lds > DCTN_LDS_BUDGET) return DCTN_ERR_UNSUPPORTED;   // (images beyond ~97 x 97: expand, then dctn_window_stats)
  hipStream_t st = (hipStream_t)stream;
  if (dctn_zero_async(sums, 2 * sizeof(double), st) != DCTN_OK) return DCTN_ERR_LAUNCH;
  (void)hipFuncSetAttribute((const void*)phi_window_stats_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(phi_window_stats_k, dim3((unsigned)B), dim3(256), lds, st, (const float*)images, (double*)sums, H, W, K);
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("phi_window_stats");
  return DCTN_OK;
}
