// Probe: can v_fma_f64 with its matrix operand BROADCAST from scalar registers (s_load_dwordx16 of a shared matrix, every wave
// streaming the whole matrix) sustain more than v_mfma_f64_16x16x4_f64's 47.9 TFLOP/s?  Lane = a column (window) of the product:
//   acc[n] += M[k][n0 + n] * p[k]   for n < 32 (32 double accumulators per lane), k streamed, M row-major (K x N doubles).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_f64_sgpr_gemm.hip -o tools/probes/valu_f64_sgpr_gemm && tools/probes/valu_f64_sgpr_gemm
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int NB = 32;   // accumulators per lane = doubles of a matrix row taken per k

// the same with the next row's scalar loads issued BEFORE the current row's multiply-adds (16 columns per wave: two rows of 16
// doubles = 64 scalar registers in flight)
__global__ __launch_bounds__(256) void k2(const double* __restrict__ M, double* out, int K, int N, int nblocks_n) {
  constexpr int NB2 = 16;
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
  const int nb = wave % nblocks_n;
  const double* row = M + (size_t)nb * NB2;
  double acc[NB2];
#pragma unroll
  for (int n = 0; n < NB2; ++n) acc[n] = 0.0;
  double p = 1.0 + 1e-9 * threadIdx.x;
  double cur[NB2];
#pragma unroll
  for (int n = 0; n < NB2; ++n) cur[n] = row[n];
  for (int kk = 0; kk < K; ++kk) {
    const double* r = row + (size_t)(kk + 1 < K ? kk + 1 : kk) * N;
    double nxt[NB2];
#pragma unroll
    for (int n = 0; n < NB2; ++n) nxt[n] = r[n];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NB2; ++n) acc[n] = __builtin_fma(cur[n], p, acc[n]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NB2; ++n) cur[n] = nxt[n];
    p += 1e-12;
  }
  double s = 0;
#pragma unroll
  for (int n = 0; n < NB2; ++n) s += acc[n];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k(const double* __restrict__ M, double* out, int K, int N, int nblocks_n) {
  // uniform per wave: which 32-column block it owns; lanes = 64 windows
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
  const int nb = wave % nblocks_n;
  const double* row = M + (size_t)nb * NB;   // uniform pointer -> scalar loads
  double acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) acc[n] = 0.0;
  double p = 1.0 + 1e-9 * threadIdx.x;
  for (int kk = 0; kk < K; ++kk) {
    const double* r = row + (size_t)kk * N;
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[n] = __builtin_fma(r[n], p, acc[n]);   // r[n]: uniform address -> SGPR operand
    p += 1e-12;
  }
  double s = 0;
#pragma unroll
  for (int n = 0; n < NB; ++n) s += acc[n];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int K = 256, N = 512;   // the cfg1 core as a matrix: 1 MiB of doubles
  double *M, *out;
  (void)hipMalloc(&M, (size_t)K * N * 8); (void)hipMemset(M, 0, (size_t)K * N * 8);
  (void)hipMalloc(&out, 8192 * 256 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int grid = 256 * wps;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, M, out, K, N, N / NB);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double flops = (double)grid * 4 * 64 * K * NB * 2;
    printf("%d waves/SIMD: %.3f ms, %.1f TFLOP/s\n", wps, best, flops / best * 1e-9);
  }
  for (int wps = 1; wps <= 8; wps *= 2) {
    const int grid = 256 * wps;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k2, dim3(grid), dim3(256), 0, 0, M, out, K, N, N / 16);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double flops = (double)grid * 4 * 64 * K * 16 * 2;
    printf("prefetched rows of 16, %d waves/SIMD: %.3f ms, %.1f TFLOP/s\n", wps, best, flops / best * 1e-9);
  }
  return 0;
}
