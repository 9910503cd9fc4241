// Probe (round 4): issue rate of v_mfma_f64_16x16x4_f64 - N independent accumulators, 1..4 waves per SIMD - against the
// 78.6 TFLOP/s the part is specified at (= one instruction per 64 cycles and SIMD at 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_rate.hip -o /tmp/f64rate && /tmp/f64rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  const double a = out[threadIdx.x & 7] + 1.0, b = out[(threadIdx.x & 7) + 8] + 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[16 + blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  double* out;
  (void)hipMalloc(&out, (16 + 1024 * 256) * 8); (void)hipMemset(out, 0, (16 + 1024 * 256) * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int nacc = 1; nacc <= 4; nacc *= 2)
    for (int wps = 1; wps <= 4; wps *= 2) {
      const int grid = 256 * wps;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters);
        if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters);
        if (nacc == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      const double n = (double)grid * 4 * iters * nacc;   // instructions
      printf("%d accumulators, %d waves/SIMD: %.1f TFLOP/s  (%.1f ns per instruction and SIMD)\n", nacc, wps, n * 2048 / best * 1e-9,
             best * 1e6 / (n / 1024.0));
    }
  return 0;
}
