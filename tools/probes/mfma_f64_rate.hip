// Probe (round 4): issue rate of v_mfma_f64_16x16x4_f64 - N independent accumulators, 1..4 waves per SIMD - against the
// 78.6 TFLOP/s the part is specified at (= one instruction per 64 cycles and SIMD at 2.4 GHz).
// Round 5: the kernel also reads the shader clock (s_memtime) and the 100 MHz wall clock (s_memrealtime) around its loop:
// CYCLES per instruction and the clock the chip held, so that "44 ns instead of 27" can be told apart - a slower
// instruction (more cycles) or a lower clock (the same cycles).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_rate.hip -o /tmp/f64rate && /tmp/f64rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_cyc[2];   // shader cycles and 100 MHz ticks of workgroup 0's loop

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  d4 acc[NACC];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  const double a = out[threadIdx.x & 7] + 1.0, b = out[(threadIdx.x & 7) + 8] + 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_cyc[0] = c1 - c0; g_cyc[1] = r1 - r0; }
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
      unsigned long long cyc[2];
      (void)hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), sizeof(cyc));
      // in-kernel: one wave's loop, cycles per instruction OF THAT WAVE (its SIMD issues `resident waves` times as many in
      // the same time; the launch's waves per SIMD are only all resident when the event time confirms it: ns x GHz =
      // cycles per instruction and SIMD)
      const double ghz = (double)cyc[0] / ((double)cyc[1] * 10.0), ns = best * 1e6 / (n / 1024.0);
      printf("%d accumulators, %d waves/SIMD: %.1f TFLOP/s  (%.1f ns = %.0f cycles per instruction and SIMD at the %.2f GHz the kernel read; "
             "one wave: %.0f cycles per instruction of its own)\n",
             nacc, wps, n * 2048 / best * 1e-9, ns, ns * ghz, ghz, (double)cyc[0] / ((double)iters * nacc));
    }
  return 0;
}
