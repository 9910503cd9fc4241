// Probe: issue rate of v_fma_f64 (and v_pk-less alternatives) against v_mfma_f64_16x16x4_f64's measured 19.5 flop per cycle and SIMD
// (105 cycles per instruction, tools/probes/mfma_f64_rate.hip): NCH independent chains per lane, 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_f64_rate.hip -o tools/probes/valu_f64_rate && tools/probes/valu_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NCH>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  double acc[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) acc[i] = (double)i;
  const double a = out[threadIdx.x & 7] + 1.0000001, b = out[(threadIdx.x & 7) + 8] + 0.5;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NCH; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  double r = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) r += acc[i];
  out[16 + blockIdx.x * 256 + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[16 + 1024 * 256 + 1] = (double)(c1 - c0);
}

int main() {
  double* out;
  (void)hipMalloc(&out, (16 + 1024 * 256 + 8) * 8); (void)hipMemset(out, 0, (16 + 1024 * 256 + 8) * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int grid = 256 * wps;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, out, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    double cyc; (void)hipMemcpy(&cyc, out + 16 + 1024 * 256 + 1, 8, hipMemcpyDeviceToHost);
    const double n = (double)grid * 4 * iters * 8 * 4;   // wave instructions
    printf("v_fma_f64, 8 chains, %d waves/SIMD: %.1f TFLOP/s (%.2f ns per wave instruction and SIMD; one wave: %.1f cycles per instruction of its own)\n",
           wps, n * 128 / best * 1e-9, best * 1e6 / (n / 1024.0), cyc / (iters * 32.0));
  }
  return 0;
}
