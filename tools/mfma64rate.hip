// Probe: sustained rate of v_mfma_f64_16x16x4_f64 on gfx950 (no memory traffic), waves/SIMD swept.
// hipcc --offload-arch=gfx950 -O2 tools/mfma64rate.hip -o tools/mfma64rate && tools/mfma64rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) double f64x4;
__global__ void rate(double* out, int iters, double a0, double b0) {
  f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  double* d;
  hipMalloc(&d, 256 * 16 * 1024 * sizeof(double));
  const int iters = 20000;
  for (int wpb = 4; wpb <= 16; wpb *= 2) {   // waves per workgroup = waves per CU (1 workgroup per CU)
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate, dim3(256), dim3(64 * wpb), 0, 0, d, 100, 1.0, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate, dim3(256), dim3(64 * wpb), 0, 0, d, iters, 1.0, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * wpb * iters * 4 * 2048.0;
    printf("waves/CU %2d: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", wpb, ms,
           flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 4.0 * wpb / 4.0));
  }
  return 0;
}
