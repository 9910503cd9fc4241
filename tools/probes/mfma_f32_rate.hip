// Probe: sustained rate of v_mfma_f32_32x32x2_f32 as the bigcore kernels issue it: NACC accumulators per wave, the B operand
// of every MFMA the product of two registers formed just before it (VMUL = 1) or a loop-invariant register (0), 1 / 2 / 3
// waves per SIMD.  Peak: 64 cycles per MFMA and SIMD = 157 TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_rate.hip -o tools/probes/mfma_f32_rate && tools/probes/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, int VMUL>
__global__ void rate(float* out, int iters, float a0, float b0) {
  f32x16 c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) c[i][v] = 0.f;
  float a = a0 + threadIdx.x, hi[NACC], tab[8];
#pragma unroll
  for (int i = 0; i < NACC; ++i) hi[i] = b0 + i;
#pragma unroll
  for (int t = 0; t < 8; ++t) tab[t] = b0 * (t + 1);
  for (int it = 0; it < iters; ++it) {
    if (VMUL == 2) {   // the products of half a block first, then its MFMAs back to back
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        float bop[NACC][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < NACC; ++i) bop[i][t] = hi[i] * tab[4 * hlf + t];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bop[i][t], c[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        const float b = VMUL ? hi[i] * tab[t] : tab[t];
        c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
      }
    }
    if (VMUL) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) hi[i] += 1e-9f;
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) r += c[i][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC, int VMUL>
void run(float* d) {
  const int iters = 4000;
  for (int wps = 1; wps <= 3; ++wps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<NACC, VMUL>), dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0f, 1e-6f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((rate<NACC, VMUL>), dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f, 1e-6f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = 256.0 * wps * 4 * iters * 8 * NACC;
    printf("%d accumulators, B %s, %d wave(s) per SIMD: %.1f TFLOP/s (%.2f of 157)\n", NACC, VMUL == 2 ? "= 8 v_mul, then 8 MFMAs" : VMUL ? "= v_mul just before" : "invariant", wps,
           mf * 4096.0 / ms * 1e-9, mf * 4096.0 / ms * 1e-9 / 157.3);
  }
}
int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 3 * 256 * sizeof(float));
  run<1, 0>(d); run<2, 0>(d); run<2, 1>(d); run<4, 1>(d); run<2, 2>(d);
  return 0;
}
