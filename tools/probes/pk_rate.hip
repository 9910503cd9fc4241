// Probe (round 4): issue rate of packed f32 vector instructions against their scalar forms, one to four waves per SIMD,
// no matrix instructions anywhere: cycles per instruction of a stream of independent v_fma_f32 / v_pk_fma_f32 /
// v_pk_mul_f32 (16 accumulators each).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  f2 a[16];
  const float s = out[threadIdx.x & 3] + 1.0001f, t = out[(threadIdx.x & 3) + 4] * 1e-6f;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = f2{(float)threadIdx.x + i, (float)i};
  const f2 s2 = {s, s + 1e-7f}, t2 = {t, t * 0.5f};
  const long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) {   // 2 scalar fma
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(s), "v"(t));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(s), "v"(t));
      } else if (MODE == 1) {   // 1 packed fma
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(s2), "v"(t2));
      } else if (MODE == 2) {   // 1 packed mul
        asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(s2));
      } else if (MODE == 3) {   // 2 scalar mul
        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(s));
        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].y) : "v"(s));
      } else {   // packed fma with a broadcast operand (op_sel_hi: the low half twice)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(s2), "v"(t2));
      }
    }
  }
  const long long c1 = __builtin_readcyclecounter();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += a[i].x + a[i].y;
  out[8 + blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, (8 + 2048 * 256) * 4); (void)hipMemset(out, 0, (8 + 2048 * 256) * 4);
  (void)hipMalloc(&cyc, 2048 * 8);
  const int iters = 2000;
  const char* names[] = {"2 x v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "2 x v_mul_f32", "v_pk_fma_f32 broadcast"};
  for (int mode = 0; mode < 5; ++mode)
    for (int wps = 1; wps <= 4; wps *= 2) {   // workgroups of 4 waves: wps workgroups per CU = wps waves per SIMD
      const int grid = 256 * wps;
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
      (void)hipDeviceSynchronize();
      long long h[2048]; (void)hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < grid; ++i) avg += (double)h[i]; avg /= grid;
      printf("%-24s %d waves/SIMD: %.2f cycles per pair of results per wave (x waves = %.2f SIMD cycles)\n", names[mode], wps,
             avg / (iters * 16.0), avg / (iters * 16.0) / wps);
    }
  return 0;
}
