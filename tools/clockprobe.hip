// In-kernel shader clock: delta s_memtime / delta s_memrealtime (100 MHz) around a busy loop,
// plus a dependent v_fma chain and an independent v_fma stream to get cycles per VALU instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out, float* sink, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4=0.f,c5=0.f,c6=0.f,c7=0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      c0 = __builtin_fmaf(a, b, c0); c1 = __builtin_fmaf(a, b, c1); c2 = __builtin_fmaf(a, b, c2); c3 = __builtin_fmaf(a, b, c3);
      c4 = __builtin_fmaf(a, b, c4); c5 = __builtin_fmaf(a, b, c5); c6 = __builtin_fmaf(a, b, c6); c7 = __builtin_fmaf(a, b, c7);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
int main() {
  unsigned long long* d; float* s;
  const int blocks[3] = {256, 1024, 4096}; const int threads[3] = {64, 256, 256};
  hipMalloc(&d, 4096 * 16); hipMalloc(&s, 4096 * 256 * 4);
  for (int rep = 0; rep < 2; ++rep)
  for (int cfg = 0; cfg < 3; ++cfg) {
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<<<blocks[cfg], threads[cfg]>>>(d, s, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    double clk = (double)h[0] / (double)h[1] * 100.0;  // MHz
    double instr = (double)iters * 16 * 8;
    printf("blocks=%d threads=%d: kernel %.3f ms, shader clock %.0f MHz, cycles per v_fma per wave %.2f (waves/SIMD=%.1f)\n",
           blocks[cfg], threads[cfg], ms, clk, (double)h[0] / instr, blocks[cfg] * (threads[cfg] / 64.0) / 1024.0);
  }
  return 0;
}
