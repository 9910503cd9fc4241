// Probe: do vector-ALU instructions hide behind matrix instructions?  Per loop turn NM matrix instructions (two accumulator
// chains) and NV independent v_fma_f32 (four chains, other registers), interleaved evenly; one or two waves per SIMD.
// Shader cycles per turn (s_memtime) for the exact-f32 instruction v_mfma_f32_32x32x2_f32 (64 cycles each) and for
// v_mfma_f32_32x32x16_bf16 (32 cycles each).  If the vector work hides, cycles stay at NM x 64 (32) until the issue slots run out;
// if both share the pipe, cycles = NM x 64 + NV x c.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_share.hip -o tools/probes/mfma_valu_share && tools/probes/mfma_valu_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NV, bool BF16>
__global__ void probe(float* out, unsigned long long* cyc, int iters, float a0) {
  constexpr int NM = 16;
  f32x16 c0, c1;
  for (int v = 0; v < 16; ++v) { c0[v] = 0.f; c1[v] = 0.f; }
  float a = a0 + threadIdx.x, b = a0 * 0.5f;
  bf16x8 ah, bh;
  for (int j = 0; j < 8; ++j) { ah[j] = (__bf16)(a0 + j); bh[j] = (__bf16)(a0 - j); }
  float f0 = a0, f1 = a0 + 1, f2 = a0 + 2, f3 = a0 + 3;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (BF16) {
        if (m & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c1, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c0, 0, 0, 0);
      } else {
        if (m & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < NV / NM; ++v) {
        asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b), "v"(a));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + f0 + f1 + f2 + f3;
}

template <int NV, bool BF16>
void run(int waves_per_simd) {
  const int blocks = 256, threads = 256 * waves_per_simd, iters = 2000;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks * threads / 64);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<NV, BF16>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double per_turn = (double)h[h.size() / 2] / iters;   // a wave's cycles per turn of 16 matrix instructions
  // NV counts groups: each "unit" of NV / 16 is FOUR v_fma_f32
  printf("%s  %d wave(s)/SIMD  vector instructions per turn %3d: %7.1f cycles per turn and wave  (matrix alone would be %d x %d)\n",
         BF16 ? "bf16 32x32x16" : "f32  32x32x2 ", waves_per_simd, NV / 16 * 4 * 16, per_turn, 16 * waves_per_simd, BF16 ? 32 : 64);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0, false>(w); run<16, false>(w); run<32, false>(w); run<64, false>(w);
    run<0, true>(w); run<16, true>(w); run<32, true>(w); run<64, true>(w);
  }
  return 0;
}
