// Probe: what does straight-line code cost the first time a CU executes it (instruction-cache misses) against the
// second time?  One wave per workgroup, 256 workgroups; the body is NI independent v_add_f32 (8 bytes each, VOP3).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/icache_cold.hip -o tools/probes/icache_cold && tools/probes/icache_cold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define BODY R256("v_add_f32_e64 %0, %0, 1.0\n\t")   // 256 x 8 bytes = 2 KiB per BODY

template <int KB>
__global__ void probe(unsigned long long* out, float* sink) {
  float v = threadIdx.x;
  unsigned long long t[4];
#pragma unroll 1
  for (int it = 0; it < 3; ++it) {
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t[it])::"memory");
    if constexpr (KB >= 2) asm volatile(BODY : "+v"(v));
    if constexpr (KB >= 4) asm volatile(BODY : "+v"(v));
    if constexpr (KB >= 8) { asm volatile(BODY : "+v"(v)); asm volatile(BODY : "+v"(v)); }
    if constexpr (KB >= 16) { asm volatile(BODY : "+v"(v)); asm volatile(BODY : "+v"(v)); asm volatile(BODY : "+v"(v)); asm volatile(BODY : "+v"(v)); }
  }
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t[3])::"memory");
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = t[1] - t[0];
    out[blockIdx.x * 4 + 1] = t[2] - t[1];
    out[blockIdx.x * 4 + 2] = t[3] - t[2];
  }
  if (v == -1.f) sink[0] = v;
}

template <int KB>
void run(unsigned long long* d, float* sink) {
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe<KB>, dim3(256), dim3(64), 0, 0, d, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> a, b, c;
    for (int i = 0; i < 256; ++i) { a.push_back(h[i * 4] * 0.01); b.push_back(h[i * 4 + 1] * 0.01); c.push_back(h[i * 4 + 2] * 0.01); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
    printf("%2d KiB of straight-line code, launch %d: first pass %.2f us (median over 256 workgroups), second %.2f, third %.2f\n", KB, rep,
           a[128], b[128], c[128]);
  }
}

int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 256 * 4 * 8); hipMalloc(&sink, 4);
  run<2>(d, sink); run<4>(d, sink); run<8>(d, sink); run<16>(d, sink);
  return 0;
}
