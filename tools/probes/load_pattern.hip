// Probe: does the ORDER in which the 64 lanes of a 16-byte-per-lane load cover one contiguous 1 KiB block matter?
//   linear:   lane l reads bytes [16 l, 16 l + 16)
//   mfma16:   lane (c = l % 16, g = l / 16) reads row c (64 B), bytes [16 g, 16 g + 16) - the A/B operand layout of
//             v_mfma_f32_16x16x4_f32 for a row-major 16x16 float matrix (what logmatmulexp.hip's fold kernels load)
// Each wave streams 1 KiB blocks (4 in flight), the whole buffer once.  Also the same two orders for stores.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/load_pattern.hip -o tools/probes/load_pattern && tools/probes/load_pattern
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ src, float* __restrict__ sink, long long nblk) {
  const int lane = threadIdx.x & 63;
  const int off = MODE == 0 ? lane : (lane & 15) * 4 + (lane >> 4);   // float4 index inside the block
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
  float4 acc = {0, 0, 0, 0};
  for (long long b = wave; b < nblk; b += 4 * nw) {
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const long long bb = b + i * nw < nblk ? b + i * nw : b; v[i] = src[bb * 64 + off]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
template <int MODE>
__global__ __launch_bounds__(256) void wr(float4* __restrict__ dst, long long nblk) {
  const int lane = threadIdx.x & 63;
  const int off = MODE == 0 ? lane : (lane & 15) * 4 + (lane >> 4);
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
  for (long long b = wave; b < nblk; b += nw) dst[b * 64 + off] = make_float4((float)b, 1.f, 2.f, 3.f);
}

int main() {
  const long long bytes = 6LL << 30, nblk = bytes / 1024;
  float4* buf; float* sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wpc = 4; wpc <= 8; wpc += 4) {   // workgroups of 4 waves per CU
    const dim3 g(256 * wpc), b(256);
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(rd<0>, g, b, 0, 0, buf, sink, nblk);
        if (mode == 1) hipLaunchKernelGGL(rd<1>, g, b, 0, 0, buf, sink, nblk);
        if (mode == 2) hipLaunchKernelGGL(wr<0>, g, b, 0, 0, buf, nblk);
        if (mode == 3) hipLaunchKernelGGL(wr<1>, g, b, 0, 0, buf, nblk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      const char* names[] = {"read  linear", "read  mfma16 order", "write linear", "write mfma16 order"};
      printf("%2d waves/CU  %-20s %.2f ms  %.2f TB/s\n", 4 * wpc, names[mode], best, bytes / best * 1e-9);
    }
  }
  return 0;
}
