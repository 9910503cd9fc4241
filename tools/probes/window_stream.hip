// Probe: the memory side of the logmatmulexp fold alone.  A wave walks windows of L = 9 matrices of 1 KiB (16 B per lane,
// 4 loads in flight across the window boundary, like lme_fold16_fwd_mfma_k) and writes 1 KiB per window; the only
// arithmetic is an add per loaded value.  WORK > 0 adds that many dependent v_fma per matrix (a stand-in for the step's
// arithmetic latency).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/window_stream.hip -o tools/probes/window_stream && tools/probes/window_stream
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WORK>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ mats, float4* __restrict__ out, long long Wn, int L) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
  long long pw = wave; int pl = 0;
  auto fetch = [&]() {
    const bool in = pw < Wn;
    const float4 q = mats[((in ? pw : wave) * L + (in ? pl : 0)) * 64 + lane];
    if (++pl == L) { pl = 0; pw += nw; }
    return q;
  };
  if (wave >= Wn) return;
  float4 q0 = fetch(), q1 = fetch(), q2 = fetch(), q3 = fetch();
  for (long long w = wave; w < Wn; w += nw) {
    float4 acc = {0, 0, 0, 0};
    for (int l = 0; l < L; ++l) {
      const float4 m = q0;
      q0 = q1; q1 = q2; q2 = q3; q3 = fetch();
      float t = m.x + m.y + m.z + m.w;
#pragma unroll
      for (int i = 0; i < WORK; ++i) t = __builtin_fmaf(t, 1.0001f, acc.x);
      acc.x += t; acc.y += m.y; acc.z += m.z; acc.w += m.w;
    }
    out[w * 64 + lane] = acc;
  }
}

int main() {
  const long long Wn = 692224; const int L = 9;
  float4 *mats, *out;
  if (hipMalloc(&mats, Wn * L * 1024) != hipSuccess || hipMalloc(&out, Wn * 1024) != hipSuccess) return 1;
  (void)hipMemset(mats, 0, Wn * L * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wpc = 4; wpc <= 8; wpc += 1) {
    for (int work = 0; work < 3; ++work) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        const dim3 g(256 * wpc), b(256);
        if (work == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, mats, out, Wn, L);
        if (work == 1) hipLaunchKernelGGL(k<64>, g, b, 0, 0, mats, out, Wn, L);
        if (work == 2) hipLaunchKernelGGL(k<256>, g, b, 0, 0, mats, out, Wn, L);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("%2d waves/CU, %3d dependent fma per matrix: %.3f ms  %.2f TB/s\n", 4 * wpc, work == 0 ? 0 : work == 1 ? 64 : 256, best,
             Wn * (L + 1) * 1024.0 / best * 1e-9);
    }
  }
  return 0;
}
