// Probe (round 4): variants of window_stream.hip's access pattern - nontemporal loads / stores, a contiguous range of
// windows per wave instead of the grid stride, two waves per window pair.  Same traffic: 9 KiB read + 1 KiB written per window.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/window_stream2.hip -o /tmp/ws2 && /tmp/ws2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NT, int CONTIG>
__global__ __launch_bounds__(256) void k(const v4f* __restrict__ mats, v4f* __restrict__ out, long long Wn, int L) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
  const long long per = (Wn + nw - 1) / nw;
  const long long wbeg = CONTIG ? wave * per : wave, wend = CONTIG ? (wbeg + per < Wn ? wbeg + per : Wn) : Wn, wstep = CONTIG ? 1 : nw;
  if (wbeg >= wend) return;
  long long pw = wbeg; int pl = 0;
  auto fetch = [&]() {
    const bool in = pw < wend;
    const v4f* src = mats + ((in ? pw : wbeg) * L + (in ? pl : 0)) * 64 + lane;
    v4f q;
    if (NT) q = __builtin_nontemporal_load(src); else q = *src;
    if (++pl == L) { pl = 0; pw += wstep; }
    return q;
  };
  v4f q[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) q[i] = fetch();
  for (long long w = wbeg; w < wend; w += wstep) {
    v4f acc = {0, 0, 0, 0};
#pragma unroll
    for (int l = 0; l < 9; ++l) {
      const v4f m = q[l];
      q[l] = fetch();
      acc += m;
    }
    if (NT) __builtin_nontemporal_store(acc, out + w * 64 + lane); else out[w * 64 + lane] = acc;
  }
}

// the fold's BACKWARD pattern: per window L matrices + dOut read, L matrices written (19 KiB per window at L = 9)
template <int NT>
__global__ __launch_bounds__(256) void kb(const v4f* __restrict__ mats, const v4f* __restrict__ dout, v4f* __restrict__ dm, long long Wn, int L) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
  for (long long w = wave; w < Wn; w += nw) {
    v4f q[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) {
      const v4f* src = mats + (w * L + l) * 64 + lane;
      if (NT) q[l] = __builtin_nontemporal_load(src); else q[l] = *src;
    }
    v4f g;
    if (NT) g = __builtin_nontemporal_load(dout + w * 64 + lane); else g = dout[w * 64 + lane];
#pragma unroll
    for (int l = 8; l >= 0; --l) {
      g += q[l];
      if (NT) __builtin_nontemporal_store(g, dm + (w * L + l) * 64 + lane); else dm[(w * L + l) * 64 + lane] = g;
    }
  }
}

int main() {
  const long long Wn = 692224; const int L = 9;
  v4f *mats, *out;
  if (hipMalloc(&mats, Wn * L * 1024) != hipSuccess || hipMalloc(&out, Wn * 1024) != hipSuccess) return 1;
  (void)hipMemset(mats, 0, Wn * L * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int variant = 0; variant < 4; ++variant)
    for (int wpc = 4; wpc <= 8; wpc += 2) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        const dim3 g(256 * wpc), b(256);
        if (variant == 0) hipLaunchKernelGGL((k<0, 0>), g, b, 0, 0, mats, out, Wn, L);
        if (variant == 1) hipLaunchKernelGGL((k<1, 0>), g, b, 0, 0, mats, out, Wn, L);
        if (variant == 2) hipLaunchKernelGGL((k<0, 1>), g, b, 0, 0, mats, out, Wn, L);
        if (variant == 3) hipLaunchKernelGGL((k<1, 1>), g, b, 0, 0, mats, out, Wn, L);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("%s%s %2d waves/CU: %.3f ms  %.2f TB/s\n", (variant & 1) ? "nontemporal " : "plain       ", (variant & 2) ? "contiguous range per wave" : "grid stride              ",
             4 * wpc, best, Wn * (L + 1) * 1024.0 / best * 1e-9);
    }
  {
    v4f* dm;
    if (hipMalloc(&dm, Wn * L * 1024) != hipSuccess) return 1;
    for (int nt = 0; nt < 2; ++nt)
      for (int wpc = 3; wpc <= 8; ++wpc) {
        if (wpc == 5 || wpc == 7) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          (void)hipEventRecord(e0);
          if (nt) hipLaunchKernelGGL(kb<1>, dim3(256 * wpc), dim3(256), 0, 0, mats, out, dm, Wn, L);
          else hipLaunchKernelGGL(kb<0>, dim3(256 * wpc), dim3(256), 0, 0, mats, out, dm, Wn, L);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1);
          best = ms < best ? ms : best;
        }
        printf("backward pattern %s %2d waves/CU: %.3f ms  %.2f TB/s\n", nt ? "nontemporal" : "plain      ", 4 * wpc, best,
               Wn * (2 * L + 1) * 1024.0 / best * 1e-9);
      }
  }
  return 0;
}
