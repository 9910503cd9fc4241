// Probe: a hipMemsetAsync recorded by stream capture, replayed several times with the buffer dirtied in between.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/memset_graph.hip -o gpurun_out/memset_graph && gpurun_out/memset_graph
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
__global__ void dirty(float* p, int n, float v) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = v; }
__global__ void touch(float* p) { if (threadIdx.x == 0) p[0] += 1.f; }
int main() {
  for (int n : {1000, 300000}) {
    float *buf, *other;
    hipMalloc(&buf, n * 4); hipMalloc(&other, 64); hipMemset(other, 0, 64);
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    hipMemsetAsync(buf, 0, (size_t)n * 4, st);
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, st, other);
    hipStreamEndCapture(st, &g);
    if (getenv("AUTOFREE")) hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagAutoFreeOnLaunch); else hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    std::vector<float> h(n);
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(dirty, dim3((n + 255) / 256), dim3(256), 0, st, buf, n, 5.f);
      hipGraphLaunch(ge, st);
      hipStreamSynchronize(st);
      hipMemcpy(h.data(), buf, n * 4, hipMemcpyDeviceToHost);
      float mx = 0; for (float v : h) mx = fabsf(v) > mx ? fabsf(v) : mx;
      printf("n %d replay %d max |buf| = %g\n", n, rep, mx);
    }
  }
  return 0;
}
