// Probe: register layout of v_mfma_f64_16x16x4_f64 on gfx950 (which D[i][j] each lane/register holds).
// hipcc --offload-arch=gfx950 -O2 tools/mfma64probe.hip -o tools/mfma64probe && tools/mfma64probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) double f64x4;
__global__ void probe(double* out) {
  const int l = threadIdx.x;
  // assumption under test: A lane l = A[i = l%16][k = l/16], B lane l = B[k = l/16][j = l%16]
  const int i = l % 16, k = l / 16;
  const double a = (i + 1) * (k == 2 ? 1.0 : 0.0);          // only k = 2 contributes
  const double b = (l % 16 + 1) * 100.0 * (k == 2 ? 1.0 : 0.0);
  f64x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[l * 4 + v] = d[v];
}
int main() {
  double* d;
  hipMalloc(&d, 256 * sizeof(double));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  double h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) {
    printf("lane %2d:", l);
    for (int v = 0; v < 4; ++v) {
      const int val = (int)(h[l * 4 + v] / 100.0 + 0.5);   // (i+1)*(j+1)
      printf("  v%d=%4d", v, val);
    }
    printf("\n");
  }
  return 0;
}
