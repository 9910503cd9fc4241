// Probe: operand / result layout of v_mfma_f64_4x4x4_4b_f64 (4 independent blocks of D(4x4) += A(4x4) B(4x4), one double per lane
// and operand).  Hypothesis checked (round 5, after reading the dependence structure this probe prints when it is wrong):
//   block = (lane / 4) % 4 for all three;
//   A lane s holds A_blk[i = s % 4][k = s / 16];  B lane s holds B_blk[k = s / 16][j = s % 4];
//   D lane d holds D_blk[i = d / 16][j = d % 4].
// Also its rate next to v_mfma_f64_16x16x4_f64 (tools/probes/mfma_f64_rate.hip): 67.8-75.6 against 47.9 TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_4x4x4_layout.hip -o tools/probes/mfma_f64_4x4x4_layout && tools/probes/mfma_f64_4x4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void k(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}

int main() {
  double hA[64], hB[64], hD[64], *dA, *dB, *dD;
  (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dD, 512);
  // random-ish distinct values; reference under the hypothesis
  for (int l = 0; l < 64; ++l) { hA[l] = 1.0 + 0.37 * l + 0.011 * l * l; hB[l] = 2.0 - 0.19 * l + 0.007 * l * l; }
  (void)hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  (void)hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int d = 0; d < 64; ++d) {
    const int blk = (d / 4) % 4, i = d / 16, j = d % 4;
    double want = 0;
    for (int kk = 0; kk < 4; ++kk) want += hA[kk * 16 + blk * 4 + i] * hB[kk * 16 + blk * 4 + j];
    worst = fmax(worst, fabs(want - hD[d]) / fabs(want));
  }
  printf("hypothesis (blk = (lane/4)%%4; A: i = lane%%4, k = lane/16; B: j = lane%%4, k = lane/16; D: i = lane/16, j = lane%%4): max rel err %.3g -> %s\n",
         worst, worst < 1e-12 ? "CONFIRMED" : "WRONG");
  if (worst >= 1e-12) {   // print the dependence structure for a manual read
    for (int s = 0; s < 64; s += 1) {
      double a[64], b[64];
      for (int l = 0; l < 64; ++l) { b[l] = l == s ? 1.0 : 0.0; a[l] = 1.0; }
      (void)hipMemcpy(dA, a, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, b, 512, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
      (void)hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
      printf("B lane %2d feeds D lanes:", s);
      for (int d = 0; d < 64; ++d) if (hD[d] != 0) printf(" %d", d);
      printf("\n");
    }
  }
  return 0;
}
