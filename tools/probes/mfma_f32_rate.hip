// Probe: sustained rate of v_mfma_f32_32x32x2_f32 as the bigcore kernels issue it: NACC accumulators per wave, the B operand
// of every MFMA the product of two registers formed just before it (VMUL = 1) or a loop-invariant register (0), 1 / 2 / 3
// waves per SIMD.  Peak: 64 cycles per MFMA and SIMD = 157 TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_rate.hip -o tools/probes/mfma_f32_rate && tools/probes/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, int VMUL>
__global__ void rate(float* out, int iters, float a0, float b0) {
  f32x16 c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) c[i][v] = 0.f;
  float a = a0 + threadIdx.x, hi[NACC], tab[8];
#pragma unroll
  for (int i = 0; i < NACC; ++i) hi[i] = b0 + i;
#pragma unroll
  for (int t = 0; t < 8; ++t) tab[t] = b0 * (t + 1);
  for (int it = 0; it < iters; ++it) {
    if (VMUL == 2) {   // the products of half a block first, then its MFMAs back to back
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        float bop[NACC][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < NACC; ++i) bop[i][t] = hi[i] * tab[4 * hlf + t];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bop[i][t], c[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        const float b = VMUL ? hi[i] * tab[t] : tab[t];
        c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
      }
    }
    if (VMUL) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) hi[i] += 1e-9f;
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) r += c[i][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// As eps_bigcore_dcore_k feeds it: per k-step 2 + 4 operands, each the product of two (three) ds_read_b32 values of a
// table in LDS, then 8 MFMAs (2 x 4 accumulators).
__global__ __launch_bounds__(512) void rate_lds(float* out, int iters, int stride) {
  extern __shared__ float tb[];
  for (int e = threadIdx.x; e < 128 * 125; e += blockDim.x) tb[e] = 1.0f + 1e-6f * e;
  __syncthreads();
  const int lane = threadIdx.x & 63, il = lane & 31, kk = lane >> 5;
  f32x16 c[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) c[a][b][v] = 0.f;
  int oa_lo[2], oa_hi[2], ob_lo[4], ob_hi[4], ob_dy[4];
#pragma unroll
  for (int a = 0; a < 2; ++a) { oa_lo[a] = il; oa_hi[a] = 64 + a; }
#pragma unroll
  for (int b = 0; b < 4; ++b) { ob_lo[b] = 80 + (il & 15); ob_hi[b] = 96 + b; ob_dy[b] = 112 + (il & 7); }
  for (int it = 0; it < iters; ++it) {
    const float* tw = tb + kk * stride;
#pragma unroll 2
    for (int ks = 0; ks < 64; ++ks, tw += 2 * stride) {
      float pa[2], pz[4];
#pragma unroll
      for (int a = 0; a < 2; ++a) pa[a] = tw[oa_lo[a]] * tw[oa_hi[a]];
#pragma unroll
      for (int b = 0; b < 4; ++b) pz[b] = tw[ob_lo[b]] * tw[ob_hi[b]] * tw[ob_dy[b]];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) c[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[a], pz[b], c[a][b], 0, 0, 0);
    }
  }
  float r = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) r += c[a][b][a + b];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

void run_lds(float* d) {
  const int iters = 40, stride = 121;
  (void)hipFuncSetAttribute((const void*)rate_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 125 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_lds, dim3(256), dim3(512), 128 * 125 * 4, 0, d, 2, stride);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(rate_lds, dim3(256), dim3(512), 128 * 125 * 4, 0, d, iters, stride);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mf = 256.0 * 8 * iters * 64 * 8;
  printf("operands from LDS tables as in the dCore kernel (16 ds_read_b32 + 10 v_mul per 8 MFMAs), 2 waves per SIMD: %.1f TFLOP/s (%.2f of 157)\n",
         mf * 4096.0 / ms * 1e-9, mf * 4096.0 / ms * 1e-9 / 157.3);
}


// Variants of the same loop.  MODE 1: the offsets a real (4^4 rows x 4^5 * 6 columns) tile has - 16 / 2 distinct row
// offsets and ~6 / 1 / 6 distinct column offsets per 32 lanes; MODE 2: as 1 without the reads inside the loop (the
// ceiling); MODE 3: as 1 with the reads of the next k-step issued before this step's MFMAs; MODE 4: as 3, tables of
// two windows interleaved so that one ds_read_b64 serves two k-steps; MODE 5: as 4 with four windows and ds_read_b128.
template <int MODE>
__global__ __launch_bounds__(512) void rate_lds2(float* out, int iters, int stride) {
  extern __shared__ float tb[];
  for (int e = threadIdx.x; e < 128 * 125; e += blockDim.x) tb[e] = 1.0f + 1e-6f * (e & 1023);
  __syncthreads();
  const int lane = threadIdx.x & 63, il = lane & 31, kk = lane >> 5, wv = threadIdx.x >> 6;
  f32x16 c[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) c[a][b][v] = 0.f;
  int off[16];
  {   // bit set in `real`: that family of offsets as a real tile has it; clear: as the first probe has it
    const int real = MODE == 0 || MODE == 6 ? 0 : MODE == 7 ? stride >> 16 : 31;
    stride &= 0xffff;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      off[2 * a] = real & 1 ? (il & 15) : il;
      off[2 * a + 1] = real & 2 ? 16 + 2 * ((wv >> 2) * 2 + a) + (il >> 4) : 64 + a;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = ((wv & 3) * 4 + b) * 32 + il, bb = col / 6;
      off[4 + 3 * b] = real & 4 ? 32 + (bb & 63) : 80 + (il & 15);
      off[5 + 3 * b] = real & 8 ? 96 + (bb >> 6) : 96 + b;
      off[6 + 3 * b] = real & 16 ? 112 + col % 6 : 112 + (il & 7);
    }
  }
  constexpr int G = MODE == 4 ? 2 : MODE == 5 ? 4 : 1;   // windows of one lane half interleaved per table entry
  float r_[G][16];
  auto rd = [&](int ks) {   // the factors of k-steps ks .. ks + G - 1
    if constexpr (G == 1) {
      const float* tw = tb + (2 * ks + kk) * stride;
#pragma unroll
      for (int i = 0; i < 16; ++i) r_[0][i] = tw[off[i]];
    } else if constexpr (G == 2) {
      const float* tw = tb + (2 * ks + kk * G) * stride;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float2 v = *reinterpret_cast<const float2*>(tw + 2 * off[i]);
        r_[0][i] = v.x; r_[1][i] = v.y;
      }
    } else {
      const float* tw = tb + (2 * ks + kk * G) * stride;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(tw + 4 * off[i]);
        r_[0][i] = v.x; r_[1][i] = v.y; r_[2][i] = v.z; r_[3][i] = v.w;
      }
    }
  };
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 2) rd(0);
#pragma unroll 2
    for (int ks = 0; ks < 64; ks += G) {
      if (MODE <= 1) rd(ks);
      float pa[G][2], pz[G][4];
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int a = 0; a < 2; ++a) pa[g][a] = r_[g][2 * a] * r_[g][2 * a + 1];
#pragma unroll
        for (int b = 0; b < 4; ++b) pz[g][b] = r_[g][4 + 3 * b] * r_[g][5 + 3 * b] * r_[g][6 + 3 * b];
      }
      if (MODE >= 3) { rd(ks + G < 64 ? ks + G : ks); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) c[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[g][a], pz[g][b], c[a][b], 0, 0, 0);
      if (MODE >= 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) r += c[a][b][a + b];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run_lds2(float* d, const char* what, int stride = 121) {
  const int iters = 40;
  (void)hipFuncSetAttribute((const void*)rate_lds2<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 125 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_lds2<MODE>, dim3(256), dim3(512), 128 * 125 * 4, 0, d, 2, stride);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(rate_lds2<MODE>, dim3(256), dim3(512), 128 * 125 * 4, 0, d, iters, stride);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mf = 256.0 * 8 * iters * 64 * 8;
  stride &= 0xffff;
  printf("LDS-table loop, %s, window stride %d floats: %.1f TFLOP/s (%.2f of 157)\n", what, stride, mf * 4096.0 / ms * 1e-9, mf * 4096.0 / ms * 1e-9 / 157.3);
}

template <int NACC, int VMUL>
void run(float* d) {
  const int iters = 4000;
  for (int wps = 1; wps <= 3; ++wps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<NACC, VMUL>), dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0f, 1e-6f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((rate<NACC, VMUL>), dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f, 1e-6f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = 256.0 * wps * 4 * iters * 8 * NACC;
    printf("%d accumulators, B %s, %d wave(s) per SIMD: %.1f TFLOP/s (%.2f of 157)\n", NACC, VMUL == 2 ? "= 8 v_mul, then 8 MFMAs" : VMUL ? "= v_mul just before" : "invariant", wps,
           mf * 4096.0 / ms * 1e-9, mf * 4096.0 / ms * 1e-9 / 157.3);
  }
}
int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 3 * 512 * sizeof(float));
  run<1, 0>(d); run<2, 0>(d); run<2, 1>(d); run<4, 1>(d); run<2, 2>(d); run_lds(d);
  run_lds2<1>(d, "a real tile's offsets");
  run_lds2<1>(d, "a real tile's offsets", 96);   // the bank offset between the lane halves: no effect
  for (int bit = 0; bit < 5; ++bit) {
    static const char* fam[5] = {"row low digits real", "row high digits real", "column low digits real", "column high digits real", "output index real"};
    run_lds2<7>(d, fam[bit], 121 | (1 << bit) << 16);
  }
  run_lds2<0>(d, "the first probe's offsets"); run_lds2<6>(d, "the first probe's offsets, reads issued before the MFMAs");
  run_lds2<2>(d, "no reads inside the loop");
  run_lds2<3>(d, "real offsets, next step's reads issued before the MFMAs");
  run_lds2<4>(d, "two windows per table entry, ds_read_b64"); run_lds2<5>(d, "four windows per table entry, ds_read_b128");
  return 0;
}
