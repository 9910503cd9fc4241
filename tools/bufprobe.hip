// Probe: how does buffer_load_dwordx4 range-check a raw buffer (stride 0) on gfx950 when only the
// tail of the 16-byte access lies beyond num_records?  (build: hipcc --offload-arch=gfx950 -O2)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const char* p, unsigned nbytes, unsigned off, unsigned soff, unsigned* out) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0);
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
int main() {
  unsigned h[32];
  for (int i = 0; i < 32; ++i) h[i] = 0x1000 + i;
  unsigned *d, *o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (unsigned soff : {0u, 16u, 48u}) {
    for (unsigned off : {32u, 52u, 56u, 60u, 64u}) {   // num_records = 64 bytes (16 dwords)
      if (off < soff) continue;
      hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, (const char*)d, 64u, off - soff, soff, o);
      unsigned r[4]; hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
      printf("voffset %u + soffset %u: %x %x %x %x\n", off - soff, soff, r[0], r[1], r[2], r[3]);
    }
  }
  return 0;
}
