// Device helpers shared by the two register-resident EPS families for Q = 2 (eps_mfma.hip: bf16 matrix cores;
// eps_q2f32.hip: exact float32 matrix cores): lane-half exchanges, raw buffer descriptors, invariant division.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(2))) int q2_int2v;
typedef __attribute__((ext_vector_type(2))) float q2_f32x2;

constexpr int q2_ilog2(int v) { return v <= 1 ? 0 : 1 + q2_ilog2(v >> 1); }

// a <- [a.lo, b.lo], b <- [a.hi, b.hi]  (lo / hi = lanes 0-31 / 32-63): with a, b = the values a lane
// computed for the first / second lane half's role, a becomes the operand of set 0 (windows of lanes
// 0-31) and b the operand of set 1 (windows of lanes 32-63).
__device__ __forceinline__ void q2_swap_halves(float& a, float& b) {
  const q2_int2v r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
  a = __int_as_float(r[0]);
  b = __int_as_float(r[1]);
}

// the same exchange on the two registers as they stand (the builtin form makes the compiler copy one operand to a
// fresh register first when both come out of one packed instruction: a v_mov per swap).  The compiler's hazard
// recogniser does not see into inline assembly, so the wait states it would put around the instruction (2 between a
// vector write of an operand and the swap; 2 before a matrix instruction reads the result) are part of the text.
__device__ __forceinline__ void q2_swap_halves_inplace(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

// LDS hand-over inside one wave (writes of all lanes visible to the reads of all lanes)
__device__ __forceinline__ void q2_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one value against a pair: a single v_pk_mul_f32 (the broadcast is an operand selector, not an instruction)
__device__ __forceinline__ q2_f32x2 q2_bmul2(float s, q2_f32x2 v) { return q2_f32x2{s, s} * v; }

// unsigned 32-bit division by an invariant (Granlund-Montgomery, round-up variant):
//   q = (t + ((n - t) >> s1)) >> s2,  t = umulhi(M, n)
struct Q2FastDiv {
  unsigned M, s1, s2;
};
__device__ __forceinline__ unsigned q2_fdiv(unsigned n, const Q2FastDiv& d) {
  const unsigned t = __umulhi(d.M, n);
  return (t + ((n - t) >> d.s1)) >> d.s2;
}
static inline Q2FastDiv q2_make_fastdiv(unsigned d) {
  Q2FastDiv f;
  if (d <= 1) {
    f.M = 0; f.s1 = 0; f.s2 = 0;
    return f;
  }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.M = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s1 = 1;
  f.s2 = l - 1;
  return f;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t q2_make_rsrc(const void* ptr, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000);
}
