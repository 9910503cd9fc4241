// Latency-first gradient all-reduce over peer-mapped buffers: the reference has no collective
// at all; the build's one collective is the mean of the parameter gradients per step, and its messages are small (BASELINE
// cfg2: 58 KB in bf16, cfg3a: 7.5 MB) - latency-bound, not bandwidth-bound.  A ring pays 2 (P - 1) hops and drives one xGMI
// link per direction per hop; on MI355X's fully connected 8-GPU node every rank can instead READ its P - 1 peers' buffers
// over its P - 1 links at once (one-shot direct all-reduce).  This file is that collective, without RCCL:
//   * every rank owns one UNCACHED device block (hipExtMallocWithFlags) = [flag lines] [staging 0] [staging 1] [counters],
//     exported once as a hipIpcMemHandle and opened by every peer (ranks of one node; two ranks on one GPU work too);
//   * one kernel per step and rank: copy the rank's gradients into staging[step & 1], publish the step number in every
//     peer's flag line, wait (bounded) for the peers' step numbers in its own flag lines, then read every rank's
//     staging in RANK ORDER, sum in float32 / float64, scale, and write the result over the rank's gradients in place -
//     every rank adds the same numbers in the same order: bitwise identical results on all ranks;
//   * staging is double-buffered by the parity of the step, so no second barrier is needed: a rank overwrites
//     staging[s & 1] at step s + 2, which its peers allow by publishing step s + 1 - after their step-s reads;
//   * the step counter lives in device memory (the launch is replayable from a HIP graph: no argument changes);
//   * a wait that does not complete within ~2 s sets an error word instead of hanging the queue (dctn_ar_status);
//   * TWO-SHOT form for large buckets (cfg3a: 7.5 MB): reading every peer's whole buffer moves (P - 1) N bytes into every
//     rank; instead rank r reduces only chunk r (N / P elements, read from every rank's staging: (P - 1) N / P bytes),
//     leaves the scaled result in its `result` area and publishes a second flag; every rank then copies chunk j from rank
//     j's result area (another (P - 1) N / P bytes).  Same sums in the same (rank) order, each formed once: bitwise the
//     one-shot values, identical on all ranks.  `result` is double-buffered like the staging.  Chosen by
//     dctn_ar_allreduce for world >= 4 and >= 512 KiB (a guess - this build has never run across xGMI);
//     dctn_ar_allreduce_algo forces either form (the two-rank tests run both).
#include "common.h"

#include <string.h>

namespace {

constexpr int AR_MAX_WORLD = 16;
constexpr int AR_FLAG_STRIDE = 16;          // ints per flag line (64 bytes: one line per peer)
constexpr int AR_THREADS = 512;
constexpr int AR_MAX_BLOCKS = 64;

struct ArState {
  int world, rank;
  size_t max_bytes, block_bytes;
  unsigned char* block;                      // this rank's block
  unsigned char* peer[AR_MAX_WORLD];         // every rank's block as mapped here (peer[rank] == block)
  bool opened[AR_MAX_WORLD];
};

struct ArP {
  unsigned char* peer[AR_MAX_WORLD];
  int world, rank;
  size_t max_bytes;
  long long n;        // elements
  double scale;   // 1 / world (or 1) in double: float64 buffers get the mean to their own precision for any world size
};

// block layout
// block layout: [flag lines: inputs staged | flag lines: chunk reduced] [staging 0] [staging 1] [result 0] [result 1] [counters]
__host__ __device__ inline size_t ar_flags_bytes() { return (size_t)2 * AR_MAX_WORLD * AR_FLAG_STRIDE * sizeof(int); }
__host__ __device__ inline size_t ar_result_off(size_t max_bytes) { return ar_flags_bytes() + 2 * max_bytes; }
__host__ __device__ inline size_t ar_counters_off(size_t max_bytes) { return ar_flags_bytes() + 4 * max_bytes; }
// counters: [0] step, [1] error word, [2] workgroups done copying, [3] workgroups finished, [4] workgroups done reducing

template <typename T> struct ArAcc { typedef float type; };
template <> struct ArAcc<double> { typedef double type; };

// every rank's flag in line set `which` (0: inputs staged, 1: chunk reduced) of this rank's block reaches step + 1
__device__ __forceinline__ void ar_wait(volatile int* my_flags, int which, int world, int step, int* counters) {
  const int tid = threadIdx.x;
  if (tid < world) {
    const long long t0 = wall_clock64();
    bool ok = true;
    while (__hip_atomic_load(const_cast<int*>(&my_flags[(which * AR_MAX_WORLD + tid) * AR_FLAG_STRIDE]), __ATOMIC_ACQUIRE,
                             __HIP_MEMORY_SCOPE_SYSTEM) < step + 1) {
      if (wall_clock64() - t0 > 200000000LL) { ok = false; break; }   // ~2 s of the 100 MHz clock
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) __hip_atomic_store(&counters[1], 1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}

// the LAST workgroup to bump counters[which_counter] publishes step + 1 in line set `which` of every rank's block
__device__ __forceinline__ void ar_publish(const ArP& p, int which, int which_counter, int step, int* counters, int* s_last) {
  const int tid = threadIdx.x;
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    const int done = __hip_atomic_fetch_add(&counters[which_counter], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = done == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (*s_last && tid < p.world) {
    int* f = reinterpret_cast<int*>(p.peer[tid]) + (which * AR_MAX_WORLD + p.rank) * AR_FLAG_STRIDE;
    __hip_atomic_store(f, step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// 16 bytes of T per lane and access (the rates the HBM and xGMI paths are quoted at are for 16 B per lane; one element
// per lane - 2 B for bf16 - left most of the path idle).  VEC = false: element accesses (a buffer that is not 16-byte
// aligned: a view into a larger gradient buffer at an odd offset).
template <typename T> struct ArVec { static constexpr int VE = 16 / (int)sizeof(T); };
template <typename T>
__device__ __forceinline__ void ar_ld16(const T* src, T (&v)[ArVec<T>::VE], bool nt) {
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  const u4 q = nt ? __builtin_nontemporal_load(reinterpret_cast<const u4*>(src)) : *reinterpret_cast<const u4*>(src);
  __builtin_memcpy(v, &q, 16);
}
template <typename T>
__device__ __forceinline__ void ar_st16(T* dst, const T (&v)[ArVec<T>::VE]) {
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  u4 q;
  __builtin_memcpy(&q, v, 16);
  *reinterpret_cast<u4*>(dst) = q;
}

// out[i] = in[i] for i in [lo, hi): 16-byte pieces where VEC (lo a multiple of VE), elements for the rest
template <typename T, bool VEC>
__device__ __forceinline__ void ar_copy(T* __restrict__ out, const T* __restrict__ in, long long lo, long long hi, bool nt) {
  constexpr int VE = ArVec<T>::VE;
  const int tid = threadIdx.x;
  long long i = lo;
  if (VEC) {
    const long long nv = (hi - lo) / VE;
    for (long long v = tid; v < nv; v += AR_THREADS) {
      T t[VE];
      ar_ld16(in + lo + v * VE, t, nt);
      ar_st16(out + lo + v * VE, t);
    }
    i = lo + nv * VE;
  }
  for (long long e = i + tid; e < hi; e += AR_THREADS) out[e] = nt ? __builtin_nontemporal_load(&in[e]) : in[e];
}

// out0[i] (and out1[i]) = scale * sum over ranks r, in rank order, of staging_r[i] for i in [lo, hi)
template <typename T, bool VEC>
__device__ __forceinline__ void ar_sum(const ArP& p, size_t stage_off, T* __restrict__ out0, T* __restrict__ out1, long long lo,
                                       long long hi) {
  typedef typename ArAcc<T>::type A;
  constexpr int VE = ArVec<T>::VE;
  const int tid = threadIdx.x;
  const A scale = (A)p.scale;
  long long i = lo;
  if (VEC) {
    const long long nv = (hi - lo) / VE;
    for (long long v = tid; v < nv; v += AR_THREADS) {
      A acc[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] = 0;
      for (int r = 0; r < p.world; ++r) {
        T t[VE];
        ar_ld16(reinterpret_cast<const T*>(p.peer[r] + stage_off) + lo + v * VE, t, true);
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += (A)t[e];
      }
      T o[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] = (T)(acc[e] * scale);
      ar_st16(out0 + lo + v * VE, o);
      if (out1) ar_st16(out1 + lo + v * VE, o);
    }
    i = lo + nv * VE;
  }
  for (long long e = i + tid; e < hi; e += AR_THREADS) {
    A acc = 0;
    for (int r = 0; r < p.world; ++r) acc += (A)__builtin_nontemporal_load(&reinterpret_cast<const T*>(p.peer[r] + stage_off)[e]);
    const T o = (T)(acc * scale);
    out0[e] = o;
    if (out1) out1[e] = o;
  }
}

// [lo, hi) of workgroup `blk` of `nblk` over [0, n), the cut points multiples of `unit` elements
__device__ __forceinline__ void ar_range(long long n, int unit, int blk, int nblk, long long& lo, long long& hi) {
  const long long units = (n + unit - 1) / unit, per = (units + nblk - 1) / nblk;
  lo = (long long)blk * per * unit;
  hi = lo + per * unit;
  if (lo > n) lo = n;
  if (hi > n) hi = n;
}

template <typename T, bool TWO, bool VEC>
__global__ __launch_bounds__(AR_THREADS) void dctn_ar_k(T* __restrict__ buf, ArP p) {
  constexpr int VE = ArVec<T>::VE;
  unsigned char* mine = p.peer[p.rank];
  int* counters = reinterpret_cast<int*>(mine + ar_counters_off(p.max_bytes));
  volatile int* my_flags = reinterpret_cast<volatile int*>(mine);
  const int tid = threadIdx.x;
  __shared__ int s_step, s_last;
  if (tid == 0) s_step = __hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int step = s_step, slot = step & 1;
  long long lo, hi;
  ar_range(p.n, VE, (int)blockIdx.x, (int)gridDim.x, lo, hi);
  // ---- 1. this rank's values -> its staging buffer (uncached memory: the stores go to memory)
  const size_t stage_off = ar_flags_bytes() + (size_t)slot * p.max_bytes;
  ar_copy<T, VEC>(reinterpret_cast<T*>(mine + stage_off), buf, lo, hi, false);
  ar_publish(p, 0, 2, step, counters, &s_last);   // every workgroup has copied: step + 1 in every rank's line of this rank
  // ---- 2. wait for every rank's step + 1 (bounded)
  ar_wait(my_flags, 0, p.world, step, counters);
  if (!TWO) {
    // ---- 3. the sum in rank order, scaled, over the rank's own values
    ar_sum<T, VEC>(p, stage_off, buf, (T*)nullptr, lo, hi);
  } else {
    // ---- 3a. this rank's chunk (chunks are cut at multiples of VE elements): the sum in rank order, scaled, into its
    //          result area (and its own values)
    const long long cs = ((p.n + p.world - 1) / p.world + VE - 1) / VE * VE;
    const size_t res_off = ar_result_off(p.max_bytes) + (size_t)slot * p.max_bytes;
    auto chunk_part = [&](int j, long long& a, long long& b) {   // this workgroup's part of rank j's chunk
      const long long j0 = (long long)j * cs < p.n ? (long long)j * cs : p.n, j1 = j0 + cs < p.n ? j0 + cs : p.n;
      long long l, h;
      ar_range(j1 - j0, VE, (int)blockIdx.x, (int)gridDim.x, l, h);
      a = j0 + l;
      b = j0 + h;
    };
    long long clo, chi;
    chunk_part(p.rank, clo, chi);
    ar_sum<T, VEC>(p, stage_off, reinterpret_cast<T*>(mine + res_off), buf, clo, chi);
    ar_publish(p, 1, 4, step, counters, &s_last);
    ar_wait(my_flags, 1, p.world, step, counters);
    // ---- 3b. every other rank's chunk from that rank's result area
    for (int j = 0; j < p.world; ++j) {
      if (j == p.rank) continue;
      long long jlo, jhi;
      chunk_part(j, jlo, jhi);
      ar_copy<T, VEC>(buf, reinterpret_cast<const T*>(p.peer[j] + res_off), jlo, jhi, true);
    }
  }
  // ---- 4. the last workgroup to finish advances the step
  __syncthreads();
  if (tid == 0) {
    const int fin = __hip_atomic_fetch_add(&counters[3], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (fin == (int)gridDim.x - 1) {
      __hip_atomic_store(&counters[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&counters[3], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&counters[4], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&counters[0], step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

extern "C" {

size_t dctn_ar_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

int dctn_ar_create(int world, int rank, size_t max_bytes, void** state_out) {
  if (!state_out) return DCTN_ERR_NULL;
  if (world < 1 || world > AR_MAX_WORLD || rank < 0 || rank >= world || max_bytes == 0) return DCTN_ERR_BAD_SHAPE;
  ArState* st = new ArState();
  st->world = world; st->rank = rank;
  st->max_bytes = (max_bytes + 255) & ~(size_t)255;
  st->block_bytes = ar_counters_off(st->max_bytes) + 256;
  for (int r = 0; r < AR_MAX_WORLD; ++r) { st->peer[r] = nullptr; st->opened[r] = false; }
  void* ptr = nullptr;
  if (hipExtMallocWithFlags(&ptr, st->block_bytes, hipDeviceMallocUncached) != hipSuccess || !ptr) {
    delete st;
    return DCTN_ERR_LAUNCH;
  }
  if (hipMemset(ptr, 0, st->block_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(ptr);
    delete st;
    return DCTN_ERR_LAUNCH;
  }
  st->block = (unsigned char*)ptr;
  st->peer[rank] = st->block;
  *state_out = st;
  return DCTN_OK;
}

int dctn_ar_export(void* state, void* handle_out) {
  if (!state || !handle_out) return DCTN_ERR_NULL;
  ArState* st = (ArState*)state;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, st->block) != hipSuccess) return DCTN_ERR_LAUNCH;
  memcpy(handle_out, &h, sizeof(h));
  return DCTN_OK;
}

int dctn_ar_connect(void* state, const void* handles) {
  if (!state || !handles) return DCTN_ERR_NULL;
  ArState* st = (ArState*)state;
  const unsigned char* hb = (const unsigned char*)handles;
  for (int r = 0; r < st->world; ++r) {
    if (r == st->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, hb + (size_t)r * sizeof(h), sizeof(h));
    void* ptr = nullptr;
    if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !ptr) return DCTN_ERR_LAUNCH;
    st->peer[r] = (unsigned char*)ptr;
    st->opened[r] = true;
  }
  return DCTN_OK;
}

int dctn_ar_allreduce_algo(void* state, void* buf, int64_t n, int dtype, int average, int algorithm, void* stream) {
  if (!state || !buf) return DCTN_ERR_NULL;
  if (algorithm < 0 || algorithm > 2) return DCTN_ERR_UNSUPPORTED;
  ArState* st = (ArState*)state;
  if (n < 0 || (size_t)n * dtype_size(dtype) > st->max_bytes) return DCTN_ERR_BAD_SHAPE;
  if (dtype != DCTN_F32 && dtype != DCTN_F64 && dtype != DCTN_BF16) return DCTN_ERR_BAD_DTYPE;
  for (int r = 0; r < st->world; ++r)
    if (!st->peer[r]) return DCTN_ERR_NULL;   // dctn_ar_connect has not run
  if (n == 0) return DCTN_OK;
  ArP p;
  for (int r = 0; r < AR_MAX_WORLD; ++r) p.peer[r] = st->peer[r];
  p.world = st->world; p.rank = st->rank; p.max_bytes = st->max_bytes; p.n = n;
  p.scale = average ? 1.0 / (double)st->world : 1.0;
  long long blocks = ((long long)n * (long long)dtype_size(dtype) + 16383) / 16384;
  if (blocks < 1) blocks = 1;
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  hipStream_t s = (hipStream_t)stream;
  const bool two = algorithm == 2 || (algorithm == 0 && st->world >= 4 && (size_t)n * dtype_size(dtype) >= (512u << 10));
  // 16-byte accesses need a 16-byte aligned buffer (the staging / result areas are: the block and max_bytes are 256-byte
  // aligned); a view at an odd offset takes the element form
  const bool vec = ((uintptr_t)buf % 16) == 0;
#define AR_LAUNCH(TT)                                                                                                        \
  do {                                                                                                                       \
    if (two && vec) hipLaunchKernelGGL((dctn_ar_k<TT, true, true>), dim3((unsigned)blocks), dim3(AR_THREADS), 0, s, (TT*)buf, p);   \
    else if (two) hipLaunchKernelGGL((dctn_ar_k<TT, true, false>), dim3((unsigned)blocks), dim3(AR_THREADS), 0, s, (TT*)buf, p);    \
    else if (vec) hipLaunchKernelGGL((dctn_ar_k<TT, false, true>), dim3((unsigned)blocks), dim3(AR_THREADS), 0, s, (TT*)buf, p);    \
    else hipLaunchKernelGGL((dctn_ar_k<TT, false, false>), dim3((unsigned)blocks), dim3(AR_THREADS), 0, s, (TT*)buf, p);            \
  } while (0)
  switch (dtype) {
    case DCTN_F32: AR_LAUNCH(float); break;
    case DCTN_F64: AR_LAUNCH(double); break;
    default: AR_LAUNCH(bf16_t); break;
  }
#undef AR_LAUNCH
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel(two ? "allreduce_direct_two_shot" : "allreduce_direct");
  return DCTN_OK;
}

int dctn_ar_allreduce(void* state, void* buf, int64_t n, int dtype, int average, void* stream) {
  return dctn_ar_allreduce_algo(state, buf, n, dtype, average, 0, stream);
}

// 0: every wait so far completed; r + 1: a wait for rank r timed out (synchronises the device)
int dctn_ar_status(void* state) {
  if (!state) return DCTN_ERR_NULL;
  ArState* st = (ArState*)state;
  if (hipDeviceSynchronize() != hipSuccess) return DCTN_ERR_LAUNCH;
  int err = 0;
  if (hipMemcpy(&err, st->block + ar_counters_off(st->max_bytes) + sizeof(int), sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
    return DCTN_ERR_LAUNCH;
  return err;
}

int dctn_ar_destroy(void* state) {
  if (!state) return DCTN_OK;
  ArState* st = (ArState*)state;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < st->world; ++r)
    if (st->opened[r]) (void)hipIpcCloseMemHandle(st->peer[r]);
  (void)hipFree(st->block);
  delete st;
  return DCTN_OK;
}

}  // extern "C"
