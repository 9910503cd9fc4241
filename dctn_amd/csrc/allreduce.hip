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
//   * a wait that does not complete within ~2 s sets an error word instead of hanging the queue (dctn_ar_status).
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
  float scale;
};

// block layout
__host__ __device__ inline size_t ar_flags_bytes() { return (size_t)AR_MAX_WORLD * AR_FLAG_STRIDE * sizeof(int); }
__host__ __device__ inline size_t ar_counters_off(size_t max_bytes) { return ar_flags_bytes() + 2 * max_bytes; }
// counters: [0] step, [1] error word, [2] workgroups done copying, [3] workgroups finished

template <typename T> struct ArAcc { typedef float type; };
template <> struct ArAcc<double> { typedef double type; };

template <typename T>
__global__ __launch_bounds__(AR_THREADS) void dctn_ar_k(T* __restrict__ buf, ArP p) {
  unsigned char* mine = p.peer[p.rank];
  int* counters = reinterpret_cast<int*>(mine + ar_counters_off(p.max_bytes));
  volatile int* my_flags = reinterpret_cast<volatile int*>(mine);
  const int tid = threadIdx.x;
  __shared__ int s_step, s_last;
  if (tid == 0) s_step = __hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int step = s_step, slot = step & 1;
  const long long per = (p.n + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < p.n ? lo + per : p.n;
  // ---- 1. this rank's values -> its staging buffer (uncached memory: the stores go to memory)
  T* stage = reinterpret_cast<T*>(mine + ar_flags_bytes() + (size_t)slot * p.max_bytes);
  for (long long i = lo + tid; i < hi; i += AR_THREADS) stage[i] = buf[i];
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    const int done = __hip_atomic_fetch_add(&counters[2], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = done == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (s_last && tid < p.world) {   // every workgroup has copied: publish step + 1 in every rank's flag line of this rank
    int* f = reinterpret_cast<int*>(p.peer[tid]) + p.rank * AR_FLAG_STRIDE;
    __hip_atomic_store(f, step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- 2. wait for every rank's step + 1 (bounded: ~2 s of the 100 MHz clock)
  if (tid < p.world) {
    const long long t0 = wall_clock64();
    bool ok = true;
    while (__hip_atomic_load(const_cast<int*>(&my_flags[tid * AR_FLAG_STRIDE]), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < step + 1) {
      if (wall_clock64() - t0 > 200000000LL) { ok = false; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) __hip_atomic_store(&counters[1], 1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // ---- 3. the sum in rank order, scaled, over the rank's own values
  typedef typename ArAcc<T>::type A;
  for (long long i = lo + tid; i < hi; i += AR_THREADS) {
    A acc = 0;
    for (int r = 0; r < p.world; ++r) {
      const T* src = reinterpret_cast<const T*>(p.peer[r] + ar_flags_bytes() + (size_t)slot * p.max_bytes);
      acc += (A)__builtin_nontemporal_load(&src[i]);
    }
    buf[i] = (T)(acc * (A)p.scale);
  }
  // ---- 4. the last workgroup to finish advances the step
  __syncthreads();
  if (tid == 0) {
    const int fin = __hip_atomic_fetch_add(&counters[3], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (fin == (int)gridDim.x - 1) {
      __hip_atomic_store(&counters[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&counters[3], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

int dctn_ar_allreduce(void* state, void* buf, int64_t n, int dtype, int average, void* stream) {
  if (!state || !buf) return DCTN_ERR_NULL;
  ArState* st = (ArState*)state;
  if (n < 0 || (size_t)n * dtype_size(dtype) > st->max_bytes) return DCTN_ERR_BAD_SHAPE;
  if (dtype != DCTN_F32 && dtype != DCTN_F64 && dtype != DCTN_BF16) return DCTN_ERR_BAD_DTYPE;
  for (int r = 0; r < st->world; ++r)
    if (!st->peer[r]) return DCTN_ERR_NULL;   // dctn_ar_connect has not run
  if (n == 0) return DCTN_OK;
  ArP p;
  for (int r = 0; r < AR_MAX_WORLD; ++r) p.peer[r] = st->peer[r];
  p.world = st->world; p.rank = st->rank; p.max_bytes = st->max_bytes; p.n = n;
  p.scale = average ? 1.0f / (float)st->world : 1.0f;
  long long blocks = ((long long)n * (long long)dtype_size(dtype) + 16383) / 16384;
  if (blocks < 1) blocks = 1;
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case DCTN_F32: hipLaunchKernelGGL(dctn_ar_k<float>, dim3((unsigned)blocks), dim3(AR_THREADS), 0, s, (float*)buf, p); break;
    case DCTN_F64: hipLaunchKernelGGL(dctn_ar_k<double>, dim3((unsigned)blocks), dim3(AR_THREADS), 0, s, (double*)buf, p); break;
    default: hipLaunchKernelGGL(dctn_ar_k<bf16_t>, dim3((unsigned)blocks), dim3(AR_THREADS), 0, s, (bf16_t*)buf, p); break;
  }
  DCTN_CHECK_LAUNCH();
  dctn_set_last_kernel("allreduce_direct");
  return DCTN_OK;
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
