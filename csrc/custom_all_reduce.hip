// Direct (peer-mapped) all-reduce for small/medium messages on the 8-GPU xGMI mesh.
//
// Every rank owns a staging buffer and a signal block, both exported with
// hipIpcGetMemHandle and mapped by all peers.  An all-reduce is
//   1. hipMemcpyAsync(input -> own staging)            [kernel boundary => visible to peers]
//   2a. one-shot  (<= ONE_SHOT_BYTES): every rank reads all peers' staging, sums
//       in rank order (bitwise identical on all ranks), writes its output;
//   2b. two-shot: kernel A reduce-scatters (rank r sums slice r from all peers
//       into its own result area), kernel B all-gathers the result areas.
// xGMI is point-to-point, so both shapes keep all 7 links of a GPU busy at
// once instead of walking a ring.
//
// Cross-GPU ordering uses per-block flags with system-scope atomics; flags and
// staging both live in uncached (fine-grained) memory, so data a peer wrote
// before raising its flag is what a reader sees after observing the flag.  Epochs are kept in device memory so a captured hipGraph replays
// correctly.  Every spin is bounded: a rank whose peer never arrives sets
// `timeout_flag` and falls through rather than hanging the GPU.
#include "common.h"

namespace {

constexpr int CAR_MAX_RANKS = 8;
constexpr int CAR_MAX_BLOCKS = 64;
constexpr int CAR_THREADS = 512;
constexpr long CAR_SPIN_LIMIT = 4L * 1000 * 1000;     // a few seconds of uncached polling

struct Signal {
  alignas(128) uint32_t epoch[CAR_MAX_BLOCKS];
  alignas(128) uint32_t start[CAR_MAX_BLOCKS][CAR_MAX_RANKS];
  alignas(128) uint32_t end[CAR_MAX_BLOCKS][CAR_MAX_RANKS];
  alignas(128) uint32_t timeout_flag;
};

struct Peers {
  Signal* sig[CAR_MAX_RANKS];
  char* data[CAR_MAX_RANKS];
};

struct Context {
  int rank, world;
  size_t data_bytes;          // staging area; a result area of the same size follows it
  Signal* self_sig;
  char* self_data;
  Peers peers;
  void* opened[2 * CAR_MAX_RANKS];
  int n_opened;
};

__device__ __forceinline__ void st_sys(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t ld_sys(uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// All ranks' block `b` meet.  `which` selects the start or end flag set.
template <bool END>
__device__ __forceinline__ void cross_barrier(const Peers& peers, Signal* self, int rank, int world) {
  const int b = blockIdx.x;
  if (END) {
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_RELEASE);   // order this block's global writes before the flag
  }
  if (threadIdx.x < world) {
    const uint32_t epoch = self->epoch[b] + 1;
    uint32_t* remote = END ? &peers.sig[threadIdx.x]->end[b][rank] : &peers.sig[threadIdx.x]->start[b][rank];
    uint32_t* local = END ? &self->end[b][threadIdx.x] : &self->start[b][threadIdx.x];
    st_sys(remote, epoch);
    long spins = 0;
    while (ld_sys(local) != epoch) {
      if (++spins > CAR_SPIN_LIMIT) { st_sys(&self->timeout_flag, 1u); break; }
    }
  }
  __syncthreads();
  if (END && threadIdx.x == 0) self->epoch[b] += 1;
}

template <typename T>
__device__ __forceinline__ void add8(float (&acc)[8], typename Vec8<T>::type v) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
}

// sum packets [begin, end) (16-byte units) of every peer's staging into dst
template <typename T>
__device__ __forceinline__ void reduce_range(const Peers& peers, int world, T* dst, long begin, long end) {
  using V = typename Vec8<T>::type;
  for (long i = begin + blockIdx.x * (long)blockDim.x + threadIdx.x; i < end; i += (long)gridDim.x * blockDim.x) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < world; ++r) add8<T>(acc, reinterpret_cast<const V*>(peers.data[r])[i]);
    V o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)acc[j];
    reinterpret_cast<V*>(dst)[i] = o;
  }
}

template <typename T>
__global__ __launch_bounds__(CAR_THREADS) void one_shot_kernel(Peers peers, Signal* self, int rank, int world,
                                                               T* out, long packets) {
  cross_barrier<false>(peers, self, rank, world);
  reduce_range<T>(peers, world, out, 0, packets);
  cross_barrier<true>(peers, self, rank, world);     // nobody may restage before all have read
}

// rank r reduces slice r into its own result area (staging + data_bytes)
template <typename T>
__global__ __launch_bounds__(CAR_THREADS) void reduce_scatter_kernel(Peers peers, Signal* self, int rank, int world,
                                                                     long packets, long result_off_packets) {
  cross_barrier<false>(peers, self, rank, world);
  const long per = (packets + world - 1) / world;
  const long b = min((long)rank * per, packets), e = min(b + per, packets);
  using V = typename Vec8<T>::type;
  T* dst = reinterpret_cast<T*>(reinterpret_cast<V*>(peers.data[rank]) + result_off_packets);
  reduce_range<T>(peers, world, dst, b, e);
  cross_barrier<true>(peers, self, rank, world);
}

template <typename T>
__global__ __launch_bounds__(CAR_THREADS) void all_gather_kernel(Peers peers, Signal* self, int rank, int world,
                                                                 T* out, long packets, long result_off_packets) {
  using V = typename Vec8<T>::type;
  cross_barrier<false>(peers, self, rank, world);
  const long per = (packets + world - 1) / world;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < packets; i += (long)gridDim.x * blockDim.x) {
    const int owner = (int)(i / per);
    reinterpret_cast<V*>(out)[i] = (reinterpret_cast<const V*>(peers.data[owner]) + result_off_packets)[i];
  }
  cross_barrier<true>(peers, self, rank, world);
}

}  // namespace

VDAMD_API int vdamd_car_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

// Allocates staging (2 * data_bytes: input area + result area) and the signal block.
VDAMD_API void* vdamd_car_create(int rank, int world, size_t data_bytes, void* data_handle_out,
                                 void* sig_handle_out) {
  if (world < 2 || world > CAR_MAX_RANKS || rank < 0 || rank >= world) return nullptr;
  Context* c = new Context();
  c->rank = rank; c->world = world; c->data_bytes = data_bytes; c->n_opened = 0;
  // Uncached (fine-grained) like the flags: a peer's reduce-scatter result is read by this GPU's
  // all-gather kernel inside one kernel pair, with only the in-kernel barrier between them, so the
  // reader must never be served a line its L2 kept from the previous call on the same addresses.
  if (hipExtMallocWithFlags((void**)&c->self_data, 2 * data_bytes, hipDeviceMallocUncached) != hipSuccess) {
    delete c; return nullptr;
  }
  if (hipExtMallocWithFlags((void**)&c->self_sig, sizeof(Signal), hipDeviceMallocUncached) != hipSuccess) {
    (void)hipFree(c->self_data); delete c; return nullptr;
  }
  (void)hipMemset(c->self_sig, 0, sizeof(Signal));
  (void)hipMemset(c->self_data, 0, 2 * data_bytes);
  (void)hipDeviceSynchronize();
  if (hipIpcGetMemHandle((hipIpcMemHandle_t*)data_handle_out, c->self_data) != hipSuccess ||
      hipIpcGetMemHandle((hipIpcMemHandle_t*)sig_handle_out, c->self_sig) != hipSuccess) {
    (void)hipFree(c->self_data); (void)hipFree(c->self_sig); delete c; return nullptr;
  }
  for (int r = 0; r < CAR_MAX_RANKS; ++r) { c->peers.sig[r] = nullptr; c->peers.data[r] = nullptr; }
  c->peers.sig[rank] = c->self_sig;
  c->peers.data[rank] = c->self_data;
  return c;
}

// handles: world entries each, in rank order (own entry ignored)
VDAMD_API int vdamd_car_open_peers(void* ctx, const void* data_handles, const void* sig_handles) {
  Context* c = (Context*)ctx;
  const hipIpcMemHandle_t* dh = (const hipIpcMemHandle_t*)data_handles;
  const hipIpcMemHandle_t* sh = (const hipIpcMemHandle_t*)sig_handles;
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, dh[r], hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return (int)e;
    c->peers.data[r] = (char*)p; c->opened[c->n_opened++] = p;
    e = hipIpcOpenMemHandle(&p, sh[r], hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return (int)e;
    c->peers.sig[r] = (Signal*)p; c->opened[c->n_opened++] = p;
  }
  return 0;
}

VDAMD_API int vdamd_car_timed_out(void* ctx) {
  Context* c = (Context*)ctx;
  uint32_t v = 0;
  (void)hipMemcpy(&v, &c->self_sig->timeout_flag, sizeof(v), hipMemcpyDeviceToHost);
  return (int)v;
}

// mode: 0 auto, 1 one-shot, 2 two-shot.  `bytes` must be a multiple of 16.
VDAMD_API int vdamd_car_all_reduce(void* ctx, const void* in, void* out, size_t bytes, int dtype, int mode,
                                   size_t one_shot_bytes, hipStream_t s) {
  Context* c = (Context*)ctx;
  if (bytes == 0) return 0;
  if (bytes % 16 != 0 || bytes > c->data_bytes) return -1;
  if (dtype != DT_BF16 && dtype != DT_F16) return -2;
  hipError_t e = hipMemcpyAsync(c->self_data, in, bytes, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return (int)e;
  const long packets = (long)(bytes / 16);
  int blocks = (int)((packets + CAR_THREADS - 1) / CAR_THREADS);
  if (blocks > CAR_MAX_BLOCKS) blocks = CAR_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  const bool one = mode == 1 || (mode == 0 && bytes <= one_shot_bytes);
  const long res_off = (long)(c->data_bytes / 16);
#define RUN(TY)                                                                                              \
  if (one) {                                                                                                 \
    hipLaunchKernelGGL(one_shot_kernel<TY>, dim3(blocks), dim3(CAR_THREADS), 0, s, c->peers, c->self_sig,     \
                       c->rank, c->world, (TY*)out, packets);                                                \
  } else {                                                                                                   \
    hipLaunchKernelGGL(reduce_scatter_kernel<TY>, dim3(blocks), dim3(CAR_THREADS), 0, s, c->peers,            \
                       c->self_sig, c->rank, c->world, packets, res_off);                                    \
    hipLaunchKernelGGL(all_gather_kernel<TY>, dim3(blocks), dim3(CAR_THREADS), 0, s, c->peers, c->self_sig,   \
                       c->rank, c->world, (TY*)out, packets, res_off);                                       \
  }
  if (dtype == DT_BF16) { RUN(bf16_t) } else { RUN(f16_t) }
#undef RUN
  HIP_LAUNCH_CHECK();
  return 0;
}

VDAMD_API void vdamd_car_destroy(void* ctx) {
  Context* c = (Context*)ctx;
  if (!c) return;
  for (int i = 0; i < c->n_opened; ++i) (void)hipIpcCloseMemHandle(c->opened[i]);
  (void)hipFree(c->self_data);
  (void)hipFree(c->self_sig);
  delete c;
}
