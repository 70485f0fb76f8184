// Direct (peer-mapped) collectives for the tensor-parallel group on the 8-GPU xGMI mesh:
// all-reduce, all-reduce fused with residual-add + RMSNorm, and all-gather, for the
// messages of a decode step (a few KiB to a few MiB, ~160 per step on a 70B model).
//
// xGMI is point-to-point (every GPU has a link to every other one), so each
// rank PULLS what it needs straight from its peers' memory and all 7 links of a
// GPU carry traffic at once; nothing walks a ring and nobody writes into
// another GPU's memory (a GPU's own L2 never has to learn about foreign stores).
//
// Memory.  Every rank owns one uncached (fine-grained) region
//     [stage 0 | stage 1 | shard 0 | shard 1]        (4 x capacity bytes)
// and one uncached signal block, both exported with hipIpcGetMemHandle and
// mapped by all peers.  "stage" holds this rank's input (its partial sums): the
// row-parallel GEMM writes its output there directly, so there is no staging
// copy.  "shard" holds data this rank produces INSIDE a kernel for its peers
// (its slice of the sum in the two-shot forms, its piece of an all-gather).
// Consecutive calls alternate between the two copies of both areas ("parity"),
// which removes every write-after-read hazard without an end-of-kernel barrier:
// a buffer written for call k was last read by peers during call k-2, and a
// peer that has arrived at call k-1's barrier has finished call k-2.
//
// Forms.
//   one-shot  (small messages): barrier; every rank sums all peers' stage areas.
//   two-shot  (larger ones):    barrier; rank r sums slice r of all stage areas
//             into its shard area (+ its output); barrier; every rank pulls the
//             other slices from its peers' shard areas.  Traffic per rank is 2/W
//             of the one-shot form.
//   fused add + RMSNorm: the same two forms by rows.  One-shot keeps the residual
//             stream replicated; two-shot keeps it row-sharded (rank r owns the
//             residual rows of its slice, normalises only those, and the gather
//             moves normalised rows), which is the sequence-parallel layout.
//
// Ordering.  Barriers are per workgroup: workgroup b of every rank meets
// workgroup b of every other rank (identical grids everywhere), through flags
// in the peers' signal blocks written and polled with system-scope atomics.
// A barrier that publishes data written by the same kernel drains every wave's
// stores, then issues a system-scope release fence before the flag stores;
// every barrier ends with a system-scope acquire fence after the poll.  The
// barrier that opens a kernel publishes data of an EARLIER kernel on the stream
// (complete at the kernel boundary, in uncached memory); it can be given the
// release fence as well (`fence_start`).  Flag values are per-workgroup epochs
// kept in device memory, compared as "at least", so a captured hipGraph replays
// correctly and a peer running one barrier ahead is harmless.  Every poll is
// bounded by the wall clock: a rank whose peer never arrives sets
// `timeout_flag` and falls through instead of hanging the GPU.
#include "common.h"

namespace {

constexpr int CAR_MAX_RANKS = 8;
constexpr int CAR_MAX_BLOCKS = 128;
constexpr int CAR_FLAT_THREADS = 512;
constexpr int CAR_ROW_ITERS = 4;                                   // 16-byte vectors per thread and row
constexpr unsigned long long CAR_TIMEOUT_TICKS = 1000ULL * 1000 * 1000;   // 10 s of the 100 MHz wall clock

struct Signal {
  alignas(128) uint32_t flag[CAR_MAX_BLOCKS][CAR_MAX_RANKS];   // [workgroup][writer rank]
  alignas(128) uint32_t epoch[CAR_MAX_BLOCKS];                 // barriers this rank's workgroup b has passed
  alignas(128) uint32_t timeout_flag;
};

struct Peers {
  Signal* sig[CAR_MAX_RANKS];
  char* data[CAR_MAX_RANKS];
  uint32_t* host_timeout;     // this rank's word in pinned host memory: the host polls it without touching a stream
  unsigned long long timeout_ticks;   // bound of every spin, in ticks of the 100 MHz wall clock
};

struct Context {
  int rank, world;
  size_t cap;                 // bytes of each of the four areas
  Signal* self_sig;
  char* self_data;
  Peers peers;
  void* opened[2 * CAR_MAX_RANKS];
  int n_opened;
  int max_blocks;             // workgroups per kernel (<= CAR_MAX_BLOCKS); all ranks use the same value
  uint32_t* host_flag;        // pinned, device-mapped: set by a kernel whose spin ran out
};

// Workgroup b of all ranks meet.  RELEASE: this kernel wrote data the peers read after the barrier.
template <bool RELEASE>
__device__ __forceinline__ void cross_barrier(const Peers& peers, Signal* self, int rank, int world,
                                              uint32_t epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every wave: its stores have been performed
  __syncthreads();
  if (threadIdx.x < 64) {                                  // first wave only (wave-uniform branch)
    if (RELEASE) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope: write back before the flags
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if ((int)threadIdx.x < world) {
      const int p = threadIdx.x;
      __hip_atomic_store(&peers.sig[p]->flag[blockIdx.x][rank], epoch, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
      uint32_t* mine = &self->flag[blockIdx.x][p];
      const unsigned long long t0 = wall_clock64();
      while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
        if (wall_clock64() - t0 > peers.timeout_ticks) {
          __hip_atomic_store(&self->timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (peers.host_timeout != nullptr)
            __hip_atomic_store(peers.host_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // system scope: nothing read below is older than the flags
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

template <typename T>
__device__ __forceinline__ void add8(float (&acc)[8], typename Vec8<T>::type v) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
}

// Sum of packet i (16 bytes) of every rank's area at byte offset `off`, in rank order
// (bitwise identical on all ranks).  All loads are issued before the first add.
template <typename T>
__device__ __forceinline__ void sum_peers(const Peers& peers, int world, long off, long i, float (&acc)[8]) {
  using V = typename Vec8<T>::type;
  V v[CAR_MAX_RANKS];
#pragma unroll
  for (int r = 0; r < CAR_MAX_RANKS; ++r)
    if (r < world) v[r] = reinterpret_cast<const V*>(peers.data[r] + off)[i];
#pragma unroll
  for (int r = 0; r < CAR_MAX_RANKS; ++r)
    if (r < world) add8<T>(acc, v[r]);
}

// ---------------------------------------------------------------- flat all-reduce
template <typename T, bool FENCE_START>
__global__ __launch_bounds__(CAR_FLAT_THREADS) void oneshot_flat_kernel(Peers peers, Signal* self, int rank,
                                                                        int world, long stage_off, T* out,
                                                                        long packets) {
  using V = typename Vec8<T>::type;
  const uint32_t epoch = self->epoch[blockIdx.x] + 1;
  cross_barrier<FENCE_START>(peers, self, rank, world, epoch);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < packets; i += (long)gridDim.x * blockDim.x) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    sum_peers<T>(peers, world, stage_off, i, acc);
    V o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)acc[j];
    reinterpret_cast<V*>(out)[i] = o;
  }
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = epoch;
}

// Slice r = packets [r*per, (r+1)*per).  Workgroup b of rank r produces, and workgroup b of
// every other rank later pulls, the packets {r*per + b*blockDim + t + j*gridDim*blockDim}.
template <typename T, bool FENCE_START>
__global__ __launch_bounds__(CAR_FLAT_THREADS) void twoshot_flat_kernel(Peers peers, Signal* self, int rank,
                                                                        int world, long stage_off, long shard_off,
                                                                        T* out, long packets, long per) {
  using V = typename Vec8<T>::type;
  uint32_t epoch = self->epoch[blockIdx.x] + 1;
  cross_barrier<FENCE_START>(peers, self, rank, world, epoch);
  const long lo = min((long)rank * per, packets), cnt = min(lo + per, packets) - lo;
  V* my_shard = reinterpret_cast<V*>(peers.data[rank] + shard_off);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < cnt; i += (long)gridDim.x * blockDim.x) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    sum_peers<T>(peers, world, stage_off, lo + i, acc);
    V o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)acc[j];
    my_shard[i] = o;
    reinterpret_cast<V*>(out)[lo + i] = o;
  }
  ++epoch;
  cross_barrier<true>(peers, self, rank, world, epoch);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
    V v[CAR_MAX_RANKS];
#pragma unroll
    for (int p = 0; p < CAR_MAX_RANKS; ++p)
      if (p < world && p != rank && (long)p * per + i < packets)
        v[p] = reinterpret_cast<const V*>(peers.data[p] + shard_off)[i];
#pragma unroll
    for (int p = 0; p < CAR_MAX_RANKS; ++p)
      if (p < world && p != rank && (long)p * per + i < packets) reinterpret_cast<V*>(out)[(long)p * per + i] = v[p];
  }
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = epoch;
}

// ------------------------------------------------- all-reduce + residual add + RMSNorm
// v = sum over ranks of stage[row] + residual[row];  residual[row] = v;  out[row] = v * rsqrt(mean(v^2)+eps) * w
// ITERS: 16-byte pieces of the row per thread (1 in the wide form, where a workgroup has a thread per piece: the
// four-piece form unrolled with every peer's loads hoisted to the top does not fit that form's 128 registers)
template <typename T, int ITERS = CAR_ROW_ITERS>
__device__ __forceinline__ void reduce_norm_row(const Peers& peers, int world, long stage_off, long row, int H,
                                                T* __restrict__ res_row, const T* __restrict__ w, float eps,
                                                T* __restrict__ out_row, T* __restrict__ shard_row, float* red) {
  using V = typename Vec8<T>::type;
  V vals[ITERS];
  float ss = 0.f;
  const int nvec = H >> 3;
  // (the first piece of the norm weight travels with the row: behind the block-wide sum it would be a second
  // trip to HBM in a kernel that is all latency at decode sizes; the wide one-pass form has only this piece)
  V g0;
  if ((int)threadIdx.x < nvec) g0 = ld8<T>(w + 8 * threadIdx.x);
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      float acc[8];
      V r = ld8<T>(res_row + 8 * i);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = (float)r[j];
      sum_peers<T>(peers, world, stage_off, row * (long)nvec + i, acc);
      V v;
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = (T)acc[j]; const float f = (float)v[j]; ss += f * f; }
      st8<T>(res_row + 8 * i, v);
      vals[it] = v;
    }
  }
  ss = block_sum(ss, red);
  const float inv = rsqrtf(ss / (float)H + eps);
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      V g = it == 0 ? g0 : ld8<T>(w + 8 * i), o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (T)((float)vals[it][j] * inv * (float)g[j]);
      st8<T>(out_row + 8 * i, o);
      if (shard_row != nullptr) st8<T>(shard_row + 8 * i, o);
    }
  }
}

template <typename T, bool FENCE_START, int ITERS>
__global__ __launch_bounds__(ITERS == 1 ? 1024 : 512) void oneshot_norm_kernel(Peers peers, Signal* self, int rank, int world, long stage_off, T* out,
                                    T* residual, const T* w, float eps, int rows, int H) {
  __shared__ float red[17];
  const uint32_t epoch = self->epoch[blockIdx.x] + 1;
  cross_barrier<FENCE_START>(peers, self, rank, world, epoch);
  for (long row = blockIdx.x; row < rows; row += gridDim.x)
    reduce_norm_row<T, ITERS>(peers, world, stage_off, row, H, residual + row * (long)H, w, eps,
                              out + row * (long)H, nullptr, red);
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = epoch;
}

// Rank r owns rows [r*per, (r+1)*per): `residual` holds only those ([per, H], row-sharded).
template <typename T, bool FENCE_START, int ITERS>
__global__ __launch_bounds__(ITERS == 1 ? 1024 : 512) void twoshot_norm_kernel(Peers peers, Signal* self, int rank, int world, long stage_off, long shard_off,
                                    T* out, T* residual, const T* w, float eps, int rows, int H, int per) {
  __shared__ float red[17];
  using V = typename Vec8<T>::type;
  uint32_t epoch = self->epoch[blockIdx.x] + 1;
  cross_barrier<FENCE_START>(peers, self, rank, world, epoch);
  const int lo = min(rank * per, rows), cnt = min(lo + per, rows) - lo;
  T* my_shard = reinterpret_cast<T*>(peers.data[rank] + shard_off);
  for (int lr = blockIdx.x; lr < cnt; lr += gridDim.x)
    reduce_norm_row<T, ITERS>(peers, world, stage_off, lo + lr, H, residual + lr * (long)H, w, eps,
                              out + (lo + lr) * (long)H, my_shard + lr * (long)H, red);
  ++epoch;
  cross_barrier<true>(peers, self, rank, world, epoch);
  const int nvec = H >> 3;
  for (int lr = blockIdx.x; lr < per; lr += gridDim.x) {
    for (int i = threadIdx.x; i < nvec; i += blockDim.x) {
      V v[CAR_MAX_RANKS];
#pragma unroll
      for (int p = 0; p < CAR_MAX_RANKS; ++p)
        if (p < world && p != rank && p * per + lr < rows)
          v[p] = reinterpret_cast<const V*>(peers.data[p] + shard_off)[lr * (long)nvec + i];
#pragma unroll
      for (int p = 0; p < CAR_MAX_RANKS; ++p)
        if (p < world && p != rank && p * per + lr < rows)
          reinterpret_cast<V*>(out)[(p * per + lr) * (long)nvec + i] = v[p];
    }
  }
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = epoch;
}

// ------------------------------------------------------------------- all-gather
// out = [rank 0's piece | rank 1's piece | ...], `packets` 16-byte packets per rank
template <typename T>
__global__ __launch_bounds__(CAR_FLAT_THREADS) void all_gather_kernel(Peers peers, Signal* self, int rank, int world,
                                                                      long shard_off, const T* in, T* out,
                                                                      long packets) {
  using V = typename Vec8<T>::type;
  const uint32_t epoch = self->epoch[blockIdx.x] + 1;
  V* my_shard = reinterpret_cast<V*>(peers.data[rank] + shard_off);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < packets; i += (long)gridDim.x * blockDim.x) {
    const V v = reinterpret_cast<const V*>(in)[i];
    my_shard[i] = v;
    reinterpret_cast<V*>(out)[rank * packets + i] = v;
  }
  cross_barrier<true>(peers, self, rank, world, epoch);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < packets; i += (long)gridDim.x * blockDim.x) {
    V v[CAR_MAX_RANKS];
#pragma unroll
    for (int p = 0; p < CAR_MAX_RANKS; ++p)
      if (p < world && p != rank) v[p] = reinterpret_cast<const V*>(peers.data[p] + shard_off)[i];
#pragma unroll
    for (int p = 0; p < CAR_MAX_RANKS; ++p)
      if (p < world && p != rank) reinterpret_cast<V*>(out)[p * packets + i] = v[p];
  }
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = epoch;
}

// One meeting of workgroup 0 of every rank: a call that only advances the buffer parity.
__global__ void barrier_kernel(Peers peers, Signal* self, int rank, int world) {
  const uint32_t epoch = self->epoch[0] + 1;
  cross_barrier<false>(peers, self, rank, world, epoch);
  if (threadIdx.x == 0) self->epoch[0] = epoch;
}

inline int row_threads(int H) {
  int t = ((H >> 3) + CAR_ROW_ITERS - 1) / CAR_ROW_ITERS;
  t = (t + 63) / 64 * 64;
  if (t < 64) t = 64;
  if (t > 512) t = 512;       // (the several-pieces-per-thread kernels are built for at most 512 threads)
  return t;
}

inline int flat_blocks(long packets, int max_blocks) {
  long b = (packets + CAR_FLAT_THREADS - 1) / CAR_FLAT_THREADS;
  if (b > max_blocks) b = max_blocks;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

// Bound of every in-kernel spin (default 10 s).  A rank whose peer does not arrive within it sets the
// timeout word and carries on with whatever it read.
VDAMD_API void vdamd_car_set_timeout_ms(void* ctx, long ms) {
  Context* c = (Context*)ctx;
  if (ms < 1) ms = 1;
  c->peers.timeout_ticks = (unsigned long long)ms * 100000ULL;
}

VDAMD_API int vdamd_car_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

// Allocates the data region (4 areas of `cap` bytes) and the signal block.
VDAMD_API void* vdamd_car_create(int rank, int world, size_t cap, void* data_handle_out, void* sig_handle_out) {
  if (world < 2 || world > CAR_MAX_RANKS || rank < 0 || rank >= world || cap == 0 || cap % 256 != 0) return nullptr;
  Context* c = new Context();
  c->rank = rank; c->world = world; c->cap = cap; c->n_opened = 0; c->max_blocks = CAR_MAX_BLOCKS;
  if (hipExtMallocWithFlags((void**)&c->self_data, 4 * cap, hipDeviceMallocUncached) != hipSuccess) {
    delete c; return nullptr;
  }
  if (hipExtMallocWithFlags((void**)&c->self_sig, sizeof(Signal), hipDeviceMallocUncached) != hipSuccess) {
    (void)hipFree(c->self_data); delete c; return nullptr;
  }
  (void)hipMemset(c->self_sig, 0, sizeof(Signal));
  (void)hipMemset(c->self_data, 0, 4 * cap);
  (void)hipDeviceSynchronize();
  if (hipIpcGetMemHandle((hipIpcMemHandle_t*)data_handle_out, c->self_data) != hipSuccess ||
      hipIpcGetMemHandle((hipIpcMemHandle_t*)sig_handle_out, c->self_sig) != hipSuccess) {
    (void)hipFree(c->self_data); (void)hipFree(c->self_sig); delete c; return nullptr;
  }
  for (int r = 0; r < CAR_MAX_RANKS; ++r) { c->peers.sig[r] = nullptr; c->peers.data[r] = nullptr; }
  c->host_flag = nullptr;
  c->peers.host_timeout = nullptr;
  c->peers.timeout_ticks = CAR_TIMEOUT_TICKS;
  if (hipHostMalloc((void**)&c->host_flag, 64, hipHostMallocMapped) == hipSuccess) {
    *c->host_flag = 0;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, c->host_flag, 0) == hipSuccess) c->peers.host_timeout = (uint32_t*)dp;
  }
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

// Address of this rank's stage area `parity` (0/1): where the producer of the next input writes.
VDAMD_API void* vdamd_car_stage_ptr(void* ctx, int parity) {
  Context* c = (Context*)ctx;
  return c->self_data + (size_t)(parity & 1) * c->cap;
}

// Workgroups per kernel.  Every rank of the group must set the same value.  The default
// (one GPU per rank) is always resident; several ranks sharing one device lower it so
// that the grids of all of them are resident together, which the barriers need.
VDAMD_API void vdamd_car_set_max_blocks(void* ctx, int n) {
  Context* c = (Context*)ctx;
  c->max_blocks = n < 1 ? 1 : (n > CAR_MAX_BLOCKS ? CAR_MAX_BLOCKS : n);
}

// Whether a bounded spin of any kernel of this rank has run out since creation.  Reads a word in
// pinned host memory: no stream is touched, so the serving loop can ask after every step.
VDAMD_API int vdamd_car_timed_out(void* ctx) {
  Context* c = (Context*)ctx;
  if (c->host_flag != nullptr && c->peers.host_timeout != nullptr)
    return (int)__atomic_load_n(c->host_flag, __ATOMIC_RELAXED);
  uint32_t v = 0;
  (void)hipMemcpy(&v, &c->self_sig->timeout_flag, sizeof(v), hipMemcpyDeviceToHost);
  return (int)v;
}

namespace {
// `in` == nullptr or == the stage area itself: the input is already in place
inline int stage_input(Context* c, int parity, const void* in, size_t bytes, hipStream_t s) {
  char* stage = c->self_data + (size_t)(parity & 1) * c->cap;
  if (in == nullptr || in == (const void*)stage) return 0;
  return (int)hipMemcpyAsync(stage, in, bytes, hipMemcpyDeviceToDevice, s);
}
}  // namespace

// Sum of `bytes` (multiple of 16) over the group into `out` (ordinary device memory).
VDAMD_API int vdamd_car_all_reduce(void* ctx, int parity, const void* in, void* out, size_t bytes, int dtype,
                                   int two_shot, int fence_start, hipStream_t s) {
  Context* c = (Context*)ctx;
  if (bytes == 0) return 0;
  if (bytes % 16 != 0 || bytes > c->cap) return -1;
  if (dtype != DT_BF16 && dtype != DT_F16) return -2;
  int rc = stage_input(c, parity, in, bytes, s);
  if (rc != 0) return rc;
  const long packets = (long)(bytes / 16);
  const long stage_off = (long)((size_t)(parity & 1) * c->cap), shard_off = (long)((size_t)(2 + (parity & 1)) * c->cap);
  const long per = (packets + c->world - 1) / c->world;
  const int blocks = two_shot ? flat_blocks(per, c->max_blocks) : flat_blocks(packets, c->max_blocks);
#define RUN(TY, FS)                                                                                              \
  if (two_shot)                                                                                                  \
    hipLaunchKernelGGL((twoshot_flat_kernel<TY, FS>), dim3(blocks), dim3(CAR_FLAT_THREADS), 0, s, c->peers,      \
                       c->self_sig, c->rank, c->world, stage_off, shard_off, (TY*)out, packets, per);            \
  else                                                                                                           \
    hipLaunchKernelGGL((oneshot_flat_kernel<TY, FS>), dim3(blocks), dim3(CAR_FLAT_THREADS), 0, s, c->peers,      \
                       c->self_sig, c->rank, c->world, stage_off, (TY*)out, packets)
  if (dtype == DT_BF16) { if (fence_start) { RUN(bf16_t, true); } else { RUN(bf16_t, false); } }
  else { if (fence_start) { RUN(f16_t, true); } else { RUN(f16_t, false); } }
#undef RUN
  HIP_LAUNCH_CHECK();
  return 0;
}

// out[rows, H] = RMSNorm(sum over ranks of x + residual) * w, residual updated in place.
// one-shot: residual is [rows, H] on every rank.  two-shot: residual is this rank's
// [ceil(rows/world), H] row shard (rows [rank*per, ...)); `out` is complete on every rank.
VDAMD_API int vdamd_car_fused_norm(void* ctx, int parity, const void* in, void* out, void* residual, const void* w,
                                   float eps, int rows, int H, int dtype, int two_shot, int fence_start,
                                   hipStream_t s) {
  Context* c = (Context*)ctx;
  if (rows == 0) return 0;
  const size_t bytes = (size_t)rows * H * 2;
  if (H % 8 != 0 || H > 8 * 512 * CAR_ROW_ITERS || bytes > c->cap) return -1;
  if (dtype != DT_BF16 && dtype != DT_F16) return -2;
  int rc = stage_input(c, parity, in, bytes, s);
  if (rc != 0) return rc;
  const long stage_off = (long)((size_t)(parity & 1) * c->cap), shard_off = (long)((size_t)(2 + (parity & 1)) * c->cap);
  const int per = (rows + c->world - 1) / c->world;
  int blocks = two_shot ? per : rows;
  if (blocks > c->max_blocks) blocks = c->max_blocks;
  int threads = row_threads(H);
  if (blocks <= 64) {
    // few rows per rank (a decode batch split over the group): one pass of a wider workgroup, so that every
    // 16-byte piece of the row is requested from every peer at once instead of a quarter of them at a time
    threads = ((H >> 3) + 63) / 64 * 64;
    if (threads > 1024) threads = 1024;
  }
  const bool one_piece = threads * 8 >= H;      // a thread per 16-byte piece
  if (!one_piece && threads > 512) threads = 512;
#define RUN(TY, FS, IT)                                                                                             \
  if (two_shot)                                                                                                     \
    hipLaunchKernelGGL((twoshot_norm_kernel<TY, FS, IT>), dim3(blocks), dim3(threads), 0, s, c->peers, c->self_sig, \
                       c->rank, c->world, stage_off, shard_off, (TY*)out, (TY*)residual, (const TY*)w, eps,         \
                       rows, H, per);                                                                               \
  else                                                                                                              \
    hipLaunchKernelGGL((oneshot_norm_kernel<TY, FS, IT>), dim3(blocks), dim3(threads), 0, s, c->peers, c->self_sig, \
                       c->rank, c->world, stage_off, (TY*)out, (TY*)residual, (const TY*)w, eps, rows, H)
#define RUN_FS(TY, IT) if (fence_start) { RUN(TY, true, IT); } else { RUN(TY, false, IT); }
  if (dtype == DT_BF16) { if (one_piece) { RUN_FS(bf16_t, 1) } else { RUN_FS(bf16_t, CAR_ROW_ITERS) } }
  else { if (one_piece) { RUN_FS(f16_t, 1) } else { RUN_FS(f16_t, CAR_ROW_ITERS) } }
#undef RUN_FS
#undef RUN
  HIP_LAUNCH_CHECK();
  return 0;
}

// out = concatenation over ranks of `bytes` bytes each (multiple of 16, any element type).
VDAMD_API int vdamd_car_all_gather(void* ctx, int parity, const void* in, void* out, size_t bytes, hipStream_t s) {
  Context* c = (Context*)ctx;
  if (bytes == 0) return 0;
  if (bytes % 16 != 0 || bytes > c->cap) return -1;
  const long packets = (long)(bytes / 16);
  const long shard_off = (long)((size_t)(2 + (parity & 1)) * c->cap);
  hipLaunchKernelGGL(all_gather_kernel<bf16_t>, dim3(flat_blocks(packets, c->max_blocks)), dim3(CAR_FLAT_THREADS), 0, s,
                     c->peers, c->self_sig, c->rank, c->world, shard_off, (const bf16_t*)in, (bf16_t*)out, packets);
  HIP_LAUNCH_CHECK();
  return 0;
}

VDAMD_API int vdamd_car_barrier(void* ctx, hipStream_t s) {
  Context* c = (Context*)ctx;
  hipLaunchKernelGGL(barrier_kernel, dim3(1), dim3(64), 0, s, c->peers, c->self_sig, c->rank, c->world);
  HIP_LAUNCH_CHECK();
  return 0;
}

VDAMD_API void vdamd_car_destroy(void* ctx) {
  Context* c = (Context*)ctx;
  if (!c) return;
  for (int i = 0; i < c->n_opened; ++i) (void)hipIpcCloseMemHandle(c->opened[i]);
  (void)hipFree(c->self_data);
  (void)hipFree(c->self_sig);
  if (c->host_flag != nullptr) (void)hipHostFree(c->host_flag);
  delete c;
}
