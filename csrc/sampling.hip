// Token sampling: greedy argmax and temperature / top-k / top-p / min-p
// inverse-CDF sampling, one 1024-thread workgroup per logits row.
//
// The kept set under top-k AND top-p AND min-p is always a prefix of the
// descending-sorted distribution, so it is described by one probability
// threshold tau (plus how many of the entries tied at tau survive).  tau is
// found without sorting by a bitwise binary search over the float's integer
// representation (positive floats order like their bit patterns); each probe
// is one reduction pass over the row (L2-resident: a 128k-entry fp32 row is
// 512 KB).  The sampled entry is then located the same way from u * kept_mass.
// Semantics match ops/torch_ref.py::sample (descending order, ties by index).
#include "common.h"

namespace {

constexpr int ST = 1024;     // threads per row

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }

struct ArgMax { float v; int i; };

__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

template <typename T>
__global__ __launch_bounds__(ST) void argmax_kernel(int* __restrict__ out, const T* __restrict__ logits, int V,
                                                    long stride) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const T* row = logits + blockIdx.x * stride;
  ArgMax best{-INFINITY, 0x7fffffff};
  for (int i = threadIdx.x; i < V; i += ST) {
    const float v = to_f(row[i]);
    if (v > best.v || (v == best.v && i < best.i)) best = ArgMax{v, i};     // NaN never wins
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax other{__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)};
    best = better(best, other);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sv[wid] = best.v; si[wid] = best.i; }
  __syncthreads();
  if (wid == 0) {
    ArgMax b2 = lane < 16 ? ArgMax{sv[lane], si[lane]} : ArgMax{-INFINITY, 0x7fffffff};
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      ArgMax other{__shfl_xor(b2.v, o, 64), __shfl_xor(b2.i, o, 64)};
      b2 = better(b2, other);
    }
    if (lane == 0) out[blockIdx.x] = b2.i == 0x7fffffff ? 0 : b2.i;
  }
}

// block-wide (sum, count) reduction; result broadcast to all threads
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  a = wave_sum(a); b = wave_sum(b);
  if (lane == 0) { red[wid] = a; red[16 + wid] = b; }
  __syncthreads();
  float ta = lane < 16 ? red[lane] : 0.f, tb = lane < 16 ? red[16 + lane] : 0.f;
  ta = wave_sum(ta); tb = wave_sum(tb);
  __syncthreads();
  a = ta; b = tb;
}

__device__ __forceinline__ float block_max(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = wave_max(v);
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = lane < 16 ? red[lane] : -INFINITY;
  t = wave_max(t);
  __syncthreads();
  return t;
}

template <typename T>
__global__ __launch_bounds__(ST) void sample_kernel(int* __restrict__ out, const T* __restrict__ logits, int V,
                                                    long stride, const float* __restrict__ temperature,
                                                    const int* __restrict__ top_k, const float* __restrict__ top_p,
                                                    const float* __restrict__ min_p,
                                                    const float* __restrict__ uniform) {
  __shared__ float red[32];
  __shared__ int s_idx;
  const int b = blockIdx.x;
  const T* row = logits + b * stride;
  const float temp = temperature[b];
  // ---- greedy row: plain argmax (lowest index on ties)
  if (temp == 0.f) {
    ArgMax best{-INFINITY, 0x7fffffff};
    for (int i = threadIdx.x; i < V; i += ST) {
      const float v = to_f(row[i]);
      if (v > best.v || (v == best.v && i < best.i)) best = ArgMax{v, i};
    }
    const float mx = block_max(best.v, red);
    int cand = (best.v == mx) ? best.i : 0x7fffffff;
    if (threadIdx.x == 0) s_idx = 0x7fffffff;
    __syncthreads();
    atomicMin(&s_idx, cand);
    __syncthreads();
    if (threadIdx.x == 0) out[b] = s_idx == 0x7fffffff ? 0 : s_idx;
    return;
  }
  const float inv_t = 1.f / temp;
  int k = top_k[b];
  if (k <= 0 || k > V) k = V;
  const float tp = top_p[b], mp = min_p[b], u = uniform[b];

  // ---- softmax statistics
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < V; i += ST) mx = fmaxf(mx, to_f(row[i]) * inv_t);
  mx = block_max(mx, red);
  float z = 0.f, dummy = 0.f;
  for (int i = threadIdx.x; i < V; i += ST) z += __expf(to_f(row[i]) * inv_t - mx);
  block_sum2(z, dummy, red);
  const float inv_z = 1.f / z;
  const float p_max = inv_z;                      // exp(0) / z
  const float minp_thr = mp * p_max;
  auto prob = [&](int i) { return __expf(to_f(row[i]) * inv_t - mx) * inv_z; };

  // ---- tau = smallest probability value whose first occurrence is kept:
  //      count(p > c) < k  AND  mass(p > c) < top_p  AND  c >= min_p * p_max
  // predicate is monotone in c; search the bit pattern from the top bit down
  // for the smallest c that satisfies it.
  unsigned lo = 0u, hi = __float_as_uint(p_max);   // p_max always satisfies it
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    const float c = __uint_as_float(mid);
    float cnt = 0.f, mass = 0.f;
    for (int i = threadIdx.x; i < V; i += ST) {
      const float p = prob(i);
      if (p > c) { cnt += 1.f; mass += p; }
    }
    block_sum2(cnt, mass, red);
    const bool ok = cnt < (float)k && mass < tp && c >= minp_thr;
    if (ok) hi = mid; else lo = mid + 1;
  }
  // the smallest *attained* value >= lo
  const float c_lo = __uint_as_float(lo);
  float tau = INFINITY;
  for (int i = threadIdx.x; i < V; i += ST) { const float p = prob(i); if (p >= c_lo) tau = fminf(tau, p); }
  tau = -block_max(-tau, red);
  // ---- kept mass: everything above tau plus the surviving ties at tau
  float cnt_gt = 0.f, mass_gt = 0.f, ties = 0.f;
  for (int i = threadIdx.x; i < V; i += ST) {
    const float p = prob(i);
    if (p > tau) { cnt_gt += 1.f; mass_gt += p; } else if (p == tau) ties += 1.f;
  }
  block_sum2(cnt_gt, mass_gt, red);
  block_sum2(ties, dummy, red);
  float keep_ties = fminf(ties, (float)k - cnt_gt);
  if (tau > 0.f) keep_ties = fminf(keep_ties, ceilf((tp - mass_gt) / tau));
  keep_ties = fmaxf(keep_ties, 1.f);
  const float total = mass_gt + keep_ties * tau;
  const float target = u * total;

  // ---- locate the sampled entry: value v* = largest c with mass(p >= c) >= target
  //      i.e. smallest c whose mass(p > c) < target
  lo = __float_as_uint(tau); hi = __float_as_uint(p_max);
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    const float c = __uint_as_float(mid);
    float mass = 0.f;
    for (int i = threadIdx.x; i < V; i += ST) { const float p = prob(i); if (p > c) mass += p; }
    block_sum2(mass, dummy, red);
    if (mass < target) hi = mid; else lo = mid + 1;
  }
  const float c2 = __uint_as_float(lo);
  float vstar = INFINITY;
  for (int i = threadIdx.x; i < V; i += ST) { const float p = prob(i); if (p >= c2) vstar = fminf(vstar, p); }
  vstar = -block_max(-vstar, red);
  float mass_gt2 = 0.f;
  for (int i = threadIdx.x; i < V; i += ST) { const float p = prob(i); if (p > vstar) mass_gt2 += p; }
  block_sum2(mass_gt2, dummy, red);
  // tie number t (index order): smallest t with mass_gt2 + (t+1) * v* >= target
  int t_sel = vstar > 0.f ? (int)ceilf((target - mass_gt2) / vstar) - 1 : 0;
  if (t_sel < 0) t_sel = 0;
  if (vstar == tau && t_sel > (int)keep_ties - 1) t_sel = (int)keep_ties - 1;
  // pick the t_sel-th index (ascending) among entries equal to v*: ties are
  // rare, so a serial scan by the threads in index order via repeated min is fine
  int chosen = -1;
  int after = -1;
  for (int round = 0; round <= t_sel; ++round) {
    int cand = 0x7fffffff;
    for (int i = threadIdx.x; i < V; i += ST)
      if (i > after && prob(i) == vstar) { cand = i; break; }
    if (threadIdx.x == 0) s_idx = 0x7fffffff;
    __syncthreads();
    atomicMin(&s_idx, cand);
    __syncthreads();
    const int found = s_idx;
    __syncthreads();
    if (found == 0x7fffffff) break;
    chosen = found;
    after = found;
  }
  if (threadIdx.x == 0) out[b] = chosen < 0 ? 0 : chosen;
}

}  // namespace

VDAMD_API int vdamd_argmax(int* out, const void* logits, int B, int V, long stride, int dtype, hipStream_t s) {
  if (B == 0) return 0;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(argmax_kernel<bf16_t>, dim3(B), dim3(ST), 0, s, out, (const bf16_t*)logits, V, stride);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(argmax_kernel<f16_t>, dim3(B), dim3(ST), 0, s, out, (const f16_t*)logits, V, stride);
  else if (dtype == DT_F32)
    hipLaunchKernelGGL(argmax_kernel<float>, dim3(B), dim3(ST), 0, s, out, (const float*)logits, V, stride);
  else return -2;
  HIP_LAUNCH_CHECK();
  return 0;
}

VDAMD_API int vdamd_sample(int* out, const void* logits, int B, int V, long stride, const float* temperature,
                           const int* top_k, const float* top_p, const float* min_p, const float* uniform,
                           int dtype, hipStream_t s) {
  if (B == 0) return 0;
#define L(TY)                                                                                                  \
  hipLaunchKernelGGL(sample_kernel<TY>, dim3(B), dim3(ST), 0, s, out, (const TY*)logits, V, stride, temperature, \
                     top_k, top_p, min_p, uniform)
  if (dtype == DT_BF16) L(bf16_t);
  else if (dtype == DT_F16) L(f16_t);
  else if (dtype == DT_F32) L(float);
  else return -2;
#undef L
  HIP_LAUNCH_CHECK();
  return 0;
}
