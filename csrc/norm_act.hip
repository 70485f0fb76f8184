// RMSNorm / fused residual-add + RMSNorm / LayerNorm / SiLU*mul / embedding gather.
// All are HBM-bandwidth-bound: 16-byte vector accesses, fp32 accumulation,
// one workgroup per row with the row held in registers between the two phases.
#include "common.h"

namespace {

constexpr int NORM_MAX_ITERS = 4;   // 8 elems * threads * 4 >= H

// x (and residual) are read once; the row stays in registers across the reduction.
template <typename T, bool ADD_RESIDUAL>
__global__ void rms_norm_kernel(T* __restrict__ out, const T* __restrict__ x, T* __restrict__ residual,
                                const T* __restrict__ w, float eps, int H, long x_stride, long out_stride) {
  __shared__ float red[17];
  const long row = blockIdx.x;
  const T* xr = x + row * x_stride;
  T* rr = ADD_RESIDUAL ? residual + row * (long)H : nullptr;
  T* orow = out + row * out_stride;
  using V = typename Vec8<T>::type;
  V vals[NORM_MAX_ITERS], gw[NORM_MAX_ITERS];
  float ss = 0.f;
  const int nvec = H >> 3;
  // (the norm weight is fetched with the row, not after the reduction: it is cold in every layer, and a load
  // issued behind the block-wide sum is a second trip to HBM in a kernel that is all latency at decode sizes)
#pragma unroll
  for (int it = 0; it < NORM_MAX_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) gw[it] = ld8<T>(w + 8 * i);
  }
#pragma unroll
  for (int it = 0; it < NORM_MAX_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      V v = ld8<T>(xr + 8 * i);
      if (ADD_RESIDUAL) {
        V r = ld8<T>(rr + 8 * i);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (T)((float)v[j] + (float)r[j]);
        st8<T>(rr + 8 * i, v);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { float f = (float)v[j]; ss += f * f; }
      vals[it] = v;
    }
  }
  ss = block_sum(ss, red);
  const float inv = rsqrtf(ss / (float)H + eps);
#pragma unroll
  for (int it = 0; it < NORM_MAX_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      V v = vals[it];
      V g = gw[it];
      V o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (T)((float)v[j] * inv * (float)g[j]);
      st8<T>(orow + 8 * i, o);
    }
  }
}

// The same with the row arriving as `ks` fp32 partial sums (the K-split slabs of the GEMM that produced it,
// [ks][rows][H]): v = sum of the slabs + residual; residual = v; out = v * rsqrt(mean(v^2) + eps) * w.
// One kernel where the split GEMM's own reduction, a bf16 round trip and the add + norm kernel ran before.
template <typename T>
__global__ void reduce_add_rms_norm_kernel(T* __restrict__ out, const float* __restrict__ partial, int ks,
                                           long slab_stride, T* __restrict__ residual, const T* __restrict__ w,
                                           float eps, int H, long out_stride) {
  __shared__ float red[17];
  const long row = blockIdx.x;
  const float* pr = partial + row * (long)H;
  T* rr = residual + row * (long)H;
  T* orow = out + row * out_stride;
  using V = typename Vec8<T>::type;
  V vals[NORM_MAX_ITERS], gw[NORM_MAX_ITERS];
  float ss = 0.f;
  const int nvec = H >> 3;
#pragma unroll
  for (int it = 0; it < NORM_MAX_ITERS; ++it) {          // (norm weight with the row: see rms_norm_kernel)
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) gw[it] = ld8<T>(w + 8 * i);
  }
#pragma unroll
  for (int it = 0; it < NORM_MAX_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      const V r = ld8<T>(rr + 8 * i);
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
      // four slabs' loads in flight at a time (a plain loop over a run-time ks waits for each in turn)
      int k = 0;
      for (; k + 4 <= ks; k += 4) {
        f32x4 la[4], lb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          la[u] = *reinterpret_cast<const f32x4*>(pr + (k + u) * slab_stride + 8 * i);
          lb[u] = *reinterpret_cast<const f32x4*>(pr + (k + u) * slab_stride + 8 * i + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a += la[u]; b += lb[u]; }
      }
      for (; k < ks; ++k) {
        a += *reinterpret_cast<const f32x4*>(pr + k * slab_stride + 8 * i);
        b += *reinterpret_cast<const f32x4*>(pr + k * slab_stride + 8 * i + 4);
      }
      V v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = (T)(a[j] + (float)r[j]);
        v[j + 4] = (T)(b[j] + (float)r[j + 4]);
      }
      st8<T>(rr + 8 * i, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { float f = (float)v[j]; ss += f * f; }
      vals[it] = v;
    }
  }
  ss = block_sum(ss, red);
  const float inv = rsqrtf(ss / (float)H + eps);
#pragma unroll
  for (int it = 0; it < NORM_MAX_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      V g = gw[it], o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (T)((float)vals[it][j] * inv * (float)g[j]);
      st8<T>(orow + 8 * i, o);
    }
  }
}

template <typename T>
__global__ void layer_norm_kernel(T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ w,
                                  const T* __restrict__ b, float eps, int H) {
  __shared__ float red[17];
  const long row = blockIdx.x;
  const T* xr = x + row * (long)H;
  T* orow = out + row * (long)H;
  using V = typename Vec8<T>::type;
  V vals[NORM_MAX_ITERS];
  float s = 0.f;
  const int nvec = H >> 3;
#pragma unroll
  for (int it = 0; it < NORM_MAX_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      V v = ld8<T>(xr + 8 * i);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[j];
      vals[it] = v;
    }
  }
  const float mean = block_sum(s, red) / (float)H;
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < NORM_MAX_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = (float)vals[it][j] - mean; ss += d * d; }
    }
  }
  const float inv = rsqrtf(block_sum(ss, red) / (float)H + eps);
#pragma unroll
  for (int it = 0; it < NORM_MAX_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      V g = ld8<T>(w + 8 * i);
      V o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (T)(((float)vals[it][j] - mean) * inv * (float)g[j]);
      if (b != nullptr) {
        V bb = ld8<T>(b + 8 * i);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)((float)o[j] + (float)bb[j]);
      }
      st8<T>(orow + 8 * i, o);
    }
  }
}

template <typename T>
__global__ void silu_and_mul_kernel(T* __restrict__ out, const T* __restrict__ x, long T_rows, int F) {
  using V = typename Vec8<T>::type;
  // (32-bit index math - the launcher refuses more than 2^31 vectors: a 64-bit division per vector is a
  // ~150-instruction routine in a kernel that otherwise spends ~40)
  const unsigned nvec_row = F >> 3;
  const unsigned total = (unsigned)T_rows * nvec_row;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned ru = i / nvec_row;
    const long r = ru, c = i - ru * nvec_row;
    const T* xr = x + r * 2L * F;
    V g = ld8<T>(xr + 8 * c);
    V u = ld8<T>(xr + F + 8 * c);
    V o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = (float)g[j];
      o[j] = (T)(fast_silu(a) * (float)u[j]);
    }
    st8<T>(out + r * (long)F + 8 * c, o);
  }
}

// One workgroup per token: copy the embedding row, or zeros when the id is
// outside this rank's vocabulary shard.
template <typename T>
__global__ void embedding_kernel(T* __restrict__ out, const int* __restrict__ ids, const T* __restrict__ w,
                                 int H, int vocab_start, int rows) {
  using V = typename Vec8<T>::type;
  const long t = blockIdx.x;
  const int id = ids[t] - vocab_start;
  const bool ok = id >= 0 && id < rows;
  const T* src = w + (long)(ok ? id : 0) * H;
  T* dst = out + t * (long)H;
  V zero;
#pragma unroll
  for (int j = 0; j < 8; ++j) zero[j] = (T)0.f;
  for (int i = threadIdx.x; i < (H >> 3); i += blockDim.x) st8<T>(dst + 8 * i, ok ? ld8<T>(src + 8 * i) : zero);
}

inline int norm_threads(int H) {
  int t = ((H >> 3) + NORM_MAX_ITERS - 1) / NORM_MAX_ITERS;   // threads so that 4 iterations cover the row
  t = (t + 63) / 64 * 64;
  if (t < 64) t = 64;
  if (t > 1024) t = 1024;
  return t;
}

}  // namespace

// out may alias x.  residual (if non-null) is updated in place to x + residual.
VDAMD_API int vdamd_rms_norm(void* out, const void* x, void* residual, const void* w, float eps, int T,
                             int H, long x_stride, long out_stride, int dtype, hipStream_t s) {
  if (T == 0) return 0;
  if (H % 8 != 0 || H > 8 * 1024 * NORM_MAX_ITERS) return -1;
  const int threads = norm_threads(H);
#define LAUNCH(TY, ADD)                                                                                   \
  hipLaunchKernelGGL((rms_norm_kernel<TY, ADD>), dim3(T), dim3(threads), 0, s, (TY*)out, (const TY*)x,    \
                     (TY*)residual, (const TY*)w, eps, H, x_stride, out_stride)
  if (dtype == DT_BF16) { if (residual) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false); }
  else if (dtype == DT_F16) { if (residual) LAUNCH(f16_t, true); else LAUNCH(f16_t, false); }
  else return -2;
#undef LAUNCH
  HIP_LAUNCH_CHECK();
  return 0;
}

// out[T][H] = RMSNorm(sum_k partial[k][T][H] + residual) * w; residual is updated in place to the sum.
VDAMD_API int vdamd_reduce_add_rms_norm(void* out, const float* partial, int ks, void* residual, const void* w,
                                        float eps, int T, int H, long out_stride, int dtype, hipStream_t s) {
  if (T == 0) return 0;
  if (H % 8 != 0 || H > 8 * 1024 * NORM_MAX_ITERS || ks < 1 || residual == nullptr) return -1;
  int threads = norm_threads(H);
  if (T <= 64) {               // a few rows: one pass of a wider workgroup, the launch is latency-bound
    threads = ((H >> 3) + 63) / 64 * 64;
    if (threads > 1024) threads = 1024;
  }
  const long slab = (long)T * H;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(reduce_add_rms_norm_kernel<bf16_t>, dim3(T), dim3(threads), 0, s, (bf16_t*)out, partial, ks,
                       slab, (bf16_t*)residual, (const bf16_t*)w, eps, H, out_stride);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(reduce_add_rms_norm_kernel<f16_t>, dim3(T), dim3(threads), 0, s, (f16_t*)out, partial, ks,
                       slab, (f16_t*)residual, (const f16_t*)w, eps, H, out_stride);
  else return -2;
  HIP_LAUNCH_CHECK();
  return 0;
}

VDAMD_API int vdamd_layer_norm(void* out, const void* x, const void* w, const void* b, float eps, int T, int H,
                               int dtype, hipStream_t s) {
  if (T == 0) return 0;
  if (H % 8 != 0 || H > 8 * 1024 * NORM_MAX_ITERS) return -1;
  const int threads = norm_threads(H);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(layer_norm_kernel<bf16_t>, dim3(T), dim3(threads), 0, s, (bf16_t*)out, (const bf16_t*)x,
                       (const bf16_t*)w, (const bf16_t*)b, eps, H);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(layer_norm_kernel<f16_t>, dim3(T), dim3(threads), 0, s, (f16_t*)out, (const f16_t*)x,
                       (const f16_t*)w, (const f16_t*)b, eps, H);
  else return -2;
  HIP_LAUNCH_CHECK();
  return 0;
}

VDAMD_API int vdamd_silu_and_mul(void* out, const void* x, long T, int F, int dtype, hipStream_t s) {
  if (T == 0) return 0;
  if (F % 8 != 0) return -1;
  const long total = T * (long)(F >> 3);
  if (total >= (1L << 31)) return -3;
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;     // 8 workgroups per CU, grid-stride the rest
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(silu_and_mul_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (bf16_t*)out,
                       (const bf16_t*)x, T, F);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(silu_and_mul_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (f16_t*)out,
                       (const f16_t*)x, T, F);
  else return -2;
  HIP_LAUNCH_CHECK();
  return 0;
}

VDAMD_API int vdamd_embedding(void* out, const int* ids, const void* w, int T, int H, int vocab_start, int rows,
                              int dtype, hipStream_t s) {
  if (T == 0) return 0;
  if (H % 8 != 0) return -1;
  int threads = ((H >> 3) + 63) / 64 * 64;
  if (threads > 256) threads = 256;
  if (dtype == DT_BF16 || dtype == DT_F16)   // pure 2-byte copy: one instantiation serves both
    hipLaunchKernelGGL(embedding_kernel<bf16_t>, dim3(T), dim3(threads), 0, s, (bf16_t*)out, ids,
                       (const bf16_t*)w, H, vocab_start, rows);
  else return -2;
  HIP_LAUNCH_CHECK();
  return 0;
}
