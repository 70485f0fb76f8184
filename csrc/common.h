// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VDAMD_API extern "C" __attribute__((visibility("default")))

// dtype codes shared with ops/hip.py
enum { DT_BF16 = 0, DT_F16 = 1, DT_F32 = 2 };

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;

template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> { using type = bf16x8; };
template <> struct Vec8<f16_t> { using type = f16x8; };
template <typename T> struct Vec4;
template <> struct Vec4<bf16_t> { using type = bf16x4; };
template <> struct Vec4<f16_t> { using type = f16x4; };

// 16-byte vector load/store of 8 half-precision elements
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type ld8(const T* p) {
  return *reinterpret_cast<const typename Vec8<T>::type*>(p);
}
template <typename T>
__device__ __forceinline__ void st8(T* p, typename Vec8<T>::type v) {
  *reinterpret_cast<typename Vec8<T>::type*>(p) = v;
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// 2^x as the one hardware instruction (v_exp_f32).  The library's exp2f wraps it in a compare, two selects, an
// add and a scale so that results below 2^-126 come out as denormals; a softmax weight that small is zero for
// every purpose here, and the wrapper is five of the eight VALU operations an attention kernel spends per score.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// x * sigmoid(x) with the hardware reciprocal (1 ulp) in place of an IEEE division: the division's scale / fix-up
// sequence is ten instructions per element, and every result here is rounded to 16 bits right after.
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + fast_exp2(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_silu(float x) { return x * fast_sigmoid(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum over up to 16 waves; every thread gets the result.
// `red` must hold >= 17 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = (lane < nw) ? red[lane] : 0.f;
  t = wave_sum(t);
  __syncthreads();
  return t;
}

#define HIP_LAUNCH_CHECK()                                  \
  do {                                                      \
    hipError_t e_ = hipGetLastError();                      \
    if (e_ != hipSuccess) return (int)e_;                   \
  } while (0)
