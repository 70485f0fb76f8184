// Dynamic per-token FP8 (OCP e4m3, the gfx950 native format) quantisation of
// activations for W8A8 linears: one workgroup per row keeps the row in
// registers, reduces |x|max, and writes x * (448 / max) as fp8 with the packed
// hardware convert (v_cvt_pk_fp8_f32 saturates and rounds to nearest even).
// scale[row] = max / 448 is what the GEMM multiplies back in.
#include "common.h"

namespace {

constexpr int FP8Q_MAX_ITERS = 4;
constexpr float FP8_E4M3_MAX = 448.f;

template <typename T>
__global__ void dynamic_fp8_quant_kernel(uint8_t* __restrict__ out, float* __restrict__ scale,
                                         const T* __restrict__ x, int K, long x_stride) {
  __shared__ float red[17];
  using V = typename Vec8<T>::type;
  const long row = blockIdx.x;
  const T* xr = x + row * x_stride;
  const int nvec = K >> 3;
  V vals[FP8Q_MAX_ITERS];
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < FP8Q_MAX_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      vals[it] = ld8<T>(xr + 8 * i);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf((float)vals[it][j]));
    }
  }
  // block max via the shared sum helper's pattern
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  amax = wave_max(amax);
  if (lane == 0) red[wid] = amax;
  __syncthreads();
  float t = lane < nw ? red[lane] : 0.f;
  amax = wave_max(t);
  const float s = fmaxf(amax, 1e-12f) / FP8_E4M3_MAX;
  const float inv = 1.f / s;
  if (threadIdx.x == 0) scale[row] = s;
  uint8_t* orow = out + row * (long)K;
#pragma unroll
  for (int it = 0; it < FP8Q_MAX_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)vals[it][0] * inv, (float)vals[it][1] * inv, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)vals[it][2] * inv, (float)vals[it][3] * inv, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)vals[it][4] * inv, (float)vals[it][5] * inv, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)vals[it][6] * inv, (float)vals[it][7] * inv, hi, true);
      *reinterpret_cast<int2*>(orow + 8 * i) = make_int2(lo, hi);
    }
  }
}

}  // namespace

// out[M][K] fp8 e4m3 (contiguous), scale[M] fp32.  K % 8 == 0, K <= 8 * 1024 * 4.
VDAMD_API int vdamd_dynamic_fp8_quant(void* out, float* scale, const void* x, int M, int K, long x_stride,
                                      int dtype, hipStream_t s) {
  if (M == 0) return 0;
  if (K % 8 != 0 || K > 8 * 1024 * FP8Q_MAX_ITERS) return -1;
  const int nvec = K >> 3;
  int threads = 64;
  while (threads < 1024 && threads * FP8Q_MAX_ITERS < nvec) threads <<= 1;
  if (threads * FP8Q_MAX_ITERS < nvec) return -1;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(dynamic_fp8_quant_kernel<bf16_t>, dim3(M), dim3(threads), 0, s, (uint8_t*)out, scale,
                       (const bf16_t*)x, K, x_stride);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL(dynamic_fp8_quant_kernel<f16_t>, dim3(M), dim3(threads), 0, s, (uint8_t*)out, scale,
                       (const f16_t*)x, K, x_stride);
  else
    return -2;
  HIP_LAUNCH_CHECK();
  return 0;
}
