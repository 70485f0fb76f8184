// Weight-streaming GEMM for decode batches of at most 16 rows:
//   y[M][N] = x[M][K] . W[N][K]^T   (+ optional residual, or SiLU(gate) * up for W = [gate ; up])
//
// One token step of a 70B-class model reads every weight once and multiplies
// it with a handful of rows, so the kernel is a memory copy with arithmetic on
// the side.  The design follows from that:
//
//  * no LDS and no barrier in the main loop.  W is the MFMA A operand straight
//    from global memory (y^T = W . x^T on 16x16x32 tiles; lane (c, g) reads
//    W[n0 + c][.. + 8g ..+8], 64 contiguous bytes of each of 16 rows per
//    instruction) and the x fragment - 16 rows x 64 bytes, the same shape -
//    comes from L2 with the same addressing.  x is at most 16 x K x 2 bytes and
//    stays resident in every XCD's L2; rows >= M read row M - 1 again.
//  * the four waves of a workgroup share one output tile and take every fourth
//    128-column block of K, so a workgroup walks 1 KiB of each of its 16 rows
//    at a time; partial sums meet once, in LDS, after the loop.
//  * DEPTH column blocks are in flight per wave (a register ring, unrolled at
//    compile time; loads retire in order, so the wait for block i leaves
//    blocks i+1.. outstanding).
//  * K can also be split across workgroups (grid.y) when N alone gives too few
//    workgroups; those partial sums go through an fp32 scratch buffer and a
//    small second kernel.
#include "common.h"

namespace {

constexpr int SW = 4;   // waves per workgroup, interleaved over K

enum Epilogue { EPI_NONE = 0, EPI_RESIDUAL = 1, EPI_SILU_MUL = 2 };

template <typename T, bool NT>
__device__ __forceinline__ typename Vec8<T>::type ldw(const T* p) {
  using V = typename Vec8<T>::type;
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
  else return *reinterpret_cast<const V*>(p);
}

// grid = (ceil(N/16 / NTW), ksplit); block = 256.
// EPI_SILU_MUL: W = [gate rows ; up rows], NTW == 2: a workgroup takes gate tile b and up tile b + N/32 and
// writes out[m][16 b ..] = silu(gate) * up, N/2 wide.
template <typename T, int NTW, int DEPTH, bool NT, int EPI>
__global__ __launch_bounds__(256) void stream_gemm_kernel(
    const T* __restrict__ x, long x_stride, const T* __restrict__ w, long w_stride, T* __restrict__ out,
    long out_stride, float* __restrict__ partial, const T* __restrict__ residual, long res_stride, int M, int N,
    int K, int ksplit) {
  using V8 = typename Vec8<T>::type;
  __shared__ f32x4 red[SW - 1][NTW][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int KB = K >> 7, ntiles = N >> 4;
  const int nt0 = EPI == EPI_SILU_MUL ? blockIdx.x : blockIdx.x * NTW;
  const int tstep = EPI == EPI_SILU_MUL ? (ntiles >> 1) : 1;      // distance between this workgroup's tiles
  const int ks = blockIdx.y;
  const int kb_lo = KB * ks / ksplit, kb_hi = KB * (ks + 1) / ksplit;
  const int iters = (kb_hi - kb_lo - wid + SW - 1) / SW;       // blocks kb_lo + wid, + SW, ...

  const T* wrow[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
    wrow[t] = w + (size_t)(16 * min(nt0 + t * tstep, ntiles - 1) + c) * w_stride + 8 * g +
              ((long)(kb_lo + wid) << 7);
  const T* xrow = x + (long)min(c, M - 1) * x_stride + 8 * g + ((long)(kb_lo + wid) << 7);

  f32x4 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  V8 wa[DEPTH][NTW][4], xb[DEPTH][4];
  auto load = [&](int slot, int it) {
    const long off = (long)it * (SW << 7);
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) wa[slot][t][s] = ldw<T, NT>(wrow[t] + off + 32 * s);
#pragma unroll
    for (int s = 0; s < 4; ++s) xb[slot][s] = ld8(xrow + off + 32 * s);
  };
#pragma unroll
  for (int p = 0; p < DEPTH; ++p)
    if (p < iters) load(p, p);
  for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
    for (int p = 0; p < DEPTH; ++p) {
      if (it + p < iters) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[t] = mfma16(wa[p][t][s], xb[p][s], acc[t]);
        if (it + p + DEPTH < iters) load(p, it + p + DEPTH);
      }
    }
  }

  // the four K-interleaved partial sums meet in wave 0
  if (wid > 0) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) red[wid - 1][t][lane] = acc[t];
  }
  __syncthreads();
  if (wid > 0) return;
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int v = 0; v < SW - 1; ++v) acc[t] += red[v][t][lane];

  const int m = c;
  if (m >= M) return;
  if (ksplit > 1) {
#pragma unroll
    for (int t = 0; t < NTW; ++t)
      if (nt0 + t * tstep < ntiles)
        *reinterpret_cast<f32x4*>(partial + ((size_t)ks * M + m) * N + 16 * (nt0 + t * tstep) + 4 * g) = acc[t];
    return;
  }
  if constexpr (EPI == EPI_SILU_MUL) {
    typename Vec4<T>::type o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gt = acc[0][r];
      o[r] = (T)(fast_silu(gt) * acc[NTW - 1][r]);
    }
    *reinterpret_cast<typename Vec4<T>::type*>(out + (long)m * out_stride + 16 * nt0 + 4 * g) = o;
  } else {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      if (nt0 + t >= ntiles) continue;
      const int n = 16 * (nt0 + t) + 4 * g;
      typename Vec4<T>::type o;
      if constexpr (EPI == EPI_RESIDUAL) {
        const typename Vec4<T>::type r4 =
            *reinterpret_cast<const typename Vec4<T>::type*>(residual + (long)m * res_stride + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)(acc[t][r] + (float)r4[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)acc[t][r];
      }
      *reinterpret_cast<typename Vec4<T>::type*>(out + (long)m * out_stride + n) = o;
    }
  }
}

// Sum of the K-split partials, with the same epilogues as the main kernel.
template <typename T, int EPI>
__global__ void stream_reduce_kernel(T* __restrict__ out, long out_stride, const float* __restrict__ partial,
                                     const T* __restrict__ residual, long res_stride, int M, int N, int ksplit) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (EPI == EPI_SILU_MUL) {
    const int n4 = N >> 3;                       // output quads: N/2 columns
    if (idx >= (long)M * n4) return;
    const unsigned mu = (unsigned)idx / (unsigned)n4;
    const int m = (int)mu, q = (int)((unsigned)idx - mu * (unsigned)n4);
    const int ng = 4 * q, nu = ng + (N >> 1);
    f32x4 sg = f32x4{0.f, 0.f, 0.f, 0.f}, su = sg;
    for (int k = 0; k < ksplit; ++k) {
      sg += *reinterpret_cast<const f32x4*>(partial + ((size_t)k * M + m) * N + ng);
      su += *reinterpret_cast<const f32x4*>(partial + ((size_t)k * M + m) * N + nu);
    }
    typename Vec4<T>::type o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (T)(fast_silu(sg[r]) * su[r]);
    *reinterpret_cast<typename Vec4<T>::type*>(out + (long)m * out_stride + ng) = o;
  } else {
    const int n4 = N >> 2;
    if (idx >= (long)M * n4) return;
    const unsigned mu = (unsigned)idx / (unsigned)n4;        // (M * N / 4 < 2^31: no 64-bit division)
  const int m = (int)mu, n = (int)((unsigned)idx - mu * (unsigned)n4) * 4;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ksplit; ++k) s += *reinterpret_cast<const f32x4*>(partial + ((size_t)k * M + m) * N + n);
    typename Vec4<T>::type o;
    if constexpr (EPI == EPI_RESIDUAL) {
      const typename Vec4<T>::type r4 =
          *reinterpret_cast<const typename Vec4<T>::type*>(residual + (long)m * res_stride + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (T)(s[r] + (float)r4[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (T)s[r];
    }
    *reinterpret_cast<typename Vec4<T>::type*>(out + (long)m * out_stride + n) = o;
  }
}

struct Args {
  const void *x, *w, *residual;
  void *out, *partial;
  long xs, ws, os, rs;
  int M, N, K, ksplit;
  hipStream_t st;
};

template <typename T, int NTW, int DEPTH, bool NT, int EPI>
void launch(const Args& a) {
  dim3 grid(EPI == EPI_SILU_MUL ? a.N / 32 : (a.N / 16 + NTW - 1) / NTW, a.ksplit);
  stream_gemm_kernel<T, NTW, DEPTH, NT, EPI><<<grid, 256, 0, a.st>>>(
      (const T*)a.x, a.xs, (const T*)a.w, a.ws, (T*)a.out, a.os, (float*)a.partial, (const T*)a.residual, a.rs, a.M,
      a.N, a.K, a.ksplit);
}

template <typename T, int EPI>
int dispatch(const Args& a, int ntw, int depth, int nt) {
#define SG_D(NTW_, D_)                                                      \
  if (ntw == NTW_ && depth == D_) {                                         \
    if (nt) launch<T, NTW_, D_, true, EPI>(a); else launch<T, NTW_, D_, false, EPI>(a); \
    ok = true;                                                              \
  }
  bool ok = false;
  if constexpr (EPI != EPI_SILU_MUL) { SG_D(1, 2) SG_D(1, 3) SG_D(1, 4) }
  SG_D(2, 2) SG_D(2, 3) SG_D(2, 4)
#undef SG_D
  if (!ok) return -4;
  HIP_LAUNCH_CHECK();
  if (a.ksplit > 1) {
    const long n = (long)a.M * (EPI == EPI_SILU_MUL ? a.N / 8 : a.N / 4);
    stream_reduce_kernel<T, EPI><<<(unsigned)((n + 255) / 256), 256, 0, a.st>>>(
        (T*)a.out, a.os, (const float*)a.partial, (const T*)a.residual, a.rs, a.M, a.N, a.ksplit);
    HIP_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

// out[M][N] (or [M][N/2] for epilogue 2) = x[M][K] . w[N][K]^T, M <= 16, K % 128 == 0, N % 16 == 0 (% 32 for
// epilogue 2).  epilogue: 0 none, 1 += residual[M][N], 2 silu(gate) * up with w = [gate rows ; up rows].
// ntw = output tiles per workgroup (1 | 2), depth = column blocks in flight per wave (2..4), nt = non-temporal
// weight loads.  partial holds ksplit * M * N floats when ksplit > 1.  Strides in elements.
VDAMD_API int vdamd_stream_gemm(void* out, long out_stride, const void* x, long x_stride, const void* w,
                                long w_stride, void* partial, const void* residual, long res_stride, int M, int N,
                                int K, int ksplit, int ntw, int depth, int nt, int epilogue, int dtype,
                                void* stream) {
  if (M <= 0) return 0;
  if (M > 16 || K % 128 || N % 16 || ksplit < 1 || ksplit > K / 128) return -1;
  if (ksplit > 1 && partial == nullptr) return -2;
  if ((x_stride | w_stride) % 8 || out_stride % 4) return -5;
  if (epilogue == EPI_RESIDUAL && (residual == nullptr || res_stride % 4)) return -6;
  if (epilogue == EPI_SILU_MUL && (N % 32 || ntw != 2)) return -7;
  Args a{x, w, residual, out, partial, x_stride, w_stride, out_stride, res_stride, M, N, K, ksplit,
         (hipStream_t)stream};
#define SG_E(T_)                                                                  \
  switch (epilogue) {                                                             \
    case EPI_NONE: return dispatch<T_, EPI_NONE>(a, ntw, depth, nt);              \
    case EPI_RESIDUAL: return dispatch<T_, EPI_RESIDUAL>(a, ntw, depth, nt);      \
    case EPI_SILU_MUL: return dispatch<T_, EPI_SILU_MUL>(a, ntw, depth, nt);      \
    default: return -8;                                                           \
  }
  if (dtype == DT_BF16) { SG_E(bf16_t) }
  if (dtype == DT_F16) { SG_E(f16_t) }
#undef SG_E
  return -3;
}
