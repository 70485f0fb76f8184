// GEMM for decode batches of 17..512 rows:  y[M][N] = x[M][K] . W[N][K]^T  (bf16 / f16, fp32 accumulate).
//
// In this range a step is bound by reading W once from HBM while the MFMA work per weight byte is
// already substantial (M flops per byte), and the output has too few 256 x 256 tiles to fill 256 CUs
// (N = 8192 gives 32).  Hence:
//
//  * one workgroup tile spans BM rows - all of M when M <= 256 - so every W byte is fetched from
//    HBM exactly once, and K is split across workgroups (SPLITK slices; fp32 partial slabs and a
//    small reduce kernel) until the grid covers the chip;
//  * both operands go global -> LDS with the 16-byte LDS-DMA (`global_load_lds_dwordx4`), in
//    pieces of 8 rows x 128 bytes (full cache lines of each row, one KiB per wave instruction), into
//    a double-buffered image; the next K tile's DMA is in flight while the MFMAs of this one run,
//    and there is exactly one barrier per 64-deep K tile;
//  * the LDS image keeps 128-byte rows; its 16-byte slots are XOR-ed with (row >> 1) & 7, applied to
//    the per-lane *source* address of the DMA (the destination of an LDS-DMA is lane-linear) and
//    again on the read, which makes every 16-lane group of a `ds_read_b128` fragment read hit 16
//    distinct slots of the 256-byte bank row;
//  * 8 waves as 2 (M) x 4 (N) on v_mfma_f32_16x16x32: W is the A operand, so a lane ends up with
//    four consecutive output columns of one row;
//  * workgroup id -> (tile, slice) puts equal slices on the same XCD (ids are dealt round-robin to
//    the 8 XCDs), so an XCD's L2 only has to hold its slice of x.
#include "common.h"

namespace {

constexpr int MG_BK = 64;            // K elements per tile = 128 bytes per row
constexpr int MG_THREADS = 512;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <typename T, int BM, int BN>
__global__ __launch_bounds__(MG_THREADS) void mid_gemm_kernel(
    const T* __restrict__ x, long x_stride, const T* __restrict__ w, long w_stride, T* __restrict__ out,
    long out_stride, float* __restrict__ partial, int M, int N, int K, int ksplit) {
  using V8 = typename Vec8<T>::type;
  constexpr int ROWS = BM + BN;                 // rows of one stage: x rows, then W rows
  constexpr int STAGE = ROWS * 128;             // bytes
  constexpr int PIECES = ROWS / 8;              // 1 KiB DMA pieces per stage
  constexpr int PPW = PIECES / 8;               // pieces per wave
  constexpr int FM = BM / 2 / 16, FN = BN / 4 / 16;
  static_assert(PIECES % 8 == 0, "tile rows must split evenly over 8 waves");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int r16 = lane & 15, q = lane >> 4;

  const int ntn = (N + BN - 1) / BN;
  const int id = blockIdx.x;
  const int ks = id % ksplit, tn = id / ksplit;
  if (tn >= ntn) return;
  const int m0 = blockIdx.y * BM, n0 = tn * BN;
  const int KT = K / MG_BK;
  const int kt_lo = (int)((long)KT * ks / ksplit), kt_hi = (int)((long)KT * (ks + 1) / ksplit);
  const int nt = kt_hi - kt_lo;

  // ---- DMA source addresses: piece p covers tile rows 8p..8p+7; this lane brings the 16-byte chunk
  // (lane & 7) ^ swz(row) of row 8p + (lane >> 3), which lands at LDS slot (lane & 7) of that row.
  const T* src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wid * PPW + i;
    const int row = 8 * p + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (row < BM) src[i] = x + (long)min(m0 + row, M - 1) * x_stride + 8 * chunk;
    else src[i] = w + (long)min(n0 + row - BM, N - 1) * w_stride + 8 * chunk;
    src[i] += (long)kt_lo * MG_BK;
  }
  auto stage = [&](int buf, int t) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      unsigned char* dst = smem + buf * STAGE + (wid * PPW + i) * 1024;
      __builtin_amdgcn_global_load_lds((glb_void*)(src[i] + (long)t * MG_BK), (lds_void*)dst, 16, 0, 0);
    }
  };

  // ---- fragment read offsets inside a stage (bytes), for k32-step s: XOR the slot with swz(row)
  int xoff[FM], woff[FN], xswz[FM], wswz[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = wm * (BM / 2) + 16 * i + r16;
    xoff[i] = row * 128;
    xswz[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int row = BM + wn * (BN / 4) + 16 * j + r16;
    woff[j] = row * 128;
    wswz[j] = (row >> 1) & 7;
  }

  f32x4 acc[FN][FM];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nt > 0) stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // tile t has landed; everyone has finished reading the other buffer
    if (t + 1 < nt) stage((t + 1) & 1, t + 1);
    const unsigned char* base = smem + (t & 1) * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      V8 a[FN], b[FM];
#pragma unroll
      for (int j = 0; j < FN; ++j)
        a[j] = *reinterpret_cast<const V8*>(base + woff[j] + (((4 * s + q) ^ wswz[j]) << 4));
#pragma unroll
      for (int i = 0; i < FM; ++i)
        b[i] = *reinterpret_cast<const V8*>(base + xoff[i] + (((4 * s + q) ^ xswz[i]) << 4));
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[j][i] = mfma16(a[j], b[i], acc[j][i]);
    }
  }

  // ---- epilogue: lane (r16, q) of fragment (j, i) holds y[m][n .. n+3], m = row r16 of x fragment i,
  // n = 4q of W fragment j
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * (BM / 2) + 16 * i + r16;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wn * (BN / 4) + 16 * j + 4 * q;
      if (n >= N) continue;                  // N % 16 == 0: a fragment column group is all in or all out
      if (ksplit == 1) {
        typename Vec4<T>::type o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)acc[j][i][r];
        *reinterpret_cast<typename Vec4<T>::type*>(out + (long)m * out_stride + n) = o;
      } else {
        *reinterpret_cast<f32x4*>(partial + ((size_t)ks * M + m) * N + n) = acc[j][i];
      }
    }
  }
}

template <typename T>
__global__ void mid_reduce_kernel(T* __restrict__ out, long out_stride, const float* __restrict__ partial, int M,
                                  int N, int ksplit) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n4 = N >> 2;
  if (idx >= (long)M * n4) return;
  const int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < ksplit; ++k) s += *reinterpret_cast<const f32x4*>(partial + ((size_t)k * M + m) * N + n);
  typename Vec4<T>::type o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = (T)s[r];
  *reinterpret_cast<typename Vec4<T>::type*>(out + (long)m * out_stride + n) = o;
}

template <typename T, int BM, int BN>
int launch(const void* x, long xs, const void* w, long ws, void* out, long os, void* partial, int M, int N, int K,
           int ksplit, hipStream_t st) {
  const int ntn = (N + BN - 1) / BN;
  dim3 grid(ntn * ksplit, (M + BM - 1) / BM);
  mid_gemm_kernel<T, BM, BN><<<grid, MG_THREADS, 0, st>>>((const T*)x, xs, (const T*)w, ws, (T*)out, os,
                                                          (float*)partial, M, N, K, ksplit);
  HIP_LAUNCH_CHECK();
  if (ksplit > 1) {
    const long n = (long)M * (N / 4);
    mid_reduce_kernel<T><<<(unsigned)((n + 255) / 256), 256, 0, st>>>((T*)out, os, (const float*)partial, M, N,
                                                                     ksplit);
    HIP_LAUNCH_CHECK();
  }
  return 0;
}

template <typename T>
int dispatch(const void* x, long xs, const void* w, long ws, void* out, long os, void* partial, int M, int N, int K,
             int ksplit, int bm, int bn, hipStream_t st) {
  if (bm == 256 && bn == 256) return launch<T, 256, 256>(x, xs, w, ws, out, os, partial, M, N, K, ksplit, st);
  if (bm == 256 && bn == 128) return launch<T, 256, 128>(x, xs, w, ws, out, os, partial, M, N, K, ksplit, st);
  if (bm == 128 && bn == 256) return launch<T, 128, 256>(x, xs, w, ws, out, os, partial, M, N, K, ksplit, st);
  if (bm == 128 && bn == 128) return launch<T, 128, 128>(x, xs, w, ws, out, os, partial, M, N, K, ksplit, st);
  if (bm == 64 && bn == 256) return launch<T, 64, 256>(x, xs, w, ws, out, os, partial, M, N, K, ksplit, st);
  return -4;
}

}  // namespace

// out[M][N] = x[M][K] . w[N][K]^T.  K % 64 == 0, N % 16 == 0; strides in elements, multiples of 8;
// pointers 16-byte aligned.  (bm, bn) is the workgroup tile: (256|128, 256|128) or (64, 256).
// partial holds ksplit * M * N floats when ksplit > 1.
VDAMD_API int vdamd_mid_gemm(void* out, long out_stride, const void* x, long x_stride, const void* w, long w_stride,
                             void* partial, int M, int N, int K, int ksplit, int bm, int bn, int dtype,
                             void* stream) {
  if (M <= 0) return 0;
  if (K % MG_BK || N % 16 || ksplit < 1 || ksplit > K / MG_BK) return -1;
  if (ksplit > 1 && partial == nullptr) return -2;
  if ((x_stride | w_stride) % 8 || out_stride % 4) return -5;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16)
    return dispatch<bf16_t>(x, x_stride, w, w_stride, out, out_stride, partial, M, N, K, ksplit, bm, bn, st);
  if (dtype == DT_F16)
    return dispatch<f16_t>(x, x_stride, w, w_stride, out, out_stride, partial, M, N, K, ksplit, bm, bn, st);
  return -3;
}
