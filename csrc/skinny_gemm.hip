// Weight-streaming GEMM for short batches (M <= 128): y[M][N] = x[M][K] . W[N][K]^T.
//
// At these sizes the product is bound by reading W once from HBM, and the
// library picks wide-M tiles that leave a third of the bandwidth unused.  Here
// W is the MFMA *A* operand straight from global memory (the swapped product
// y^T = W . x^T on 16x16x32 tiles): for a 128-column block of K, lane
// (c = lane & 15, g = lane >> 4) reads W[n0 + c][128 kb + 32 s + 8 g ..+8] for
// MFMA step s: one instruction covers 64 contiguous bytes of each of 16 rows,
// and the four steps of a block complete two 128-byte lines per row back to
// back.  The four waves of a workgroup own different 16-row tiles and
// share the x block through LDS.  A ds_read_b128 is served in four groups of
// 16 lanes that mix two values of g ({0-3,12-15,20-27}, ...), so padded rows
// still collide; instead the 16-byte slot of a 256-byte row is XOR-ed with the
// row index (slot ^ (row & 15)), which makes every group hit 16 distinct slots.  Next block's W and x are in flight while the current one is
// multiplied.  K may be split over grid.y; partial sums are then reduced by a
// second kernel.
#include "common.h"

constexpr int SG_XPAD = 128;   // 256-byte rows; 16-byte slots XOR-swizzled by the row (see below)
constexpr int SG_WAVES = 4;

// grid = (ceil(N / (64*NTW)), ksplit), block = 256.  Requires K % 128 == 0, N % 16 == 0, M <= 16*MT.
// SILU: x is [M][2K] = [gate | up] and the operand is silu(gate) * up, formed on the way into LDS
// (up_off = K): the activation kernel between the two MLP GEMMs and its round trip disappear.
template <typename T, int MT, int NTW, bool SILU = false>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(
    const T* __restrict__ x, long x_stride, const T* __restrict__ w, long w_stride, T* __restrict__ out,
    long out_stride, float* __restrict__ partial, int M, int N, int K, int ksplit,
    const int* __restrict__ tiles, long w_group_stride, long up_off = 0) {
  using V8 = typename Vec8<T>::type;
  __shared__ __attribute__((aligned(16))) T xl[2][MT * 16][SG_XPAD];
  if (tiles != nullptr) {
    // grouped mode (mixture-of-experts): blockIdx.z names a row tile
    // (group, first row, rows) of the group-sorted x; rows == 0 marks padding
    const int* d = tiles + 3 * blockIdx.z;
    M = d[2];
    if (M == 0) return;
    w += (size_t)d[0] * w_group_stride;
    x += (long)d[1] * x_stride;
    out += (long)d[1] * out_stride;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int KB = K >> 7, ntiles = N >> 4;
  const int nt0 = (blockIdx.x * SG_WAVES + wid) * NTW;
  const int ks = blockIdx.y;
  const int kb_lo = KB * ks / ksplit, kb_hi = KB * (ks + 1) / ksplit;

  const T* wrow[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
    wrow[t] = w + (size_t)(16 * min(nt0 + t, ntiles - 1) + c) * w_stride + 8 * g;
  // cooperative copy of x: thread -> (row xr of a 16-row pass, 16-byte column xc)
  const int xr = threadIdx.x >> 4, xc = threadIdx.x & 15;
  const T* xsrc[MT];
#pragma unroll
  for (int p = 0; p < MT; ++p) xsrc[p] = x + (long)min(16 * p + xr, M - 1) * x_stride + 8 * xc;

  f32x4 acc[NTW][MT];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_w = [&](int kb, V8 (&a)[NTW][4]) {
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) a[t][s] = ld8(wrow[t] + (kb << 7) + 32 * s);
  };
  auto load_x = [&](int p, int kb) -> V8 {
    V8 gt = ld8(xsrc[p] + (kb << 7));
    if constexpr (SILU) {
      const V8 up = ld8(xsrc[p] + up_off + (kb << 7));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gv = (float)gt[j];
        gt[j] = (T)(fast_silu(gv) * (float)up[j]);
      }
    }
    return gt;
  };
  V8 ac[NTW][4], an[NTW][4];
  V8 xreg[MT];
  if (kb_lo < kb_hi) {
    load_w(kb_lo, ac);
#pragma unroll
    for (int p = 0; p < MT; ++p) st8(&xl[0][16 * p + xr][8 * (xc ^ xr)], load_x(p, kb_lo));
  }
  __syncthreads();
  for (int kb = kb_lo; kb < kb_hi; ++kb) {
    const int buf = (kb - kb_lo) & 1;
    const bool more = kb + 1 < kb_hi;
    if (more) {
      load_w(kb + 1, an);
#pragma unroll
      for (int p = 0; p < MT; ++p) xreg[p] = load_x(p, kb + 1);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      V8 b[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) b[mt] = ld8(&xl[buf][16 * mt + c][8 * ((4 * s + g) ^ c)]);
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = mfma16(ac[t][s], b[mt], acc[t][mt]);
    }
    if (more) {
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) ac[t][s] = an[t][s];
#pragma unroll
      for (int p = 0; p < MT; ++p) st8(&xl[buf ^ 1][16 * p + xr][8 * (xc ^ xr)], xreg[p]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    if (nt0 + t >= ntiles) continue;
    const int n = 16 * (nt0 + t) + 4 * g;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = 16 * mt + c;
      if (m >= M) continue;
      if (ksplit == 1) {
        typename Vec4<T>::type o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)acc[t][mt][r];
        *reinterpret_cast<typename Vec4<T>::type*>(out + (long)m * out_stride + n) = o;
      } else {
        *reinterpret_cast<f32x4*>(partial + ((size_t)ks * M + m) * N + n) = acc[t][mt];
      }
    }
  }
}

template <typename T>
__global__ void skinny_reduce_kernel(T* __restrict__ out, long out_stride, const float* __restrict__ partial,
                                     int M, int N, int ksplit) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n4 = N >> 2;
  if (idx >= (long)M * n4) return;
  const unsigned mu = (unsigned)idx / (unsigned)n4;        // (M * N / 4 < 2^31: no 64-bit division)
  const int m = (int)mu, n = (int)((unsigned)idx - mu * (unsigned)n4) * 4;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < ksplit; ++k) s += *reinterpret_cast<const f32x4*>(partial + ((size_t)k * M + m) * N + n);
  typename Vec4<T>::type o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = (T)s[r];
  *reinterpret_cast<typename Vec4<T>::type*>(out + (long)m * out_stride + n) = o;
}

template <typename T, int MT, int NTW>
static void launch_one(const void* x, long xs, const void* w, long ws, void* out, long os, void* partial, int M,
                       int N, int K, int ksplit, hipStream_t st, const int* tiles = nullptr, int num_tiles = 1,
                       long w_group_stride = 0, long up_off = 0) {
  dim3 grid((N / 16 + SG_WAVES * NTW - 1) / (SG_WAVES * NTW), ksplit, num_tiles);
  if (up_off)
    skinny_gemm_kernel<T, MT, NTW, true><<<grid, 256, 0, st>>>((const T*)x, xs, (const T*)w, ws, (T*)out, os,
                                                                (float*)partial, M, N, K, ksplit, tiles,
                                                                w_group_stride, up_off);
  else
    skinny_gemm_kernel<T, MT, NTW, false><<<grid, 256, 0, st>>>((const T*)x, xs, (const T*)w, ws, (T*)out, os,
                                                                 (float*)partial, M, N, K, ksplit, tiles,
                                                                 w_group_stride, 0);
}

template <typename T>
static int launch_skinny(const void* x, long xs, const void* w, long ws, void* out, long os, void* partial, int M,
                         int N, int K, int ksplit, int ntw, hipStream_t st, const int* tiles = nullptr,
                         int num_tiles = 1, long wgs = 0, long up = 0) {
  const int mt = (M + 15) / 16;        // grouped mode: M is the row capacity of a tile
#define SG_CASE(MT_)                                                                                            \
  if (ntw == 1) launch_one<T, MT_, 1>(x, xs, w, ws, out, os, partial, M, N, K, ksplit, st, tiles, num_tiles, wgs, up);       \
  else if (ntw == 2) launch_one<T, MT_, 2>(x, xs, w, ws, out, os, partial, M, N, K, ksplit, st, tiles, num_tiles, wgs, up);  \
  else launch_one<T, MT_, 4>(x, xs, w, ws, out, os, partial, M, N, K, ksplit, st, tiles, num_tiles, wgs, up)
  if (mt <= 1) { SG_CASE(1); }
  else if (mt <= 2) { SG_CASE(2); }
  else if (mt <= 4) { SG_CASE(4); }
  else {       // 8 row tiles: at most two output tiles per wave fit the register file
    if (ntw == 1) launch_one<T, 8, 1>(x, xs, w, ws, out, os, partial, M, N, K, ksplit, st, tiles, num_tiles, wgs, up);
    else launch_one<T, 8, 2>(x, xs, w, ws, out, os, partial, M, N, K, ksplit, st, tiles, num_tiles, wgs, up);
  }
#undef SG_CASE
  HIP_LAUNCH_CHECK();
  if (ksplit > 1 && out != nullptr) {            // out == nullptr: the caller consumes the partial sums itself
    const long n = (long)M * (N / 4);
    skinny_reduce_kernel<T><<<(unsigned)((n + 255) / 256), 256, 0, st>>>((T*)out, os, (const float*)partial, M, N,
                                                                        ksplit);
    HIP_LAUNCH_CHECK();
  }
  return 0;
}

// ntw = output tiles per wave (1, 2 or 4; 4 is clamped to 2 when M > 64).
// `partial` holds ksplit*M*N floats when ksplit > 1.  Strides in elements,
// multiples of 8; pointers 16-byte aligned.
// up_off != 0: x holds [gate | up] per row (up at x + up_off elements) and the product is taken with
// silu(gate) * up.
VDAMD_API int vdamd_skinny_gemm(void* out, long out_stride, const void* x, long x_stride, const void* w,
                                long w_stride, void* partial, int M, int N, int K, int ksplit, int ntw,
                                int dtype, long up_off, void* stream) {
  if (M <= 0) return 0;
  if (up_off % 8) return -5;
  if (M > 128 || K % 128 || N % 16 || ksplit < 1 || ksplit > K / 128) return -1;
  if (ntw != 1 && ntw != 2 && ntw != 4) return -4;
  if (ksplit > 1 && partial == nullptr) return -2;
  if (out == nullptr && ksplit == 1) return -2;      // partial sums only: there must be a split to leave them in
  if ((x_stride | w_stride) % 8 || out_stride % 4) return -5;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16)
    return launch_skinny<bf16_t>(x, x_stride, w, w_stride, out, out_stride, partial, M, N, K, ksplit, ntw, st, nullptr,
                                 1, 0, up_off);
  if (dtype == DT_F16)
    return launch_skinny<f16_t>(x, x_stride, w, w_stride, out, out_stride, partial, M, N, K, ksplit, ntw, st, nullptr,
                                1, 0, up_off);
  return -3;
}

// Grouped form for mixture-of-experts: x rows are sorted by group (expert) and
// cut into tiles of at most `tile_rows` rows; tiles[i] = (group, first row,
// rows) on the device, rows == 0 for padding entries.  w[g] = w + g * w_group_stride.
VDAMD_API int vdamd_skinny_gemm_grouped(void* out, long out_stride, const void* x, long x_stride, const void* w,
                                        long w_stride, long w_group_stride, const int* tiles, int num_tiles,
                                        int tile_rows, int N, int K, int ntw, int dtype, void* stream) {
  if (num_tiles <= 0) return 0;
  if (tile_rows <= 0 || tile_rows > 128 || tile_rows % 16 || K % 128 || N % 16 || tiles == nullptr) return -1;
  if (ntw != 1 && ntw != 2 && ntw != 4) return -4;
  if ((x_stride | w_stride) % 8 || out_stride % 4 || w_group_stride % 8) return -5;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16)
    return launch_skinny<bf16_t>(x, x_stride, w, w_stride, out, out_stride, nullptr, tile_rows, N, K, 1, ntw, st, tiles,
                                 num_tiles, w_group_stride);
  if (dtype == DT_F16)
    return launch_skinny<f16_t>(x, x_stride, w, w_stride, out, out_stride, nullptr, tile_rows, N, K, 1, ntw, st, tiles,
                                num_tiles, w_group_stride);
  return -3;
}
