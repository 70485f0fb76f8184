// GEMM for large batches (M >= 512, compute-bound):  y[M][N] = x[M][K] . W[N][K]^T  (bf16 / f16, fp32
// accumulate), with the epilogues the decoder layer wants fused in: plain, SiLU(gate) * up over a
// [gate ; up] weight, and "+ residual".
//
// Structure (one 256 x 256 output tile per 512-thread workgroup, K walked in 64-deep tiles):
//
//  * Both operands are K-contiguous, so both go global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, one
//    wave instruction = 8 rows x 128 bytes) and never pass through registers.  A K tile is four
//    "half-tile" images of 128 rows x 128 bytes: A0, B0, A1, B1 (x rows 0-127 / 128-255 of the tile, W
//    rows likewise).  The 16-byte slots of a row are XOR-ed with (row >> 1) & 7 - applied to the per-lane
//    SOURCE address, the LDS-DMA destination being lane-linear - so that the four 16-lane groups of a
//    `ds_read_b128` fragment read each hit 16 distinct slots of the 256-byte bank row: no bank conflicts.
//  * 8 waves as 2 (M) x 4 (N).  A wave's 128 x 64 outputs are four 64 x 32 quadrants, quadrant (mq, nq)
//    taken from half-tiles A<mq> and B<nq>; a K tile is four phases, one quadrant each over the full
//    64-deep tile (16 MFMAs 16x16x32).  Phase 0 reads A0 and B0 fragments, phase 1 A1, phase 2 B1,
//    phase 3 nothing: every fragment is read from LDS once per K tile and half-tile images are needed -
//    and free again - one after the other, not all at once.
//  * That makes the staging a simple stream: half-tile h = 4 t + {A0, B0, A1, B1} is issued in phase
//    g = h - 6 (two LDS buffers of four images; an image is re-staged two or three phases after its last
//    read), and each phase ends its issue block with a counted `s_waitcnt vmcnt(8)`: four half-tiles
//    (64 KiB per CU) stay in flight across the barriers, nothing is ever drained inside the loop.
//  * Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA cluster while
//    its partner issues LDS reads and DMA for its next phase.
//  * The MFMA A operand is the W fragment, so a lane ends up with 4 consecutive output columns of one
//    row: 8-byte stores, and gate / up of the same column in the same lane for the SiLU epilogue.
//  * Workgroup id -> tile: ids are dealt round-robin to the 8 XCDs, so XCD i gets the i-th contiguous
//    chunk of the column-major tile order: the row tiles of a W panel run side by side on one L2.
#include "common.h"

namespace {

constexpr int BG_THREADS = 512;
constexpr int BG_HALF = 128 * 128;          // bytes of one half-tile image (128 rows x 64 elements)
constexpr int BG_BUF = 4 * BG_HALF;         // A0, B0, A1, B1 of one K tile
enum { BG_EPI_NONE = 0, BG_EPI_RESIDUAL = 1, BG_EPI_SILU_MUL = 2 };   // as in stream_gemm.hip

typedef __attribute__((address_space(3))) void bg_lds_void;
typedef const __attribute__((address_space(1))) void bg_glb_void;

template <typename T, int EPI>
__global__ __launch_bounds__(BG_THREADS) void big_gemm_kernel(
    const T* x, long xs, const T* w, long ws, T* out, long os, const T* res, long rs, int M, int N, int K,
    int ntm, int ntn, int up_row) {
  using V8 = typename Vec8<T>::type;
  using V4 = typename Vec4<T>::type;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BG_BUF];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int r16 = lane & 15, q = lane >> 4;

  const int per = gridDim.x >> 3;
  const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tile >= ntm * ntn) return;
  const int tm = tile % ntm, tn = tile / ntm;
  const int m0 = tm * 256;
  // first W row of the two B halves, and the row limit each is clamped to
  const int nb0 = EPI == BG_EPI_SILU_MUL ? tn * 128 : tn * 256;
  const int nb1 = EPI == BG_EPI_SILU_MUL ? up_row + tn * 128 : tn * 256 + 128;
  const int nl0 = EPI == BG_EPI_SILU_MUL ? up_row : N;

  // ---- staging: this wave moves pieces 2 wid and 2 wid + 1 (8 rows x 128 B each) of every half-tile
  unsigned offA[2][2], offB[2][2];          // [half][piece] byte offsets of this lane's 16 bytes
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (2 * wid + i) + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    offA[0][i] = (unsigned)(((long)min(m0 + r, M - 1) * xs + 8 * c) * 2);
    offA[1][i] = (unsigned)(((long)min(m0 + 128 + r, M - 1) * xs + 8 * c) * 2);
    offB[0][i] = (unsigned)(((long)min(nb0 + r, nl0 - 1) * ws + 8 * c) * 2);
    offB[1][i] = (unsigned)(((long)min(nb1 + r, N - 1) * ws + 8 * c) * 2);
  }
  const char* xb = reinterpret_cast<const char*>(x);
  const char* wb = reinterpret_cast<const char*>(w);
  const int KT = K / 64;
  auto stage = [&](int kind, int buf, int kt) {     // kind: 0 A0, 1 B0, 2 A1, 3 B1
    const long kb = (long)min(kt, KT - 1) * 128;
    const char* base = ((kind & 1) ? wb : xb) + kb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned off = kind == 0 ? offA[0][i] : kind == 1 ? offB[0][i] : kind == 2 ? offA[1][i] : offB[1][i];
      __builtin_amdgcn_global_load_lds((bg_glb_void*)(base + off),
                                       (bg_lds_void*)(smem + buf * BG_BUF + kind * BG_HALF + (2 * wid + i) * 1024),
                                       16, 0, 0);
    }
  };

  // ---- fragment reads: lane (r16, q), k-step s reads 16-byte slot (4 s + q) ^ swz of its row
  const int swz = (r16 >> 1) & 7;
  const int so0 = (q ^ swz) << 4, so1 = ((4 + q) ^ swz) << 4;
  const int xro = (wr * 64 + r16) * 128, wro = (wc * 32 + r16) * 128;
  auto read_x = [&](V8 (&dst)[4][2], int buf, int half) {
    const unsigned char* b = smem + buf * BG_BUF + (half ? 2 : 0) * BG_HALF + xro;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dst[i][0] = *reinterpret_cast<const V8*>(b + i * 2048 + so0);
      dst[i][1] = *reinterpret_cast<const V8*>(b + i * 2048 + so1);
    }
  };
  auto read_w = [&](V8 (&dst)[2][2], int buf, int half) {
    const unsigned char* b = smem + buf * BG_BUF + (half ? 3 : 1) * BG_HALF + wro;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      dst[j][0] = *reinterpret_cast<const V8*>(b + j * 2048 + so0);
      dst[j][1] = *reinterpret_cast<const V8*>(b + j * 2048 + so1);
    }
  };

  f32x4 acc00[4][2], acc10[4][2], acc11[4][2], acc01[4][2];     // [mq][nq] quadrants, [m frag][n frag]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc00[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc10[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc11[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc01[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  auto mma = [&](f32x4 (&acc)[4][2], const V8 (&xf)[4][2], const V8 (&wf)[2][2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(wf[j][s], xf[i][s], acc[i][j]);
  };
  V8 xf0[4][2], xf1[4][2], wf0[2][2], wf1[2][2];

  // One phase: issue this phase's fragment reads and one half-tile of DMA, leave four half-tiles in
  // flight, meet the other waves, multiply, meet again.  The half-tile the NEXT phase reads is older
  // than the four left in flight, so after the first barrier it is in LDS for every wave.
#define BG_PHASE(READS, KIND, BUF, KTILE, ACC, XF, WF)              \
  READS;                                                            \
  stage(KIND, BUF, KTILE);                                          \
  __builtin_amdgcn_sched_barrier(0);                                \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                  \
  __builtin_amdgcn_s_barrier();                                     \
  __builtin_amdgcn_sched_barrier(0);                                \
  __builtin_amdgcn_s_setprio(1);                                    \
  mma(ACC, XF, WF);                                                 \
  __builtin_amdgcn_s_setprio(0);                                    \
  __builtin_amdgcn_sched_barrier(0);                                \
  __builtin_amdgcn_s_barrier();                                     \
  __builtin_amdgcn_sched_barrier(0);

  // prologue: half-tiles 0..5 (all of K tile 0, A0 and B0 of K tile 1)
  stage(0, 0, 0); stage(1, 0, 0); stage(2, 0, 0); stage(3, 0, 0);
  stage(0, 1, 1); stage(1, 1, 1);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) __builtin_amdgcn_s_barrier();          // waves 4-7 run one barrier behind
  __builtin_amdgcn_sched_barrier(0);

  for (int t = 0; t < KT; t += 2) {
    BG_PHASE(read_w(wf0, 0, 0); __builtin_amdgcn_sched_barrier(0); read_x(xf0, 0, 0), 2, 1, t + 1, acc00, xf0, wf0)
    BG_PHASE(read_x(xf1, 0, 1), 3, 1, t + 1, acc10, xf1, wf0)
    BG_PHASE(read_w(wf1, 0, 1), 0, 0, t + 2, acc11, xf1, wf1)
    BG_PHASE((void)0, 1, 0, t + 2, acc01, xf0, wf1)
    BG_PHASE(read_w(wf0, 1, 0); __builtin_amdgcn_sched_barrier(0); read_x(xf0, 1, 0), 2, 0, t + 2, acc00, xf0, wf0)
    BG_PHASE(read_x(xf1, 1, 1), 3, 0, t + 2, acc10, xf1, wf0)
    BG_PHASE(read_w(wf1, 1, 1), 0, 1, t + 3, acc11, xf1, wf1)
    BG_PHASE((void)0, 1, 1, t + 3, acc01, xf0, wf1)
  }
#undef BG_PHASE
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the clamped tail prefetches must not outlive the LDS

  // ---- epilogue: lane (r16, q) of fragment (i, j) holds y[m][n .. n + 4], m = row base + 16 i + r16,
  // n = column base + 16 j + 4 q
  auto store_rows = [&](const f32x4 (&a0)[4][2], const f32x4 (&a1)[4][2], int mq) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 128 * mq + 64 * wr + 16 * i + r16;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cn = wc * 32 + 16 * j + 4 * q;
        if constexpr (EPI == BG_EPI_SILU_MUL) {
          const int n = tn * 128 + cn;
          if (n >= up_row) continue;
          V4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g = a0[i][j][r];
            o[r] = (T)(g / (1.f + __expf(-g)) * a1[i][j][r]);
          }
          *reinterpret_cast<V4*>(out + (long)m * os + n) = o;
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int n = tn * 256 + 128 * h + cn;
            if (n >= N) continue;
            const f32x4 a = h ? a1[i][j] : a0[i][j];
            V4 o;
            if constexpr (EPI == BG_EPI_RESIDUAL) {
              const V4 rv = *reinterpret_cast<const V4*>(res + (long)m * rs + n);
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = (T)(a[r] + (float)rv[r]);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = (T)a[r];
            }
            *reinterpret_cast<V4*>(out + (long)m * os + n) = o;
          }
        }
      }
    }
  };
  store_rows(acc00, acc01, 0);
  store_rows(acc10, acc11, 1);
}

template <typename T, int EPI>
int launch(void* out, long os, const void* x, long xs, const void* w, long ws, const void* res, long rs, int M, int N,
           int K, int up_row, hipStream_t st) {
  const int ntm = (M + 255) / 256;
  const int ntn = EPI == BG_EPI_SILU_MUL ? (up_row + 127) / 128 : (N + 255) / 256;
  const int grid = (ntm * ntn + 7) / 8 * 8;
  big_gemm_kernel<T, EPI><<<grid, BG_THREADS, 0, st>>>((const T*)x, xs, (const T*)w, ws, (T*)out, os, (const T*)res,
                                                       rs, M, N, K, ntm, ntn, up_row);
  HIP_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int dispatch(void* out, long os, const void* x, long xs, const void* w, long ws, const void* res, long rs, int M, int N,
             int K, int epi, hipStream_t st) {
  if (epi == BG_EPI_NONE) return launch<T, BG_EPI_NONE>(out, os, x, xs, w, ws, res, rs, M, N, K, 0, st);
  if (epi == BG_EPI_SILU_MUL) return launch<T, BG_EPI_SILU_MUL>(out, os, x, xs, w, ws, res, rs, M, N, K, N / 2, st);
  if (epi == BG_EPI_RESIDUAL) return launch<T, BG_EPI_RESIDUAL>(out, os, x, xs, w, ws, res, rs, M, N, K, 0, st);
  return -6;
}

}  // namespace

// out = epilogue(x[M][K] . w[N][K]^T).  K % 128 == 0, N % 16 == 0 (N % 32 == 0 with epi 2); strides in
// elements, multiples of 8; pointers 16-byte aligned; x and w each span less than 4 GiB.
// epi 0: out[M][N];  epi 1: out[M][N] = product + res;  epi 2: w = [gate ; up], out[M][N/2] = silu(gate) * up.
VDAMD_API int vdamd_big_gemm(void* out, long out_stride, const void* x, long x_stride, const void* w, long w_stride,
                             const void* res, long res_stride, int M, int N, int K, int epi, int dtype,
                             void* stream) {
  if (M <= 0) return 0;
  if (K % 128 || N % 16 || (epi == BG_EPI_SILU_MUL && N % 32)) return -1;
  if (epi == BG_EPI_RESIDUAL && res == nullptr) return -2;
  if ((x_stride | w_stride) % 8 || out_stride % 4 || (epi == BG_EPI_RESIDUAL && res_stride % 4)) return -5;
  if ((long)M * x_stride * 2 >= (1L << 32) || (long)N * w_stride * 2 >= (1L << 32)) return -7;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16)
    return dispatch<bf16_t>(out, out_stride, x, x_stride, w, w_stride, res, res_stride, M, N, K, epi, st);
  if (dtype == DT_F16)
    return dispatch<f16_t>(out, out_stride, x, x_stride, w, w_stride, res, res_stride, M, N, K, epi, st);
  return -3;
}
