// GEMM for large batches (M >= 512, compute-bound):  y[M][N] = x[M][K] . W[N][K]^T  (bf16 / f16, fp32
// accumulate), with the epilogues the decoder layer wants fused in: plain, "+ residual", and
// SiLU(gate) * up over a [gate ; up] weight.
//
// Structure (256 x 256 output tiles, one persistent 512-thread workgroup per CU, K walked in 64-deep tiles):
//
//  * Both operands are K-contiguous, so both go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`,
//    one wave instruction = 8 rows x 128 bytes; address = descriptor + per-lane offset + scalar K offset,
//    no arithmetic per load) and never pass through registers.  A K tile is four "half-tile" images of
//    128 rows x 128 bytes: A0, B0, A1, B1 (x rows 0-127 / 128-255 of the tile, W rows likewise).  The
//    16-byte slots of a row are XOR-ed with (row >> 1) & 7 - applied to the per-lane SOURCE address, the
//    LDS-DMA destination being lane-linear - so that the four 16-lane groups of a `ds_read_b128` fragment
//    read each hit 16 distinct slots of the 256-byte bank row: no bank conflicts (counter-verified).
//  * 8 waves as 2 (M) x 4 (N).  A wave's 128 x 64 outputs are four 64 x 32 quadrants, quadrant (mq, nq)
//    taken from half-tiles A<mq> and B<nq>; a K tile is four phases, one quadrant each over the full
//    64-deep tile (16 MFMAs 16x16x32).  Phase 0 reads A0 fragments, phase 1 A1, phase 2 B1, phase 3 the
//    B0 fragments of the NEXT K tile: 8, 8, 4, 4 `ds_read_b128` per phase, every fragment read once.
//  * The staging is one endless stream: half-tile h = 4 t + {A0, B0, A1, B1} is issued in phase g = h - 6
//    (two LDS buffers of four images; an image is re-staged two or more phases after its last read), and
//    each phase ends its issue block with a counted `s_waitcnt vmcnt(8)` (6 before the look-ahead read):
//    four half-tiles (64 KiB per CU) stay in flight across the barriers and nothing is ever drained.
//    The stream runs across output tiles: the last six phases of a tile already fetch the next tile's
//    first K tiles, and the output stores of a tile retire while the next tile's first half-tiles,
//    issued before them, are still arriving.
//  * Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA cluster while
//    its partner issues LDS reads and DMA for its next phase (12 % faster than lock-step, measured).
//  * The MFMA A operand is the W fragment, so a lane ends up with 4 consecutive output columns of one
//    row, and gate / up of the same column in the same lane for the SiLU epilogue; `v_permlane16_swap`
//    pairs the two column fragments of a lane with its neighbour row's, giving 16-byte stores.
//  * Workgroup -> tiles: bands of 8 row tiles; inside a band each XCD owns a range of tile columns and
//    its 32 workgroups hold an 8 x 4 block of tiles at a time, sharing x and W panels in that XCD's L2
//    (81 % L2 hit rate measured).
#include "common.h"

namespace {

constexpr int BG_THREADS = 512;
constexpr int BG_HALF = 128 * 128;          // bytes of one half-tile image (128 rows x 64 elements)
constexpr int BG_BUF = 4 * BG_HALF;         // A0, B0, A1, B1 of one K tile
enum { BG_EPI_NONE = 0, BG_EPI_RESIDUAL = 1, BG_EPI_SILU_MUL = 2 };   // as in stream_gemm.hip

typedef __attribute__((address_space(3))) void bg_lds_void;
typedef unsigned bg_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned bg_u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ bg_u32x2 bg_pack4(float a, float b, float c, float d) {
  typename Vec4<T>::type o;
  o[0] = (T)a; o[1] = (T)b; o[2] = (T)c; o[3] = (T)d;
  return __builtin_bit_cast(bg_u32x2, o);
}

// VAR bit 0: 8-byte stores (one per fragment) instead of the paired 16-byte ones.
template <typename T, int EPI, int VAR>
__global__ __launch_bounds__(BG_THREADS) void big_gemm_kernel(
    const T* x, long xs, const T* w, long ws, T* out, long os, const T* res, long rs, int M, int N, int K,
    int ntm, int ntn, int up_row) {
#if defined(__HIP_DEVICE_COMPILE__)      // the buffer-descriptor builtins exist in the device pass only
  using V8 = typename Vec8<T>::type;
  using V4 = typename Vec4<T>::type;
  constexpr bool WIDE = !(VAR & 1);
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BG_BUF];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int r16 = lane & 15, q = lane >> 4;

  // ---- this workgroup's tiles.  The output is cut into bands of 8 row tiles; inside a band XCD i
  // (workgroup ids are dealt round-robin to the XCDs) owns a contiguous range of tile columns and walks
  // it column by column, so at any time the 32 workgroups of an XCD hold an 8 x 4 block of tiles sharing
  // 12 operand panels in that XCD's L2, and all XCDs work on the same band of x (which then stays in
  // the memory-side cache while its W panels stream past).
  const int xcd = blockIdx.x & 7, stride = gridDim.x >> 3;
  const int c0 = (int)((long)ntn * xcd >> 3), c1 = (int)((long)ntn * (xcd + 1) >> 3);
  const int ncols = c1 - c0;
  const int chunk_end = ntm * ncols;         // tiles of this XCD
  int tile = blockIdx.x >> 3;                // index into this XCD's tile list
  if (tile >= chunk_end) return;
  auto decode = [&](int tl, int& m0_, int& tn_) {
    const int band = tl / (8 * ncols);
    const int rows = min(8, ntm - 8 * band);
    const int inband = tl - band * 8 * ncols;
    m0_ = (8 * band + inband % rows) * 256;
    tn_ = c0 + inband / rows;
  };

  // ---- staging: this wave moves pieces 2 wid and 2 wid + 1 (8 rows x 128 B each) of every half-tile
  unsigned offA[2][2], offB[2][2];          // [half][piece] byte offsets of this lane's 16 bytes
  auto set_stage_tile = [&](int tl) {
    // lane-derived constants are recomputed here (from an opaque copy of the lane id) rather than kept
    // in registers across the K loop, where there are none to spare
    int ln = lane;
    asm volatile("" : "+v"(ln));
    int m0, tn;
    decode(tl, m0, tn);
    // first W row of the two B halves, and the row limit the first is clamped to
    const int nb0 = EPI == BG_EPI_SILU_MUL ? tn * 128 : tn * 256;
    const int nb1 = EPI == BG_EPI_SILU_MUL ? up_row + tn * 128 : tn * 256 + 128;
    const int nl0 = EPI == BG_EPI_SILU_MUL ? up_row : N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 8 * (2 * wid + i) + (ln >> 3);
      const int c = (ln & 7) ^ ((r >> 1) & 7);
      // (all byte offsets fit 32 bits: checked on the host)
      offA[0][i] = ((unsigned)min(m0 + r, M - 1) * (unsigned)xs + 8 * c) * 2;
      offA[1][i] = ((unsigned)min(m0 + 128 + r, M - 1) * (unsigned)xs + 8 * c) * 2;
      offB[0][i] = ((unsigned)min(nb0 + r, nl0 - 1) * (unsigned)ws + 8 * c) * 2;
      offB[1][i] = ((unsigned)min(nb1 + r, N - 1) * (unsigned)ws + 8 * c) * 2;
    }
  };
  const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)((long)M * xs * 2), 0x00020000);
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(w), 0, (int)((long)N * ws * 2), 0x00020000);
  const unsigned obytes = (unsigned)((long)M * os * 2);
  const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)obytes, 0x00020000);
  const int KT = K / 64;
  // K tile indices past the end of the tile address the staging tile's first K tiles: by then the
  // offsets are those of the next output tile
  auto stage = [&](int kind, int buf, int kt) {     // kind: 0 A0, 1 B0, 2 A1, 3 B1
    const int kb = (kt >= KT ? kt - KT : kt) * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned off = kind == 0 ? offA[0][i] : kind == 1 ? offB[0][i] : kind == 2 ? offA[1][i] : offB[1][i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds((kind & 1) ? wrsrc : xrsrc,
                                               (bg_lds_void*)(smem + buf * BG_BUF + kind * BG_HALF + (2 * wid + i) * 1024),
                                               16, off, kb, 0, 0);
    }
  };

  // ---- fragment reads: lane (r16, q), k-step s reads 16-byte slot (4 s + q) ^ swz of its row
  const int swz = (r16 >> 1) & 7;
  const int so0 = (q ^ swz) << 4, so1 = ((4 + q) ^ swz) << 4;
  const int xro = (wr * 64 + r16) * 128, wro = (wc * 32 + r16) * 128;
  auto read_x = [&](V8 (&dst)[4][2], int buf, int half) {
    const unsigned char* b = smem + buf * BG_BUF + (half ? 2 : 0) * BG_HALF + xro;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i][0] = *reinterpret_cast<const V8*>(b + i * 2048 + so0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i][1] = *reinterpret_cast<const V8*>(b + i * 2048 + so1);
  };
  auto read_w = [&](V8 (&dst)[2][2], int buf, int half) {
    const unsigned char* b = smem + buf * BG_BUF + (half ? 3 : 1) * BG_HALF + wro;
#pragma unroll
    for (int j = 0; j < 2; ++j) dst[j][0] = *reinterpret_cast<const V8*>(b + j * 2048 + so0);
#pragma unroll
    for (int j = 0; j < 2; ++j) dst[j][1] = *reinterpret_cast<const V8*>(b + j * 2048 + so1);
  };

  f32x4 acc00[4][2], acc10[4][2], acc11[4][2], acc01[4][2];     // [mq][nq] quadrants, [m frag][n frag]
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
  // flight (three when the next phase reads one half-tile further ahead), meet the other waves,
  // multiply, meet again.  What the NEXT phase reads is older than what is left in flight, so after the
  // first barrier it is in LDS for every wave.
#define BG_PHASE(READS, KIND, BUF, KTILE, WAITN, ACC, XF, WF)                          \
  READS;                                                                               \
  stage(KIND, BUF, KTILE);                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");                         \
  __builtin_amdgcn_s_barrier();                                                        \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  __builtin_amdgcn_s_setprio(1);                                                       \
  mma(ACC, XF, WF);                                                                    \
  __builtin_amdgcn_s_setprio(0);                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  __builtin_amdgcn_s_barrier();                                                        \
  __builtin_amdgcn_sched_barrier(0);

  int cm0, ctn;                             // the tile being computed (the staging tile runs ahead)
  decode(tile, cm0, ctn);
  set_stage_tile(tile);
  // half-tiles 0..5: all of K tile 0, A0 and B0 of K tile 1
  stage(0, 0, 0); stage(1, 0, 0); stage(2, 0, 0); stage(3, 0, 0);
  stage(0, 1, 1); stage(1, 1, 1);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) __builtin_amdgcn_s_barrier();          // waves 4-7 run one barrier behind
  __builtin_amdgcn_sched_barrier(0);
  read_w(wf0, 0, 0);

  for (;;) {
    const int next = tile + stride;
    const bool has_next = next < chunk_end;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc00[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc10[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc11[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc01[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    // The output stores of the tile before are still in flight here; being older than the half-tile
    // issued in the first phase they are waited for by that phase's counted wait (they have the four
    // phases in which they were issued and the six half-tiles already in flight to hide behind).
    for (int t = 0; t < KT; t += 2) {
      BG_PHASE(read_x(xf0, 0, 0), 2, 1, t + 1, 8, acc00, xf0, wf0)
      BG_PHASE(read_x(xf1, 0, 1), 3, 1, t + 1, 8, acc10, xf1, wf0)
      {
        // everything staged from here on in the last pair of K tiles belongs to the next output tile
        int tt = t;
        asm volatile("" : "+s"(tt));       // one loop body for every iteration: nothing to peel or hoist
        if (tt == KT - 2) {
          int nt = has_next ? next : tile;
          asm volatile("" : "+s"(nt));
          set_stage_tile(nt);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      BG_PHASE(read_w(wf1, 0, 1), 0, 0, t + 2, 6, acc11, xf1, wf1)
      BG_PHASE(read_w(wf0, 1, 0), 1, 0, t + 2, 8, acc01, xf0, wf1)
      BG_PHASE(read_x(xf0, 1, 0), 2, 0, t + 2, 8, acc00, xf0, wf0)
      BG_PHASE(read_x(xf1, 1, 1), 3, 0, t + 2, 8, acc10, xf1, wf0)
      BG_PHASE(read_w(wf1, 1, 1), 0, 1, t + 3, 6, acc11, xf1, wf1)
      BG_PHASE(read_w(wf0, 0, 0), 1, 1, t + 3, 8, acc01, xf0, wf1)
    }
    // nothing below may be computed before the K loop (it would be live across it)
    asm volatile("" : "+s"(cm0), "+s"(ctn));

    // ---- epilogue: lane (r16, q) of fragment (i, j) holds y[m][n .. n + 4], m = row base + 16 i + r16,
    // n = column base + 16 j + 4 q.  Rows and columns outside the matrix go to an out-of-range buffer
    // offset and are dropped by the hardware.
    int eln = lane;
    asm volatile("" : "+v"(eln));           // as above: the epilogue's lane constants are made here
    const int er16 = eln & 15, eq = eln >> 4;
    auto put = [&](const f32x4 (&a0)[4][2], const f32x4 (&a1)[4][2], int mq) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = cm0 + 128 * mq + 64 * wr + 16 * i + er16;
        constexpr int NH = EPI == BG_EPI_SILU_MUL ? 1 : 2;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const int nbase = EPI == BG_EPI_SILU_MUL ? ctn * 128 + wc * 32 : ctn * 256 + 128 * h + wc * 32;
          const int nlim = EPI == BG_EPI_SILU_MUL ? up_row : N;
          bg_u32x2 pk[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int n = nbase + 16 * j + 4 * eq;
            float v[4];
            if constexpr (EPI == BG_EPI_SILU_MUL) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float g = a0[i][j][r];
                v[r] = fast_silu(g) * a1[i][j][r];
              }
            } else {
              const f32x4 a = h ? a1[i][j] : a0[i][j];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = a[r];
              if constexpr (EPI == BG_EPI_RESIDUAL) {
                if (m < M && n < nlim) {
                  const V4 rv = *reinterpret_cast<const V4*>(res + (long)m * rs + n);
#pragma unroll
                  for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                }
              }
            }
            pk[j] = bg_pack4<T>(v[0], v[1], v[2], v[3]);
          }
          if constexpr (WIDE) {
            // rows q and q ^ 1 of the lane grid trade halves: even rows end up with columns 4 q .. 4 q + 8
            // of fragment 0, odd rows with columns 4 (q - 1) .. 4 (q - 1) + 8 of fragment 1
            const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
            const bg_u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            const int n = nbase + ((eq & 1) ? 16 + 4 * (eq - 1) : 4 * eq);
            const unsigned off = (m < M && n < nlim) ? (unsigned)(((long)m * os + n) * 2) : obytes;
            __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, off, 0, 0);
          } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int n = nbase + 16 * j + 4 * eq;
              const unsigned off = (m < M && n < nlim) ? (unsigned)(((long)m * os + n) * 2) : obytes;
              __builtin_amdgcn_raw_buffer_store_b64(pk[j], orsrc, off, 0, 0);
            }
          }
        }
      }
    };
    put(acc00, acc01, 0);
    put(acc10, acc11, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!has_next) break;
    tile = next;
    decode(tile, cm0, ctn);
  }
#undef BG_PHASE
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the tail prefetches must not outlive the LDS
#endif
}

int bg_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus = n;
  }
  return cus;
}

template <typename T, int EPI, int VAR>
int launch(void* out, long os, const void* x, long xs, const void* w, long ws, const void* res, long rs, int M, int N,
           int K, int up_row, hipStream_t st) {
  const int ntm = (M + 255) / 256;
  const int ntn = EPI == BG_EPI_SILU_MUL ? (up_row + 127) / 128 : (N + 255) / 256;
  const int per = ntm * ((ntn + 7) / 8);                     // most tiles any XCD gets
  const int grid = 8 * (per < bg_num_cus() / 8 ? per : bg_num_cus() / 8);
  big_gemm_kernel<T, EPI, VAR><<<grid, BG_THREADS, 0, st>>>((const T*)x, xs, (const T*)w, ws, (T*)out, os,
                                                            (const T*)res, rs, M, N, K, ntm, ntn, up_row);
  HIP_LAUNCH_CHECK();
  return 0;
}

template <typename T, int VAR>
int dispatch(void* out, long os, const void* x, long xs, const void* w, long ws, const void* res, long rs, int M, int N,
             int K, int epi, hipStream_t st) {
  if (epi == BG_EPI_NONE) return launch<T, BG_EPI_NONE, VAR>(out, os, x, xs, w, ws, res, rs, M, N, K, 0, st);
  if (epi == BG_EPI_SILU_MUL)
    return launch<T, BG_EPI_SILU_MUL, VAR>(out, os, x, xs, w, ws, res, rs, M, N, K, N / 2, st);
  if (epi == BG_EPI_RESIDUAL) return launch<T, BG_EPI_RESIDUAL, VAR>(out, os, x, xs, w, ws, res, rs, M, N, K, 0, st);
  return -6;
}

}  // namespace

// out = epilogue(x[M][K] . w[N][K]^T).  K % 128 == 0, N % 16 == 0 (N % 32 == 0 with epi 2); strides in
// elements, multiples of 8; pointers 16-byte aligned; x, w and out each span less than 4 GiB.
// epi 0: out[M][N];  epi 1: out[M][N] = product + res;  epi 2: w = [gate ; up], out[M][N/2] = silu(gate) * up.
// variant 0: 16-byte output stores; 1: 8-byte ones.
VDAMD_API int vdamd_big_gemm(void* out, long out_stride, const void* x, long x_stride, const void* w, long w_stride,
                             const void* res, long res_stride, int M, int N, int K, int epi, int variant,
                             int dtype, void* stream) {
  if (M <= 0) return 0;
  if (K % 128 || N % 16 || (epi == BG_EPI_SILU_MUL && N % 32)) return -1;
  if (epi == BG_EPI_RESIDUAL && res == nullptr) return -2;
  if ((x_stride | w_stride | out_stride) % 8 || (epi == BG_EPI_RESIDUAL && res_stride % 4)) return -5;
  const long lim = (1L << 32) - 64;
  if ((long)M * x_stride * 2 >= lim || (long)N * w_stride * 2 >= lim || (long)M * out_stride * 2 >= lim) return -7;
  if (((uintptr_t)out | (uintptr_t)x | (uintptr_t)w) % 16) return -8;
  hipStream_t st = (hipStream_t)stream;
#define BG_VAR(V)                                                                                              \
  if (variant == V && dtype == DT_BF16)                                                                        \
    return dispatch<bf16_t, V>(out, out_stride, x, x_stride, w, w_stride, res, res_stride, M, N, K, epi, st);
  BG_VAR(0) BG_VAR(1)
#undef BG_VAR
  if (dtype == DT_F16 && variant == 0)
    return dispatch<f16_t, 0>(out, out_stride, x, x_stride, w, w_stride, res, res_stride, M, N, K, epi, st);
  return -3;
}
