// GEMM for decode batches of 17..512 rows:  y[M][N] = x[M][K] . W[N][K]^T  (bf16 / f16, fp32 accumulate).
//
// In this range a step is bound by reading W once from HBM while the MFMA work per weight byte is
// already substantial (M flops per byte), and the output has too few large tiles to fill 256 CUs
// (N = 8192 gives 32 tiles of 256 columns).  Hence:
//
//  * one workgroup tile spans BM rows - all of M when M <= 256 - so every W byte is fetched from
//    HBM exactly once, and K is split across workgroups (SPLITK slices; fp32 partial slabs and a
//    small reduce kernel) until the grid covers the chip;
//  * W never touches LDS.  The 8 waves of a workgroup sit side by side along N; a wave owns 16*FN
//    output columns for ALL rows of the tile, so its W fragments (the MFMA A operand: lane (r, q)
//    holds W[n + r][k + 8q .. +8], 16 rows x 64 bytes per instruction) are private to it and go from
//    global memory straight into a three-deep register ring: two K tiles of W (8 KiB per wave,
//    64 KiB per CU) are in flight while the third is multiplied, with no barrier in their way;
//  * x is shared by all waves and comes from L2: 128-byte lines into registers two tiles ahead,
//    into a double-buffered LDS image one tile ahead (16-byte slots XOR-ed with (row >> 1) & 7, so
//    the 16-lane groups of a `ds_read_b128` fragment read hit 16 distinct slots of the 256-byte
//    bank row), one barrier per 64-deep K tile;
//  * every load is an ordinary one, issued in the fixed order x(t+2), W(t+2): the compiler's own
//    counted `s_waitcnt vmcnt(N)` then leaves exactly the younger tiles in flight at each use;
//  * workgroup id -> (tile, slice) puts equal slices on the same XCD (ids are dealt round-robin to
//    the 8 XCDs), so an XCD's L2 only has to hold its slice of x.
#include "common.h"

namespace {

constexpr int MG_BK = 64;            // K elements per tile = 128 bytes per row
constexpr int MG_THREADS = 512;
constexpr int MG_WAVES = 8;

template <typename T, int BM, int FN>
__global__ __launch_bounds__(MG_THREADS) void mid_gemm_kernel(
    const T* x, long x_stride, const T* w, long w_stride, T* out, long out_stride, float* partial, int M, int N,
    int K, int ksplit) {
  // x and w are deliberately not __restrict__: loads from provably unaliased read-only memory may be
  // moved across the barriers, and the pipeline below depends on WHEN each load is issued.
  using V8 = typename Vec8<T>::type;
  constexpr int BN = MG_WAVES * 16 * FN;
  constexpr int FM = BM / 16;
  constexpr int XL = BM * 8 / MG_THREADS;       // 16-byte chunks of the x tile per thread
  constexpr int STAGE = BM * 128;               // bytes of one x image
  static_assert(XL >= 1 && FM % 2 == 0, "unsupported tile");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;

  const int ntn = (N + BN - 1) / BN;
  const int id = blockIdx.x;
  const int ks = id % ksplit, tn = id / ksplit;
  if (tn >= ntn) return;
  const int m0 = blockIdx.y * BM, n0 = tn * BN + wid * 16 * FN;
  const int KT = K / MG_BK;
  const int kt_lo = KT * ks / ksplit, kt_hi = KT * (ks + 1) / ksplit;
  const int nt = kt_hi - kt_lo;

  // W fragment rows of this lane (clamped: columns past N are computed and dropped)
  const T* wsrc[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j)
    wsrc[j] = w + (long)min(n0 + 16 * j + r16, N - 1) * w_stride + 8 * q + (long)kt_lo * MG_BK;
  // x staging: chunk c = threadIdx.x + 512 i -> row c >> 3, 16-byte slot c & 7 (8 threads = one 128-byte line)
  const T* xsrc[XL];
  int xdst[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int c = threadIdx.x + MG_THREADS * i;
    const int row = c >> 3, slot = c & 7;
    xsrc[i] = x + (long)min(m0 + row, M - 1) * x_stride + 8 * slot + (long)kt_lo * MG_BK;
    xdst[i] = row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
  }
  // x fragment reads: lane (r16, q) of fragment i reads row 16 i + r16, slot (4 s + q) ^ swz(row)
  int xoff = r16 * 128;                       // + 16 i * 128
  const int xswz = (r16 >> 1) & 7;            // (16 i + r16) >> 1 & 7 == r16 >> 1 & 7

  f32x4 acc[FN][FM];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  V8 wreg[3][FN][2], xreg[2][XL];
  auto load_w = [&](int slot, int t) {
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) wreg[slot][j][s] = ld8(wsrc[j] + (long)t * MG_BK + 32 * s);
  };
  auto load_x = [&](int slot, int t) {
#pragma unroll
    for (int i = 0; i < XL; ++i) xreg[slot][i] = ld8(xsrc[i] + (long)t * MG_BK);
  };
  auto write_x = [&](int slot, int stage) {
#pragma unroll
    for (int i = 0; i < XL; ++i) *reinterpret_cast<V8*>(smem + stage * STAGE + xdst[i]) = xreg[slot][i];
  };
  // One K tile: 2 k-steps x FM row fragments, taken two fragments at a time.  The reads of pair p + 1
  // are issued before the MFMAs of pair p, so an LDS round trip is covered by 2 * FN MFMAs of this
  // wave plus whatever the SIMD's other wave issues.
  auto compute = [&](int wslot, int stage) {
    const unsigned char* base = smem + stage * STAGE + xoff;
    constexpr int PAIRS = FM;                      // 2 k-steps * FM / 2
    const int so0 = ((q) ^ xswz) << 4, so1 = ((4 + q) ^ xswz) << 4;
    V8 b[2][2];
    auto rd = [&](int p, V8 (&dst)[2]) {
      const int s = p / (FM / 2), i = (p % (FM / 2)) * 2;
      const int so = s ? so1 : so0;
      dst[0] = *reinterpret_cast<const V8*>(base + i * 2048 + so);
      dst[1] = *reinterpret_cast<const V8*>(base + (i + 1) * 2048 + so);
    };
    rd(0, b[0]);
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      if (p + 1 < PAIRS) rd(p + 1, b[(p + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);           // or the reads sink down to their first use
      const int s = p / (FM / 2), i = (p % (FM / 2)) * 2;
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[j][i + e] = mfma16(wreg[wslot][j][s], b[p & 1][e], acc[j][i + e]);
    }
  };

  // Loads and LDS writes are unconditional (past the end they repeat the last tile and are never
  // used): a branch around a load would make the compiler's wait for it a full drain.
  if (nt <= 0) return;                     // block-uniform; cannot happen for ksplit <= K / 64
  const int last = nt - 1;
  load_x(0, 0); load_w(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load_x(1, min(1, last)); load_w(1, min(1, last));
  __builtin_amdgcn_sched_barrier(0);
  write_x(0, 0);
  __syncthreads();
  for (int t = 0; t < nt; t += 6) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int tt = t + u;
      if (tt >= nt) break;
      load_x(u & 1, min(tt + 2, last));
      load_w((u + 2) % 3, min(tt + 2, last));
      __builtin_amdgcn_sched_barrier(0);     // keep the issue order the in-flight accounting relies on
      compute(u % 3, u & 1);
      __builtin_amdgcn_sched_barrier(0);
      write_x((u + 1) & 1, (u + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: lane (r16, q) of fragment (j, i) holds y[m0 + 16 i + r16][n0 + 16 j + 4 q .. +4]
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + 16 * i + r16;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + 16 * j + 4 * q;
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

// ---------------------------------------------------------------------------------------------
// Deep-prefetch variant.  The loop above can keep two K tiles of W in flight and no more: loads
// retire in order, and the wait for the x registers of tile t+1 also waits for every W load issued
// before them.  Here x never enters a register: it goes global -> LDS by LDS-DMA
// (`global_load_lds_dwordx4`, 8 rows x 128 bytes per wave instruction, the XOR swizzle applied to
// the per-lane SOURCE address because the destination of an LDS-DMA is lane-linear) into a
// three-stage image, two tiles ahead; W goes into a four-slot register ring, four tiles ahead.
// Issue order per iteration is [x(t+2) | ... compute(t) ... | W(t+4)], so the one wait of an
// iteration, `vmcnt(12)`, retires x(t) - and with it W(t), W(t+1), which are older - while W(t+2),
// x(t+1), W(t+3) stay in flight: an HBM round trip has three to four tile times to complete.
// The W loads are inline asm: next to an LDS-DMA in flight the compiler would wait `vmcnt(0)` for
// any ordinary load it can see; their destinations are only read after the counted wait, which
// names them as read-write operands so that nothing can be scheduled across it.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <typename V8>
__device__ __forceinline__ void asm_load2(V8& a, V8& b, const void* p) {
  asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64"
               : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}

template <typename T, int BM, int FN>
__global__ __launch_bounds__(MG_THREADS) void mid_gemm_deep_kernel(
    const T* x, long x_stride, const T* w, long w_stride, T* out, long out_stride, float* partial, int M, int N,
    int K, int ksplit) {
  using V8 = typename Vec8<T>::type;
  constexpr int BN = MG_WAVES * 16 * FN;
  constexpr int FM = BM / 16;
  constexpr int STAGE = BM * 128;               // bytes of one x image
  constexpr int XS = 3;                         // x stages
  constexpr int WS = 4;                         // W ring slots
  constexpr int PPW = BM / 8 / MG_WAVES;        // 1 KiB DMA pieces per wave per tile
  static_assert(PPW >= 1 && FM % 2 == 0, "unsupported tile");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[XS * STAGE];

  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, q = lane >> 4;
  const int ntn = (N + BN - 1) / BN;
  const int id = blockIdx.x;
  const int ks = id % ksplit, tn = id / ksplit;
  if (tn >= ntn) return;
  const int m0 = blockIdx.y * BM, n0 = tn * BN + wid * 16 * FN;
  const int KT = K / MG_BK;
  const int kt_lo = KT * ks / ksplit, kt_hi = KT * (ks + 1) / ksplit;
  const int nt = kt_hi - kt_lo;
  if (nt <= 0) return;
  const int last = nt - 1;

  const T* wsrc[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j)
    wsrc[j] = w + (long)min(n0 + 16 * j + r16, N - 1) * w_stride + 8 * q + (long)kt_lo * MG_BK;
  // DMA piece p (8 rows x 128 B) of this wave: lane -> row 8p + (lane >> 3), source chunk (lane & 7) ^ swz(row)
  const T* xsrc[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = 8 * (wid * PPW + i) + (lane >> 3);
    xsrc[i] = x + (long)min(m0 + row, M - 1) * x_stride + 8 * ((lane & 7) ^ ((row >> 1) & 7)) + (long)kt_lo * MG_BK;
  }
  const int xoff = r16 * 128;
  const int xswz = (r16 >> 1) & 7;
  const int so0 = (q ^ xswz) << 4, so1 = ((4 + q) ^ xswz) << 4;

  f32x4 acc[FN][FM];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  V8 wreg[WS][FN][2];

  auto dma_x = [&](int stage, int t) {
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(xsrc[i] + (long)t * MG_BK),
                                       (lds_void*)(smem + stage * STAGE + (wid * PPW + i) * 1024), 16, 0, 0);
  };
  auto load_w = [&](int slot, int t) {
#pragma unroll
    for (int j = 0; j < FN; ++j) asm_load2(wreg[slot][j][0], wreg[slot][j][1], wsrc[j] + (long)t * MG_BK);
  };

  // prologue, in the order the steady-state accounting expects: W(0) W(1) x(0) W(2) x(1) W(3)
  load_w(0, 0);
  load_w(1, min(1, last));
  dma_x(0, 0);
  load_w(2, min(2, last));
  dma_x(1, min(1, last));
  load_w(3, min(3, last));

  for (int t = 0; t < nt; t += 12) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int tt = t + u;
      if (tt >= nt) break;
      const int stage = u % XS, slot = u % WS;
      // x(tt) and W(tt) have landed once only W(tt+2), x(tt+1), W(tt+3) are outstanding
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * 2 * FN + PPW) : "memory");
      // the W fragments of this slot are valid from here on: make every later use depend on this point
#pragma unroll
      for (int j = 0; j < FN; ++j)
        asm volatile("" : "+v"(wreg[slot][j][0]), "+v"(wreg[slot][j][1]));
      __builtin_amdgcn_s_barrier();               // every wave's pieces of x(tt) are in LDS; stage of x(tt-1) is free
      __builtin_amdgcn_sched_barrier(0);
      dma_x((u + 2) % XS, min(tt + 2, last));
      __builtin_amdgcn_sched_barrier(0);
      {
        const unsigned char* base = smem + stage * STAGE + xoff;
        V8 b[2][2];
        auto rd = [&](int p, V8 (&dst)[2]) {
          const int s = p / (FM / 2), i = (p % (FM / 2)) * 2;
          const int so = s ? so1 : so0;
          dst[0] = *reinterpret_cast<const V8*>(base + i * 2048 + so);
          dst[1] = *reinterpret_cast<const V8*>(base + (i + 1) * 2048 + so);
        };
        rd(0, b[0]);
#pragma unroll
        for (int p = 0; p < FM; ++p) {
          if (p + 1 < FM) rd(p + 1, b[(p + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          const int s = p / (FM / 2), i = (p % (FM / 2)) * 2;
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[j][i + e] = mfma16(wreg[slot][j][s], b[p & 1][e], acc[j][i + e]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      load_w(slot, min(tt + WS, last));           // into the slot just consumed
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stray tail prefetches must not outlive the registers / LDS

#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + 16 * i + r16;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + 16 * j + 4 * q;
      if (n >= N) continue;
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
  const unsigned mu = (unsigned)idx / (unsigned)n4;        // (M * N / 4 < 2^31: no 64-bit division)
  const int m = (int)mu, n = (int)((unsigned)idx - mu * (unsigned)n4) * 4;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < ksplit; ++k) s += *reinterpret_cast<const f32x4*>(partial + ((size_t)k * M + m) * N + n);
  typename Vec4<T>::type o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = (T)s[r];
  *reinterpret_cast<typename Vec4<T>::type*>(out + (long)m * out_stride + n) = o;
}

template <typename T, int BM, int FN>
int launch(const void* x, long xs, const void* w, long ws, void* out, long os, void* partial, int M, int N, int K,
           int ksplit, int variant, hipStream_t st) {
  constexpr int BN = MG_WAVES * 16 * FN;
  const int ntn = (N + BN - 1) / BN;
  dim3 grid(ntn * ksplit, (M + BM - 1) / BM);
  if constexpr (BM >= 64 && FN <= 2) {
    if (variant == 1) {
      mid_gemm_deep_kernel<T, BM, FN><<<grid, MG_THREADS, 0, st>>>((const T*)x, xs, (const T*)w, ws, (T*)out, os,
                                                                   (float*)partial, M, N, K, ksplit);
    } else {
      mid_gemm_kernel<T, BM, FN><<<grid, MG_THREADS, 0, st>>>((const T*)x, xs, (const T*)w, ws, (T*)out, os,
                                                              (float*)partial, M, N, K, ksplit);
    }
  } else {
    mid_gemm_kernel<T, BM, FN><<<grid, MG_THREADS, 0, st>>>((const T*)x, xs, (const T*)w, ws, (T*)out, os,
                                                            (float*)partial, M, N, K, ksplit);
  }
  HIP_LAUNCH_CHECK();
  if (ksplit > 1 && out != nullptr) {            // out == nullptr: the caller consumes the K-split slabs itself
    const long n = (long)M * (N / 4);
    mid_reduce_kernel<T><<<(unsigned)((n + 255) / 256), 256, 0, st>>>((T*)out, os, (const float*)partial, M, N,
                                                                     ksplit);
    HIP_LAUNCH_CHECK();
  }
  return 0;
}

template <typename T>
int dispatch(const void* x, long xs, const void* w, long ws, void* out, long os, void* partial, int M, int N, int K,
             int ksplit, int bm, int bn, int variant, hipStream_t st) {
#define MG_CASE(BM_, FN_)                                                                       \
  if (bm == BM_ && bn == MG_WAVES * 16 * FN_)                                                   \
    return launch<T, BM_, FN_>(x, xs, w, ws, out, os, partial, M, N, K, ksplit, variant, st);
  MG_CASE(256, 2) MG_CASE(256, 1) MG_CASE(128, 2) MG_CASE(128, 1) MG_CASE(64, 2) MG_CASE(64, 4)
#undef MG_CASE
  return -4;
}

}  // namespace

// out[M][N] = x[M][K] . w[N][K]^T.  K % 64 == 0, N % 16 == 0; strides in elements, multiples of 8;
// pointers 16-byte aligned.  (bm, bn) is the workgroup tile: (256, 128|256), (128, 128|256), (64, 256|512).
// partial holds ksplit * M * N floats when ksplit > 1.  variant 0: x staged through registers, W two tiles
// ahead; variant 1: x by LDS-DMA, W four tiles ahead (tiles with bn <= 256).
VDAMD_API int vdamd_mid_gemm(void* out, long out_stride, const void* x, long x_stride, const void* w, long w_stride,
                             void* partial, int M, int N, int K, int ksplit, int bm, int bn, int variant, int dtype,
                             void* stream) {
  if (M <= 0) return 0;
  if (K % MG_BK || N % 16 || ksplit < 1 || ksplit > K / MG_BK) return -1;
  if (ksplit > 1 && partial == nullptr) return -2;
  if (out == nullptr && ksplit == 1) return -2;
  if ((x_stride | w_stride) % 8 || out_stride % 4) return -5;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16)
    return dispatch<bf16_t>(x, x_stride, w, w_stride, out, out_stride, partial, M, N, K, ksplit, bm, bn, variant, st);
  if (dtype == DT_F16)
    return dispatch<f16_t>(x, x_stride, w, w_stride, out, out_stride, partial, M, N, K, ksplit, bm, bn, variant, st);
  return -3;
}
