// Weight-only 4-bit (W4A16) linear layers for gfx950.
//
// Weights live in an MFMA-operand-native layout produced once at load time
// (ops/quant.py: pack_w4): for output tile nt (16 rows) and K block kb (128
// columns) the 1 KiB tile qw[nt][kb][lane][4] holds, for lane = c + 16*g, the
// four words of row n = 16*nt + c covering k = 128*kb + 32*g + 8*s + e
// (s = word, e = 0..7).  Element e sits in nibble (e>>1) + 4*(e&1), so
// (w >> 4*i) & 0x000F000F isolates elements (2i, 2i+1) in the two halves of a
// dword and OR-ing the exponent of 128.0 (bf16) / 1024.0 (fp16) turns them into
// a packed pair holding OFF + q exactly - one shift and one and-or per two
// weights, no integer->float conversion.  A wave's 16-byte-per-lane load of a
// tile is one fully coalesced 1 KiB read and is directly the A operand of four
// 16x16x32 MFMAs.
//
// Zero points and scales are applied to the fp32 group sums, not per weight:
//   y[m][n] = sum_g s[n][g] * ( P_g[n][m] - (OFF + z[n][g]) * X_g[m] )
// with P_g = sum_{k in g} (OFF + q[n][k]) x[m][k]  (MFMA) and
//      X_g = sum_{k in g} x[m][k]                  (one MFMA against ones).
// sz[nt][group][16][2] holds (s, OFF + z) per row in the activation dtype.
//
// FMT = 1 is the 4-bit *floating point* form (NVFP4 checkpoints): E2M1 values,
// one OCP e4m3 scale per 16 input columns, one fp32 multiplier per output row.
// The tile layout is the same with element e in nibble e (the checkpoint's own
// byte order), so gfx950's packed fp4 -> f32 conversion reads a word one byte
// at a time.  That conversion only honours the exponent of its scale operand,
// so the e4m3 block scale is a real multiply (v_pk_mul_f32) before the
// round to bf16; lane (c, g) needs exactly two scales per tile - words 0,1 sit
// in block 2g and words 2,3 in block 2g+1 - stored as one dword fs[nt][kb][lane]
// holding the two scales as bf16 (e4m3 -> bf16 is exact), which become fp32
// with a shift and a mask.  (Converting e4m3 bytes in the loop with
// v_cvt_pk_f32_fp8 produced wrong products, non-deterministically, whenever two
// workgroups shared a CU on MI355X; the integer route has no such dependence.)
// Products accumulate straight into the output tile and the per-row multiplier
// is applied in the epilogue.
#include "common.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { W4_INT = 0, W4_FP4 = 1 };

template <typename T>
__device__ __forceinline__ typename Vec8<T>::type fp4_unpack(uint32_t w, float s) {
  typename Vec8<T>::type r;
  const f32x2 a = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 0) * s;
  const f32x2 b = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 1) * s;
  const f32x2 c = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 2) * s;
  const f32x2 d = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 3) * s;
  r[0] = (T)a[0]; r[1] = (T)a[1]; r[2] = (T)b[0]; r[3] = (T)b[1];
  r[4] = (T)c[0]; r[5] = (T)c[1]; r[6] = (T)d[0]; r[7] = (T)d[1];
  return r;
}

// two bf16 block scales in one dword -> fp32
__device__ __forceinline__ float fp4_scale(uint32_t pair, int hi) {
  return __uint_as_float(hi ? (pair & 0xffff0000u) : (pair << 16));
}

template <typename T> struct W4Magic;
template <> struct W4Magic<bf16_t> { static constexpr uint32_t OR = 0x43004300u; static constexpr float OFF = 128.f; };
template <> struct W4Magic<f16_t> { static constexpr uint32_t OR = 0x64006400u; static constexpr float OFF = 1024.f; };

template <typename T>
__device__ __forceinline__ typename Vec8<T>::type w4_unpack(uint32_t w) {
  union { uint32_t u[4]; typename Vec8<T>::type v; } r;
  r.u[0] = (w & 0x000F000Fu) | W4Magic<T>::OR;
  r.u[1] = ((w >> 4) & 0x000F000Fu) | W4Magic<T>::OR;
  r.u[2] = ((w >> 8) & 0x000F000Fu) | W4Magic<T>::OR;
  r.u[3] = ((w >> 12) & 0x000F000Fu) | W4Magic<T>::OR;
  return r.v;
}

constexpr int W4_NTW = 4;          // output tiles (16 rows each) per wave
constexpr int W4_WAVES = 4;        // waves per workgroup; they split the K range

// grid = (ceil(N/64), ksplit, ceil(M / (16*MT))), block = 256.
template <typename T, int MT, int FMT>
__global__ __launch_bounds__(256) void w4a16_gemm_kernel(
    const T* __restrict__ x, long x_stride, const uint32_t* __restrict__ qw, const T* __restrict__ sz,
    T* __restrict__ out, long out_stride, float* __restrict__ partial, int M, int N, int K,
    int kb_per_group, int ksplit, const float* __restrict__ alpha, const int* __restrict__ tiles,
    long qw_group_stride, long sz_group_stride, int up_off = 0) {
  // up_off > 0: x is [M][2K'] = [gate | up] with the up half `up_off` elements further on, and the left
  // operand is silu(gate) * up, formed as the fragments are loaded (no activation kernel in between)
  using V8 = typename Vec8<T>::type;
  constexpr bool PREFETCH_X = MT <= 2;
  int m_base = blockIdx.z * (16 * MT);
  if (tiles != nullptr) {
    // grouped mode, as in w4a16_gemm_lds_kernel: blockIdx.z names a row tile of one group.
    // With a handful of rows per expert this variant spreads each expert over 4x the
    // workgroups of the LDS variant (64 instead of 256 output rows each).
    const int* d = tiles + 3 * blockIdx.z;
    M = d[2];
    if (M == 0) return;
    qw += (size_t)d[0] * qw_group_stride;
    sz += (size_t)d[0] * sz_group_stride;
    if (alpha != nullptr) alpha += (size_t)d[0] * N;
    x += (long)d[1] * x_stride;
    out += (long)d[1] * out_stride;
    m_base = 0;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int KB = K >> 7, ntiles = N >> 4, NG = KB / kb_per_group;
  const int nt0 = blockIdx.x * W4_NTW;
  const int ks = blockIdx.y;
  const int kb_lo = KB * ks / ksplit, kb_hi = KB * (ks + 1) / ksplit;
  const int span = kb_hi - kb_lo;
  const int w_lo = kb_lo + span * wid / W4_WAVES, w_hi = kb_lo + span * (wid + 1) / W4_WAVES;
  const int m0 = m_base;

  // tiles past the end are computed on a clamped index and discarded
  int ntc[W4_NTW];
#pragma unroll
  for (int t = 0; t < W4_NTW; ++t) ntc[t] = min(nt0 + t, ntiles - 1);
  const T* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = min(m0 + 16 * mt + c, M - 1);
    xrow[mt] = x + (long)row * x_stride + 32 * g;
  }

  f32x4 acc[W4_NTW][MT];
#pragma unroll
  for (int t = 0; t < W4_NTW; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  V8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;

  auto load_w = [&](int kb, u32x4 (&w)[W4_NTW], V8 (&s)[W4_NTW], uint32_t (&f)[W4_NTW]) {
#pragma unroll
    for (int t = 0; t < W4_NTW; ++t) {
      w[t] = *reinterpret_cast<const u32x4*>(qw + (((size_t)ntc[t] * KB + kb) << 8) + lane * 4);
      if (FMT == W4_INT)
        s[t] = ld8(sz + ((((size_t)ntc[t] * NG + kb / kb_per_group) * 16 + 4 * g) << 1));
      else
        f[t] = reinterpret_cast<const uint32_t*>(sz)[(((size_t)ntc[t] * KB + kb) << 6) + lane];
    }
  };
  auto load_x = [&](int kb, V8 (&bx)[MT][4]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 4; ++s) bx[mt][s] = ld8(xrow[mt] + (kb << 7) + 8 * s);
    if (up_off > 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const V8 up = ld8(xrow[mt] + up_off + (kb << 7) + 8 * s);
          V8 a;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float gt = (float)bx[mt][s][j];
            a[j] = (T)(fast_silu(gt) * (float)up[j]);
          }
          bx[mt][s] = a;
        }
    }
  };

  u32x4 wc[W4_NTW], wn[W4_NTW];
  V8 sc[W4_NTW], sn[W4_NTW];
  uint32_t fc[W4_NTW], fn[W4_NTW];
  V8 bxc[MT][4], bxn[MT][4];
  if (w_lo < w_hi) {
    load_w(w_lo, wc, sc, fc);
    load_x(w_lo, bxc);
  }
  for (int kb = w_lo; kb < w_hi; ++kb) {
    const bool more = kb + 1 < w_hi;
    if (more) {
      load_w(kb + 1, wn, sn, fn);
      if (PREFETCH_X) load_x(kb + 1, bxn);
    }
    f32x4 xs[MT];
    if (FMT == W4_INT) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        xs[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) xs[mt] = mfma16(ones, bxc[mt][s], xs[mt]);
      }
    }
#pragma unroll
    for (int t = 0; t < W4_NTW; ++t) {
      if (FMT == W4_FP4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const V8 a = fp4_unpack<T>(wc[t][s], fp4_scale(fc[t], s >> 1));
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[t][mt] = mfma16(a, bxc[mt][s], acc[t][mt]);
        }
        continue;
      }
      f32x4 p[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) p[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const V8 a = w4_unpack<T>(wc[t][s]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) p[mt] = mfma16(a, bxc[mt][s], p[mt]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s_r = (float)sc[t][2 * r], z_r = (float)sc[t][2 * r + 1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt][r] += s_r * (p[mt][r] - z_r * xs[mt][r]);
      }
    }
    if (more) {
#pragma unroll
      for (int t = 0; t < W4_NTW; ++t) { wc[t] = wn[t]; sc[t] = sn[t]; fc[t] = fn[t]; }
      if (PREFETCH_X) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int s = 0; s < 4; ++s) bxc[mt][s] = bxn[mt][s];
      } else {
        load_x(kb + 1, bxc);
      }
    }
  }

  // cross-wave reduction: wave w ends up owning tile w
  __shared__ float red[W4_WAVES - 1][W4_NTW][MT * 4][64];
#pragma unroll
  for (int t = 0; t < W4_NTW; ++t) {
    if (t == wid) continue;
    const int slot = ((wid - t + W4_WAVES) % W4_WAVES) - 1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[slot][t][mt * 4 + r][lane] = acc[t][mt][r];
  }
  __syncthreads();
  f32x4 res[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    res[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < W4_NTW; ++t)
      if (t == wid) res[mt] = acc[t][mt];
#pragma unroll
    for (int slot = 0; slot < W4_WAVES - 1; ++slot)
#pragma unroll
      for (int r = 0; r < 4; ++r) res[mt][r] += red[slot][wid][mt * 4 + r][lane];
  }
  const int nt = nt0 + wid;
  if (nt >= ntiles) return;
  const int n = 16 * nt + 4 * g;
  if (FMT == W4_FP4 && alpha != nullptr) {
    const f32x4 al = *reinterpret_cast<const f32x4*>(alpha + n);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) res[mt] *= al;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + c;
    if (m >= M) continue;
    if (ksplit == 1) {
      typename Vec4<T>::type o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (T)res[mt][r];
      *reinterpret_cast<typename Vec4<T>::type*>(out + (long)m * out_stride + n) = o;
    } else {
      *reinterpret_cast<f32x4*>(partial + ((size_t)ks * M + m) * N + n) = res[mt];
    }
  }
}

// Batches above one MFMA tile of rows: the four waves of a workgroup take
// *different* output tiles (16 tiles = 256 rows per workgroup) and share each
// 128-column block of x through LDS, so x crosses L2 once per 16 KiB of
// weights instead of once per wave.  The 16-byte slots of each 256-byte LDS row
// are XOR-swizzled with the row index so that the 16-lane groups a
// ds_read_b128 is served in ({0-3,12-15,20-27}, ...) hit distinct banks.
// grid = (ceil(N/256), ksplit, ceil(M / (16*MT))), block = 256.
constexpr int W4_XPAD = 128;
// Row r XORs its slots with r after swapping bits 0 and 2: the two lane sets a
// ds_read_b128 group mixes ({0-3,12-15} with g, {4-11} with g+1) read slots
// that differ by 4 (slot = 4g + s), and rows whose swizzles differ by exactly 4
// then always belong to the same set.
__device__ __forceinline__ int w4_swz(int r) { return (r & 10) | ((r & 1) << 2) | ((r >> 2) & 1); }

// NTW: output tiles per wave.  Up to 64 rows a wave keeps 4 tiles x MT <= 4 row
// tiles; above that it keeps 2 tiles x MT <= 8, so every decoded 4-bit tile feeds
// twice as many MFMAs (the decode, not HBM, bounds these kernels at that size).
template <typename T, int MT, int FMT, int NTW>
__global__ __launch_bounds__(256) void w4a16_gemm_lds_kernel(
    const T* __restrict__ x, long x_stride, const uint32_t* __restrict__ qw, const T* __restrict__ sz,
    T* __restrict__ out, long out_stride, float* __restrict__ partial, int M, int N, int K,
    int kb_per_group, int ksplit, const int* __restrict__ tiles, long qw_group_stride, long sz_group_stride,
    const float* __restrict__ alpha) {
  using V8 = typename Vec8<T>::type;
  __shared__ __attribute__((aligned(16))) T xl[2][MT * 16][W4_XPAD];
  int m0 = blockIdx.z * (16 * MT);
  if (tiles != nullptr) {
    // grouped mode (mixture-of-experts): blockIdx.z names a row tile
    // (group, first row, rows) of the group-sorted x; rows == 0 marks padding
    const int* d = tiles + 3 * blockIdx.z;
    M = d[2];
    if (M == 0) return;
    qw += (size_t)d[0] * qw_group_stride;
    sz += (size_t)d[0] * sz_group_stride;      // in elements of T (FMT 1: two per scale dword)
    if (alpha != nullptr) alpha += (size_t)d[0] * N;
    x += (long)d[1] * x_stride;
    out += (long)d[1] * out_stride;
    m0 = 0;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int KB = K >> 7, ntiles = N >> 4, NG = KB / kb_per_group;
  const int nt0 = (blockIdx.x * W4_WAVES + wid) * NTW;
  const int ks = blockIdx.y;
  const int kb_lo = KB * ks / ksplit, kb_hi = KB * (ks + 1) / ksplit;

  int ntc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) ntc[t] = min(nt0 + t, ntiles - 1);
  // cooperative copy of x: thread -> (row xr of a 16-row pass, 16-byte column xc)
  const int xr = threadIdx.x >> 4, xc = threadIdx.x & 15;
  const T* xsrc[MT];
#pragma unroll
  for (int p = 0; p < MT; ++p) xsrc[p] = x + (long)min(m0 + 16 * p + xr, M - 1) * x_stride + 8 * xc;

  f32x4 acc[NTW][MT];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  V8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;

  auto load_w = [&](int kb, u32x4 (&w)[NTW], V8 (&s)[NTW], uint32_t (&f)[NTW]) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      w[t] = *reinterpret_cast<const u32x4*>(qw + (((size_t)ntc[t] * KB + kb) << 8) + lane * 4);
      if (FMT == W4_INT)
        s[t] = ld8(sz + ((((size_t)ntc[t] * NG + kb / kb_per_group) * 16 + 4 * g) << 1));
      else
        f[t] = reinterpret_cast<const uint32_t*>(sz)[(((size_t)ntc[t] * KB + kb) << 6) + lane];
    }
  };
  u32x4 wc[NTW], wn[NTW];
  V8 sc[NTW], sn[NTW];
  uint32_t fc[NTW], fn[NTW];
  V8 xreg[MT];
  if (kb_lo < kb_hi) {
    load_w(kb_lo, wc, sc, fc);
#pragma unroll
    for (int p = 0; p < MT; ++p) st8(&xl[0][16 * p + xr][8 * (xc ^ w4_swz(xr))], ld8(xsrc[p] + (kb_lo << 7)));
  }
  __syncthreads();
  for (int kb = kb_lo; kb < kb_hi; ++kb) {
    const int buf = (kb - kb_lo) & 1;
    const bool more = kb + 1 < kb_hi;
    if (more) {
      load_w(kb + 1, wn, sn, fn);
#pragma unroll
      for (int p = 0; p < MT; ++p) xreg[p] = ld8(xsrc[p] + ((kb + 1) << 7));
    }
    if constexpr (MT <= 4) {
      V8 bx[MT][4];
  #pragma unroll
      for (int mt = 0; mt < MT; ++mt)
  #pragma unroll
        for (int s = 0; s < 4; ++s) bx[mt][s] = ld8(&xl[buf][16 * mt + c][8 * ((4 * g + s) ^ w4_swz(c))]);
      f32x4 xs[MT];
      if (FMT == W4_INT) {
  #pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          xs[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  #pragma unroll
          for (int s = 0; s < 4; ++s) xs[mt] = mfma16(ones, bx[mt][s], xs[mt]);
        }
      }
  #pragma unroll
      for (int t = 0; t < NTW; ++t) {
        if (FMT == W4_FP4) {
  #pragma unroll
          for (int s = 0; s < 4; ++s) {
            const V8 a = fp4_unpack<T>(wc[t][s], fp4_scale(fc[t], s >> 1));
  #pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] = mfma16(a, bx[mt][s], acc[t][mt]);
          }
          continue;
        }
        f32x4 p[MT];
  #pragma unroll
        for (int mt = 0; mt < MT; ++mt) p[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  #pragma unroll
        for (int s = 0; s < 4; ++s) {
          const V8 a = w4_unpack<T>(wc[t][s]);
  #pragma unroll
          for (int mt = 0; mt < MT; ++mt) p[mt] = mfma16(a, bx[mt][s], p[mt]);
        }
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s_r = (float)sc[t][2 * r], z_r = (float)sc[t][2 * r + 1];
  #pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[t][mt][r] += s_r * (p[mt][r] - z_r * xs[mt][r]);
        }
      }
    } else {
      // many row tiles: walk the four 32-column MFMA steps outermost so only one
      // step's x fragments are live
      f32x4 xs[MT], p[NTW][MT];
      if (FMT == W4_INT) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          xs[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < NTW; ++t) p[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        V8 bxs[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) bxs[mt] = ld8(&xl[buf][16 * mt + c][8 * ((4 * g + s) ^ w4_swz(c))]);
        if (FMT == W4_INT) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) xs[mt] = mfma16(ones, bxs[mt], xs[mt]);
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          if (FMT == W4_FP4) {
            const V8 a = fp4_unpack<T>(wc[t][s], fp4_scale(fc[t], s >> 1));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] = mfma16(a, bxs[mt], acc[t][mt]);
          } else {
            const V8 a = w4_unpack<T>(wc[t][s]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) p[t][mt] = mfma16(a, bxs[mt], p[t][mt]);
          }
        }
      }
      if (FMT == W4_INT) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float s_r = (float)sc[t][2 * r], z_r = (float)sc[t][2 * r + 1];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt][r] += s_r * (p[t][mt][r] - z_r * xs[mt][r]);
          }
      }
    }
    if (more) {
#pragma unroll
      for (int t = 0; t < NTW; ++t) { wc[t] = wn[t]; sc[t] = sn[t]; fc[t] = fn[t]; }
#pragma unroll
      for (int p = 0; p < MT; ++p) st8(&xl[buf ^ 1][16 * p + xr][8 * (xc ^ w4_swz(xr))], xreg[p]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    if (nt0 + t >= ntiles) continue;
    const int n = 16 * (nt0 + t) + 4 * g;
    if (FMT == W4_FP4 && alpha != nullptr) {
      const f32x4 al = *reinterpret_cast<const f32x4*>(alpha + n);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] *= al;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + 16 * mt + c;
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

// out[m][n] = sum_ks partial[ks][m][n]; one thread per 4 outputs
template <typename T>
__global__ void w4a16_reduce_kernel(T* __restrict__ out, long out_stride, const float* __restrict__ partial,
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

// native 4-bit -> dense [N][K]; one workgroup per 16x128 tile, thread = (row, word)
template <typename T>
__global__ __launch_bounds__(256) void w4_dequant_kernel(T* __restrict__ out, const uint32_t* __restrict__ qw,
                                                         const T* __restrict__ sz, int N, int K,
                                                         int kb_per_group) {
  const int KB = K >> 7, NG = KB / kb_per_group;
  const int tile = blockIdx.x;
  const int nt = tile / KB, kb = tile % KB;
  const int kk = threadIdx.x & 15, row = threadIdx.x >> 4;      // word along K, row in tile
  const int lane = row + 16 * (kk >> 2), word = kk & 3;
  const uint32_t w = qw[((size_t)tile << 8) + lane * 4 + word];
  const T* szp = sz + ((((size_t)nt * NG + kb / kb_per_group) * 16 + row) << 1);
  const float s = (float)szp[0], z = (float)szp[1] - W4Magic<T>::OFF;
  typename Vec8<T>::type o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = (T)(((float)((w >> (4 * i)) & 0xF) - z) * s);
    o[2 * i + 1] = (T)(((float)((w >> (16 + 4 * i)) & 0xF) - z) * s);
  }
  st8(out + (size_t)(16 * nt + row) * K + (kb << 7) + 8 * kk, o);
}

template <typename T, int FMT>
static int launch_w4a16(void* out, long out_stride, const void* x, long x_stride, const void* qw,
                        const void* sz, const float* alpha, void* partial, int M, int N, int K, int G,
                        int ksplit, int variant, hipStream_t st) {
  // variant 1 above 64 rows: chunks of <= 128 rows, 2 output tiles per wave
  const bool wide = variant == 1 && M > 64;
  const int chunks = wide ? (M + 127) / 128 : (M + 63) / 64;
  const int mt = wide ? ((M + chunks - 1) / chunks + 15) / 16 : min(4, (M + 15) / 16);
  const int ntw = wide ? 2 : W4_NTW;
  const int tiles_per_wg = variant == 1 ? ntw * W4_WAVES : W4_NTW;
  dim3 grid((N / 16 + tiles_per_wg - 1) / tiles_per_wg, ksplit, (M + 16 * mt - 1) / (16 * mt));
#define W4_LDS(MT_, NTW_)                                                                                    \
  w4a16_gemm_lds_kernel<T, MT_, FMT, NTW_><<<grid, 256, 0, st>>>((const T*)x, x_stride, (const uint32_t*)qw, \
                                                               (const T*)sz, (T*)out, out_stride,           \
                                                               (float*)partial, M, N, K, G / 128, ksplit,   \
                                                               nullptr, 0, 0, alpha)
#define W4_LAUNCH(MT_)                                                                                      \
  if (variant == 1)                                                                                         \
    W4_LDS(MT_, W4_NTW);                                                                                    \
  else                                                                                                      \
    w4a16_gemm_kernel<T, MT_, FMT><<<grid, 256, 0, st>>>((const T*)x, x_stride, (const uint32_t*)qw,       \
                                                      (const T*)sz, (T*)out, out_stride, (float*)partial,  \
                                                      M, N, K, G / 128, ksplit, alpha, nullptr, 0, 0)
  if (wide) {
    switch (mt) {
      case 3: W4_LDS(3, 2); break;      // 65..96 rows in two chunks cannot happen; kept for completeness
      case 4: W4_LDS(4, 2); break;
      case 5: W4_LDS(5, 2); break;
      case 6: W4_LDS(6, 2); break;
      case 7: W4_LDS(7, 2); break;
      default: W4_LDS(8, 2); break;
    }
  } else {
    switch (mt) {
      case 1: W4_LAUNCH(1); break;
      case 2: W4_LAUNCH(2); break;
      case 3: W4_LAUNCH(3); break;
      default: W4_LAUNCH(4); break;
    }
  }
#undef W4_LAUNCH
#undef W4_LDS
  HIP_LAUNCH_CHECK();
  if (ksplit > 1 && out != nullptr) {            // out == nullptr: the caller consumes the K-split slabs itself
    const long n = (long)M * (N / 4);
    w4a16_reduce_kernel<T><<<(unsigned)((n + 255) / 256), 256, 0, st>>>((T*)out, out_stride,
                                                                        (const float*)partial, M, N, ksplit);
    HIP_LAUNCH_CHECK();
  }
  return 0;
}

// fp4 tiles -> dense [N][K]; same thread mapping as w4_dequant_kernel
template <typename T>
__global__ __launch_bounds__(256) void fp4_dequant_kernel(T* __restrict__ out, const uint32_t* __restrict__ qw,
                                                          const uint32_t* __restrict__ fs,
                                                          const float* __restrict__ alpha, int N, int K) {
  const int KB = K >> 7;
  const int tile = blockIdx.x;
  const int nt = tile / KB, kb = tile % KB;
  const int kk = threadIdx.x & 15, row = threadIdx.x >> 4;
  const int lane = row + 16 * (kk >> 2), word = kk & 3;
  const uint32_t w = qw[((size_t)tile << 8) + lane * 4 + word];
  float s = fp4_scale(fs[((size_t)tile << 6) + lane], word >> 1);
  if (alpha != nullptr) s *= alpha[16 * nt + row];
  st8(out + (size_t)(16 * nt + row) * K + (kb << 7) + 8 * kk, fp4_unpack<T>(w, s));
}

// y[M][N] = x[M][K] . dequant(qw, sz)^T.  Requires K % 128 == 0, N % 16 == 0,
// G % 128 == 0, K % G == 0, 1 <= ksplit <= K/128; `partial` holds ksplit*M*N
// floats when ksplit > 1.  variant 0: waves split K (one 64-row tile group per
// workgroup, x straight from L2); variant 1: waves split N and share x in LDS.
VDAMD_API int vdamd_w4a16_gemm(void* out, long out_stride, const void* x, long x_stride, const void* qw,
                               const void* sz, void* partial, int M, int N, int K, int G, int ksplit,
                               int variant, int dtype, void* stream) {
  if (M <= 0) return 0;
  if (K % 128 || N % 16 || G % 128 || K % G || ksplit < 1 || ksplit > K / 128) return -1;
  if (ksplit > 1 && partial == nullptr) return -2;
  if (variant != 0 && variant != 1) return -4;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) return launch_w4a16<bf16_t, W4_INT>(out, out_stride, x, x_stride, qw, sz, nullptr, partial, M, N, K, G, ksplit, variant, st);
  if (dtype == DT_F16) return launch_w4a16<f16_t, W4_INT>(out, out_stride, x, x_stride, qw, sz, nullptr, partial, M, N, K, G, ksplit, variant, st);
  return -3;
}

// NVFP4 weights: qw as above with element e in nibble e, fs[N/16][K/128][64]
// dwords of two bf16 block scales, alpha[N] fp32 per-row multipliers (may be null).
VDAMD_API int vdamd_fp4a16_gemm(void* out, long out_stride, const void* x, long x_stride, const void* qw,
                                const void* fs, const float* alpha, void* partial, int M, int N, int K,
                                int ksplit, int variant, int dtype, void* stream) {
  if (M <= 0) return 0;
  if (K % 128 || N % 16 || ksplit < 1 || ksplit > K / 128) return -1;
  if (ksplit > 1 && partial == nullptr) return -2;
  if (variant != 0 && variant != 1) return -4;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) return launch_w4a16<bf16_t, W4_FP4>(out, out_stride, x, x_stride, qw, fs, alpha, partial, M, N, K, 128, ksplit, variant, st);
  if (dtype == DT_F16) return launch_w4a16<f16_t, W4_FP4>(out, out_stride, x, x_stride, qw, fs, alpha, partial, M, N, K, 128, ksplit, variant, st);
  return -3;
}

VDAMD_API int vdamd_fp4_dequant(void* out, const void* qw, const void* fs, const float* alpha, int N, int K,
                                int dtype, void* stream) {
  if (K % 128 || N % 16) return -1;
  hipStream_t st = (hipStream_t)stream;
  const unsigned tiles = (unsigned)((N / 16) * (K / 128));
  if (dtype == DT_BF16)
    fp4_dequant_kernel<bf16_t><<<tiles, 256, 0, st>>>((bf16_t*)out, (const uint32_t*)qw, (const uint32_t*)fs, alpha, N, K);
  else if (dtype == DT_F16)
    fp4_dequant_kernel<f16_t><<<tiles, 256, 0, st>>>((f16_t*)out, (const uint32_t*)qw, (const uint32_t*)fs, alpha, N, K);
  else
    return -3;
  HIP_LAUNCH_CHECK();
  return 0;
}

VDAMD_API int vdamd_w4_dequant(void* out, const void* qw, const void* sz, int N, int K, int G, int dtype,
                               void* stream) {
  if (K % 128 || N % 16 || G % 128 || K % G) return -1;
  hipStream_t st = (hipStream_t)stream;
  const unsigned tiles = (unsigned)((N / 16) * (K / 128));
  if (dtype == DT_BF16)
    w4_dequant_kernel<bf16_t><<<tiles, 256, 0, st>>>((bf16_t*)out, (const uint32_t*)qw, (const bf16_t*)sz, N, K, G / 128);
  else if (dtype == DT_F16)
    w4_dequant_kernel<f16_t><<<tiles, 256, 0, st>>>((f16_t*)out, (const uint32_t*)qw, (const f16_t*)sz, N, K, G / 128);
  else
    return -3;
  HIP_LAUNCH_CHECK();
  return 0;
}

template <typename T, int FMT>
static int launch_w4_grouped(void* out, long os, const void* x, long xs, const void* qw, const void* sz, long qgs,
                             long sgs, const float* alpha, const int* tiles, int num_tiles, int tile_rows, int N,
                             int K, int G, hipStream_t st, int up_off = 0) {
  if (up_off > 0 && tile_rows > 32) return -6;      // only the register-operand variant forms the activation
  if (tile_rows <= 32) {
    // few rows per group: 64 output rows per workgroup, the four waves split K
    dim3 grid((N / 16 + W4_NTW - 1) / W4_NTW, 1, num_tiles);
#define W4G0(MT_)                                                                                               \
  w4a16_gemm_kernel<T, MT_, FMT><<<grid, 256, 0, st>>>((const T*)x, xs, (const uint32_t*)qw, (const T*)sz, (T*)out, \
                                                    os, nullptr, tile_rows, N, K, G / 128, 1, alpha, tiles, qgs, sgs, \
                                                    up_off)
    if (tile_rows <= 16) W4G0(1); else W4G0(2);
#undef W4G0
    HIP_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((N / 16 + W4_NTW * W4_WAVES - 1) / (W4_NTW * W4_WAVES), 1, num_tiles);
#define W4G(MT_)                                                                                                \
  w4a16_gemm_lds_kernel<T, MT_, FMT, W4_NTW><<<grid, 256, 0, st>>>((const T*)x, xs, (const uint32_t*)qw, (const T*)sz, \
                                                          (T*)out, os, nullptr, tile_rows, N, K, G / 128, 1,   \
                                                          tiles, qgs, sgs, alpha)
  switch (tile_rows / 16) {
    case 3: W4G(3); break;
    default: W4G(4); break;
  }
#undef W4G
  HIP_LAUNCH_CHECK();
  return 0;
}

// Grouped form for mixture-of-experts (see vdamd_skinny_gemm_grouped): group g's
// packed weights start at qw + g * qw_group_stride words, sz + g * sz_group_stride
// elements; tile_rows is 16, 32, 48 or 64.
// up_off > 0 (tile_rows <= 32 only): x rows are [gate | up] with the up half up_off elements after the
// gate half, and the product is taken with silu(gate) * up.
VDAMD_API int vdamd_w4a16_gemm_grouped(void* out, long out_stride, const void* x, long x_stride, const void* qw,
                                       const void* sz, long qw_group_stride, long sz_group_stride,
                                       const int* tiles, int num_tiles, int tile_rows, int N, int K, int G,
                                       int dtype, int up_off, void* stream) {
  if (num_tiles <= 0) return 0;
  if (tile_rows <= 0 || tile_rows > 64 || tile_rows % 16 || tiles == nullptr) return -1;
  if (K % 128 || N % 16 || G % 128 || K % G || up_off < 0 || up_off % 8) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16)
    return launch_w4_grouped<bf16_t, W4_INT>(out, out_stride, x, x_stride, qw, sz, qw_group_stride, sz_group_stride,
                                             nullptr, tiles, num_tiles, tile_rows, N, K, G, st, up_off);
  if (dtype == DT_F16)
    return launch_w4_grouped<f16_t, W4_INT>(out, out_stride, x, x_stride, qw, sz, qw_group_stride, sz_group_stride,
                                            nullptr, tiles, num_tiles, tile_rows, N, K, G, st, up_off);
  return -3;
}

// fs_group_stride counts 16-bit halves (2 * N/16 * K/128 * 64 per group);
// alpha is [groups][N] or null.
VDAMD_API int vdamd_fp4a16_gemm_grouped(void* out, long out_stride, const void* x, long x_stride, const void* qw,
                                        const void* fs, long qw_group_stride, long fs_group_stride,
                                        const float* alpha, const int* tiles, int num_tiles, int tile_rows, int N,
                                        int K, int dtype, void* stream) {
  if (num_tiles <= 0) return 0;
  if (tile_rows <= 0 || tile_rows > 64 || tile_rows % 16 || tiles == nullptr) return -1;
  if (K % 128 || N % 16) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16)
    return launch_w4_grouped<bf16_t, W4_FP4>(out, out_stride, x, x_stride, qw, fs, qw_group_stride, fs_group_stride,
                                             alpha, tiles, num_tiles, tile_rows, N, K, 128, st);
  if (dtype == DT_F16)
    return launch_w4_grouped<f16_t, W4_FP4>(out, out_stride, x, x_stride, qw, fs, qw_group_stride, fs_group_stride,
                                            alpha, tiles, num_tiles, tile_rows, N, K, 128, st);
  return -3;
}
