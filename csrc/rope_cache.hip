// Rotary embedding (NeoX pairing) fused with the paged KV-cache write.
//
// Cache layouts are MFMA-operand-native (see attention.hip):
//   k_cache[num_blocks][Hkv][block_size][D]   key row contiguous in D
//   v_cache[num_blocks][Hkv][D/16][block_size/8][16][8]   V^T in MFMA-fragment order
//
// Each thread owns one 8-element (16-byte) piece of one token; a token has
//   items [0, Hq*D/16)            rotate a q pair-chunk in place
//   items [.., +Hkv*D/16)         rotate a k pair-chunk, write it to k_cache
//   items [.., +Hkv*D/8)          copy 8 v elements (head-dim index c, c + D/8, ...) to v_cache:
//                                 element-sized stores 16 bytes apart, so the lanes of a head fill
//                                 8 element slots of each 128-byte line per store instruction and a
//                                 token dirties D/16 runs of 256 bytes per head
// and (token, item) pairs are laid out flat over the grid, so a rank that holds
// one KV head (TP=8: 88 items per token) still fills 256-thread workgroups.
//
// Optional per-head RMSNorm of q and k before the rotation (Qwen3): the D/16
// threads that share a head are consecutive, aligned lanes, so the sum of
// squares is a few lane shuffles and the normalised values never leave fp32.
#include "common.h"

namespace {

// 8 floats (already divided by the cache scale) -> 8 OCP e4m3 bytes, saturating at +-448
__device__ __forceinline__ uint2 pack_fp8x8(const float (&f)[8]) {
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = fminf(fmaxf(f[j], -448.f), 448.f);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return make_uint2((unsigned)lo, (unsigned)hi);
}

// sum over the slabs of 8 consecutive fp32 partial sums, rounded to T as the GEMM's own reduction would
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type sum8(const float* __restrict__ p, int nslabs, long slab_stride) {
  f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
  int s = 0;
  for (; s + 4 <= nslabs; s += 4) {               // four slabs' loads in flight at a time
    f32x4 la[4], lb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      la[u] = *reinterpret_cast<const f32x4*>(p + (s + u) * slab_stride);
      lb[u] = *reinterpret_cast<const f32x4*>(p + (s + u) * slab_stride + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { a += la[u]; b += lb[u]; }
  }
  for (; s < nslabs; ++s) {
    a += *reinterpret_cast<const f32x4*>(p + s * slab_stride);
    b += *reinterpret_cast<const f32x4*>(p + s * slab_stride + 4);
  }
  typename Vec8<T>::type v;
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[j] = (T)a[j]; v[j + 4] = (T)b[j]; }
  return v;
}

// KV8: the cache holds one e4m3 byte per element (same layouts); values are divided by the
// KV head's scale before they are narrowed.
// SLABS (with WRITE_CACHE and ROTATE): q, k and v arrive as the fp32 K-split partial sums
// part[nslabs][T][(Hq + 2 Hkv) D] of the projection that produced them; the rotated q goes to `q`
// (a buffer of its own), k and v only to the cache.
template <typename T, bool WRITE_CACHE, bool ROTATE, bool KV8, bool SLABS = false>
__global__ void rope_cache_kernel(const int* __restrict__ positions, T* __restrict__ q, T* __restrict__ k,
                                  const T* __restrict__ v, long q_stride, long k_stride, long v_stride,
                                  const float* __restrict__ cos_sin, void* __restrict__ k_cache_v,
                                  void* __restrict__ v_cache_v, const int* __restrict__ slot_mapping, int Hq, int Hkv,
                                  int D, int block_size, int T_tokens, const T* __restrict__ q_norm_w,
                                  const T* __restrict__ k_norm_w, float norm_eps,
                                  const float* __restrict__ k_scales, const float* __restrict__ v_scales,
                                  const float* __restrict__ part = nullptr, int nslabs = 0, long slab_stride = 0) {
  T* k_cache = reinterpret_cast<T*>(k_cache_v);
  T* v_cache = reinterpret_cast<T*>(v_cache_v);
  using V = typename Vec8<T>::type;
  const int half = D >> 1;
  const int cpp = half >> 3;                 // 8-element chunks per half head
  const int q_items = Hq * cpp, k_items = Hkv * cpp, v_items = WRITE_CACHE ? Hkv * (D >> 3) : 0;
  const int items = q_items + k_items + v_items;
  // (32-bit index math: the launcher refuses T * items >= 2^31, and a 64-bit division is a ~150-instruction
  // routine, three of which used to open every thread of a kernel that moves 48 bytes per thread)
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned tu = gid / (unsigned)items;
  if (tu >= (unsigned)T_tokens) return;
  const long t = tu;
  const int item = (int)(gid - tu * (unsigned)items);
  const int pos = ROTATE ? positions[t] : 0;
  const float* cs = cos_sin + (long)pos * D;   // [cos(half) | sin(half)]
  int slot = -1;
  if (WRITE_CACHE) slot = slot_mapping[t];
  const unsigned blk_u = slot >= 0 ? (unsigned)slot / (unsigned)block_size : 0u;
  const long blk = blk_u;
  const int off = slot >= 0 ? (int)((unsigned)slot - blk_u * (unsigned)block_size) : 0;

  {
    if (item < q_items + k_items) {
      const bool is_q = item < q_items;
      const int it = is_q ? item : item - q_items;
      const int h = it / cpp, c = it - h * cpp;
      T* base = is_q ? q + t * q_stride + (long)h * D : k + t * k_stride + (long)h * D;
      V x1, x2;
      if constexpr (SLABS) {
        const float* pb = part + t * (long)(Hq + 2 * Hkv) * D + (long)(is_q ? h : Hq + h) * D;
        x1 = sum8<T>(pb + 8 * c, nslabs, slab_stride);
        x2 = sum8<T>(pb + half + 8 * c, nslabs, slab_stride);
      } else {
        x1 = ld8<T>(base + 8 * c);
        x2 = ld8<T>(base + half + 8 * c);
      }
      V o1 = x1, o2 = x2;
      float f1[8], f2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { f1[j] = (float)x1[j]; f2[j] = (float)x2[j]; }
      const T* nw = is_q ? q_norm_w : k_norm_w;
      if (ROTATE && nw != nullptr) {
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f1[j] * f1[j] + f2[j] * f2[j];
        for (int o = 1; o < cpp; o <<= 1) ss += __shfl_xor(ss, o, 64);      // the head's cpp lanes are aligned
        const float r = rsqrtf(ss / (float)D + norm_eps);
        const V w1 = ld8<T>(nw + 8 * c), w2 = ld8<T>(nw + half + 8 * c);
#pragma unroll
        for (int j = 0; j < 8; ++j) { f1[j] *= r * (float)w1[j]; f2[j] *= r * (float)w2[j]; }
      }
      if (ROTATE) {
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + 8 * c);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(cs + 8 * c + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(cs + half + 8 * c);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(cs + half + 8 * c + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float cj = j < 4 ? c0[j & 3] : c1[j & 3];
          const float sj = j < 4 ? s0[j & 3] : s1[j & 3];
          const float a = f1[j], b = f2[j];
          o1[j] = (T)(a * cj - b * sj);
          o2[j] = (T)(b * cj + a * sj);
        }
      }
      if (is_q) {
        if (ROTATE) { st8<T>(base + 8 * c, o1); st8<T>(base + half + 8 * c, o2); }
      } else {
        if (!WRITE_CACHE && ROTATE) { st8<T>(base + 8 * c, o1); st8<T>(base + half + 8 * c, o2); }
        if (WRITE_CACHE && slot >= 0) {
          const long row = ((blk * Hkv + h) * block_size + off) * (long)D;
          if constexpr (KV8) {
            const float inv = (k_scales != nullptr) ? 1.f / k_scales[h] : 1.f;
            float g1[8], g2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { g1[j] = (float)o1[j] * inv; g2[j] = (float)o2[j] * inv; }
            uint8_t* dst = reinterpret_cast<uint8_t*>(k_cache_v) + row;
            *reinterpret_cast<uint2*>(dst + 8 * c) = pack_fp8x8(g1);
            *reinterpret_cast<uint2*>(dst + half + 8 * c) = pack_fp8x8(g2);
          } else {
            T* dst = k_cache + row;
            st8<T>(dst + 8 * c, o1);
            st8<T>(dst + half + 8 * c, o2);
          }
        }
      }
    } else if (slot >= 0) {
      const int it = item - q_items - k_items;
      const int cpd = D >> 3;
      const int h = it / cpd, c = it - h * cpd;
      // this thread's head-dim indices are c + cpd*j: the head's cpd lanes store adjacent 16-byte groups
      // (from slabs: 8c + j, so that the partial sums are read as two 16-byte vectors per slab)
      T x[8];
      const int d0 = SLABS ? 8 * c : c, dstep = SLABS ? 1 : cpd;
      if constexpr (SLABS) {
        const V xv = sum8<T>(part + t * (long)(Hq + 2 * Hkv) * D + (long)(Hq + Hkv + h) * D + 8 * c, nslabs,
                             slab_stride);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = xv[j];
      } else {
        const T* src = v + t * v_stride + (long)h * D + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = src[j * cpd];
      }
      // element index of head-dim row d: 16-row slab d/16, key group off/8, row d%16, key off%8
      const long head0 = (blk * Hkv + h) * (long)D * block_size + (off >> 3) * 128 + (off & 7);
      auto at = [&](int d) { return head0 + (long)(d >> 4) * (block_size * 16) + (d & 15) * 8; };
      if constexpr (KV8) {
        const float inv = (v_scales != nullptr) ? 1.f / v_scales[h] : 1.f;
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = (float)x[j] * inv;
        const uint2 pk = pack_fp8x8(g);
        uint8_t* dst = reinterpret_cast<uint8_t*>(v_cache_v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          dst[at(d0 + j * dstep)] = (uint8_t)(((j < 4 ? pk.x : pk.y) >> (8 * (j & 3))) & 0xFF);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v_cache[at(d0 + j * dstep)] = x[j];
      }
    }
  }
}

template <typename T>
int launch(const int* positions, void* q, void* k, const void* v, long qs, long ks, long vs, const float* cos_sin,
           void* k_cache, void* v_cache, const int* slot_mapping, int T_tokens, int Hq, int Hkv, int D,
           int block_size, int mode, const void* qnw, const void* knw, float eps, int kv_fp8,
           const float* k_scales, const float* v_scales, hipStream_t s, const float* part = nullptr,
           int nslabs = 0) {
  // mode 0 and 2 write the cache (v items exist), mode 1 rotates only
  const long items = Hq * (D >> 4) + Hkv * (D >> 4) + (mode == 1 ? 0 : Hkv * (D >> 3));
  const long total = items * T_tokens;
  if (total + 256 >= (1L << 31)) return -3;
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
#define L(WC, ROT, K8)                                                                                          \
  hipLaunchKernelGGL((rope_cache_kernel<T, WC, ROT, K8>), dim3(blocks), dim3(threads), 0, s, positions, (T*)q,  \
                     (T*)k, (const T*)v, qs, ks, vs, cos_sin, k_cache, v_cache, slot_mapping, Hq, Hkv,          \
                     D, block_size, T_tokens, (const T*)qnw, (const T*)knw, eps, k_scales, v_scales)
  if (part != nullptr) {                 // rotate + cache from K-split partial sums
    const long slab_stride = (long)T_tokens * (Hq + 2 * Hkv) * D;
#define LS(K8)                                                                                                  \
  hipLaunchKernelGGL((rope_cache_kernel<T, true, true, K8, true>), dim3(blocks), dim3(threads), 0, s, positions, \
                     (T*)q, (T*)nullptr, (const T*)nullptr, qs, 0L, 0L, cos_sin, k_cache, v_cache, slot_mapping, \
                     Hq, Hkv, D, block_size, T_tokens, (const T*)qnw, (const T*)knw, eps, k_scales, v_scales,    \
                     part, nslabs, slab_stride)
    if (kv_fp8) LS(true); else LS(false);
#undef LS
  } else if (mode == 1) L(false, true, false);  // rotate q,k in place only
  else if (mode == 0) {                  // rotate q,k + write k,v to cache
    if (kv_fp8) L(true, true, true); else L(true, true, false);
  } else {                               // cache write only (k,v already final)
    if (kv_fp8) L(true, false, true); else L(true, false, false);
  }
#undef L
  HIP_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// mode 0: rotate + cache, 1: rotate only, 2: cache only.  Strides are in elements.
// q_norm_w / k_norm_w (D elements each, both or neither): RMSNorm each head before rotating.
VDAMD_API int vdamd_rope_cache(const int* positions, void* q, void* k, const void* v, long q_stride,
                               long k_stride, long v_stride, const float* cos_sin, void* k_cache, void* v_cache,
                               const int* slot_mapping, int T, int Hq, int Hkv, int D, int block_size, int mode,
                               const void* q_norm_w, const void* k_norm_w, float norm_eps, int dtype,
                               int kv_fp8, const float* k_scales, const float* v_scales, hipStream_t s) {
  if (T == 0) return 0;
  if (D % 16 != 0) return -1;
  if ((q_norm_w == nullptr) != (k_norm_w == nullptr)) return -1;
  // the head's D/16 lanes must form an aligned power-of-two group for the shuffle reduction
  if (q_norm_w != nullptr && (mode == 2 || ((D / 16) & (D / 16 - 1)) != 0 || D / 16 > 64)) return -1;
  if (dtype == DT_BF16)
    return launch<bf16_t>(positions, q, k, v, q_stride, k_stride, v_stride, cos_sin, k_cache, v_cache,
                          slot_mapping, T, Hq, Hkv, D, block_size, mode, q_norm_w, k_norm_w, norm_eps, kv_fp8,
                          k_scales, v_scales, s);
  if (dtype == DT_F16)
    return launch<f16_t>(positions, q, k, v, q_stride, k_stride, v_stride, cos_sin, k_cache, v_cache,
                         slot_mapping, T, Hq, Hkv, D, block_size, mode, q_norm_w, k_norm_w, norm_eps, kv_fp8,
                         k_scales, v_scales, s);
  return -2;
}

// vdamd_rope_cache (mode 0) reading q, k and v as fp32 K-split partial sums part[nslabs][T][(Hq + 2 Hkv) D]
// (contiguous) and writing the rotated q to q_out[T][q_stride]; k and v go to the cache only.
VDAMD_API int vdamd_rope_cache_slabs(const int* positions, const float* part, int nslabs, void* q_out, long q_stride,
                                     const float* cos_sin, void* k_cache, void* v_cache, const int* slot_mapping,
                                     int T, int Hq, int Hkv, int D, int block_size, const void* q_norm_w,
                                     const void* k_norm_w, float norm_eps, int dtype, int kv_fp8,
                                     const float* k_scales, const float* v_scales, hipStream_t s) {
  if (T == 0) return 0;
  if (D % 16 != 0 || nslabs < 1 || part == nullptr || q_out == nullptr) return -1;
  if ((q_norm_w == nullptr) != (k_norm_w == nullptr)) return -1;
  if (q_norm_w != nullptr && (((D / 16) & (D / 16 - 1)) != 0 || D / 16 > 64)) return -1;
  if (dtype == DT_BF16)
    return launch<bf16_t>(positions, q_out, nullptr, nullptr, q_stride, 0, 0, cos_sin, k_cache, v_cache, slot_mapping,
                          T, Hq, Hkv, D, block_size, 0, q_norm_w, k_norm_w, norm_eps, kv_fp8, k_scales, v_scales, s,
                          part, nslabs);
  if (dtype == DT_F16)
    return launch<f16_t>(positions, q_out, nullptr, nullptr, q_stride, 0, 0, cos_sin, k_cache, v_cache, slot_mapping,
                         T, Hq, Hkv, D, block_size, 0, q_norm_w, k_norm_w, norm_eps, kv_fp8, k_scales, v_scales, s,
                         part, nslabs);
  return -2;
}
