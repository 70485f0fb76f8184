// Paged attention for gfx950: split-KV decode and varlen causal prefill.
//
// Cache layouts make K and V^T direct MFMA operands from global memory (no LDS
// staging, no transposes):
//   k_cache[num_blocks][Hkv][block_size][D]     key row contiguous in D
//   v_cache[num_blocks][Hkv][D/16][block_size/8][16][8]
//       V^T in MFMA-fragment order: the 8 keys of one D row are a 16-byte chunk, 16 D rows of a
//       key group are adjacent (256 bytes), and the key groups of a block follow each other inside
//       a 16-row slab.  One V^T fragment load of a wave (16 D rows x 32 keys) is one contiguous
//       1 KiB run, and a new token's values land in D/16 runs of 256 bytes per head.
//
// Both kernels use the "swapped" formulation on v_mfma_f32_16x16x32:
//   S^T[key][q]  = K[key][:] . Q[q][:]          A = K rows,   B = Q^T
//   O^T[d][q]   += V^T[d][key] . P^T[key][q]    A = V^T rows, B = P^T
// With C/D layout (col = lane&15, row = 4*(lane>>4)+reg) a lane holds, for ONE
// query column q = lane&15, four consecutive keys per S^T tile.  Two S^T tiles
// whose K rows are interleaved (tile0 row i <-> key 8*(i>>2)+(i&3), tile1 the
// same +4) give every lane 8 consecutive keys 8*(lane>>4)..+7 of a 32-key
// group - exactly the B-operand layout of the PV product, so P never leaves
// registers and V^T is fetched with plain 16-byte loads.  The softmax
// running max is shared by the four lanes that hold one query column (two
// xor-shuffles per 32 keys); row sums stay lane-local until the end.
#include "common.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E_F = 1.4426950408889634f;
constexpr int MAX_SPLITS = 64;

template <typename T>
__device__ __forceinline__ typename Vec8<T>::type zero8() {
  typename Vec8<T>::type z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (T)0.f;
  return z;
}

// 8 OCP e4m3 bytes -> 8 elements of T (gfx950 packed convert, two elements per instruction).
// The instruction's scale operand is only ever given 1.0 here: the per-head K scale is folded
// into the softmax scale and the V scale into the output normalisation, which costs nothing
// per element and does not depend on how the hardware rounds a general scale.
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type fp8x8_to(uint2 w, float scale) {
  union { uint32_t u[4]; typename Vec8<T>::type v; } r;
  if constexpr (sizeof(T) == 2 && __is_same(T, bf16_t)) {
    r.u[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.x, scale, false));
    r.u[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.x, scale, true));
    r.u[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.y, scale, false));
    r.u[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.y, scale, true));
  } else {
    r.u[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.x, scale, false));
    r.u[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.x, scale, true));
    r.u[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.y, scale, false));
    r.u[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.y, scale, true));
  }
  // Two wait states between the packed converts and whatever reads their results.  Without them some
  // instantiations (D = 96 prefill over an e4m3 cache, after an unrelated change of the surrounding
  // schedule) produced wrong products for a few percent of the outputs, deterministically: the same
  // result-forwarding behaviour the NVFP4 GEMM met with v_cvt_pk_f32_fp8, which the compiler's hazard
  // table does not cover.
  asm volatile("s_nop 1" : "+v"(r.u[0]), "+v"(r.u[1]), "+v"(r.u[2]), "+v"(r.u[3]));
  return r.v;
}

// 8 consecutive cache elements starting at element offset `off`: 16 bytes of T, or 8 bytes
// of e4m3 widened to T (KV8: the cache holds one byte per element, same layouts).
template <typename T, bool KV8>
__device__ __forceinline__ typename Vec8<T>::type ld_kv(const void* cache, long off, float scale) {
  if constexpr (KV8) {
    return fp8x8_to<T>(*reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(cache) + off), scale);
  } else {
    return ld8<T>(reinterpret_cast<const T*>(cache) + off);
  }
}

// ---------------------------------------------------------------------------
// One 32-key group for NQ query sub-tiles of 16 rows each.
//   qf[n][s]   : Q^T B-operand fragments (s = k-step over D)
//   acc[n][dt] : O^T accumulators, m[n]/l[n] softmax state (per lane)
//   qpos_hi[n] : largest key index the lane's query row may attend to
// ---------------------------------------------------------------------------
// K / V^T fragments of one 32-key group, as loaded from the paged cache.
template <typename T, int D>
struct KVFrag {
  typename Vec8<T>::type kfa[D / 32], kfb[D / 32], vf[D / 16];
};

template <typename T, int D, bool KV8>
__device__ __forceinline__ void load_kv_group(KVFrag<T, D>& f, const void* __restrict__ k_cache,
                                              const void* __restrict__ v_cache, const int* __restrict__ bt_row,
                                              int kvh, int Hkv, int block_size, int base, int seq_len,
                                              float k_scale, float v_scale) {
  const int lane = threadIdx.x & 63;
  const int r16 = lane & 15, g4 = lane >> 4;
  // ---- K rows for the two interleaved tiles
  const int key0 = base + 8 * (r16 >> 2) + (r16 & 3);        // tile 0 row; tile 1 row = key0 + 4
  const int ka = min(key0, seq_len - 1);
  const int kb = min(key0 + 4, seq_len - 1);
  const long blk_a = bt_row[ka / block_size], blk_b = bt_row[kb / block_size];
  const long ka_off = ((blk_a * Hkv + kvh) * block_size + (ka % block_size)) * (long)D + 8 * g4;
  const long kb_off = ((blk_b * Hkv + kvh) * block_size + (kb % block_size)) * (long)D + 8 * g4;
#pragma unroll
  for (int s = 0; s < D / 32; ++s) {
    f.kfa[s] = ld_kv<T, KV8>(k_cache, ka_off + 32 * s, k_scale);
    f.kfb[s] = ld_kv<T, KV8>(k_cache, kb_off + 32 * s, k_scale);
  }
  // ---- V^T rows: this lane's 8 keys are base + 8*g4 .. +7 (one block: block_size % 8 == 0)
  const int vkey = min(base + 8 * g4, seq_len - 1) & ~7;
  const long blk_v = bt_row[vkey / block_size];
  const int slab = block_size * 16;                           // elements of one 16-row slab of a block
  const long v_off = (blk_v * Hkv + kvh) * (long)(D / 16) * slab + ((vkey % block_size) >> 3) * 128 + 8 * r16;
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt) f.vf[dt] = ld_kv<T, KV8>(v_cache, v_off + (long)dt * slab, v_scale);
}

// Score capping (Gemma-2): cap * tanh(score / cap) on the natural-log scale; `s2` and the result are
// scores already multiplied by log2(e).
// tanh(x) = 1 - 2 / (1 + e^(2x)) with x = s2 * ln2 / cap, so e^(2x) = 2^(s2 * 2 / cap): one exponential and one
// reciprocal (absolute error ~1e-7 of the cap, far below a bf16 score's spacing) where the library's tanhf is a
// branchy polynomial.
__device__ __forceinline__ float soft_cap_log2(float s2, float cap) {
  const float e = fast_exp2(s2 * (2.f / cap));
  const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
  return t * (cap * 1.4426950408889634f);
}

template <typename T, int D, int NQ>
__device__ __forceinline__ void attend_loaded(
    const KVFrag<T, D>& f, int base, int seq_len, float scale_log2e, float softcap,
    const typename Vec8<T>::type (&qf)[NQ][D / 32], f32x4 (&acc)[NQ][D / 16], float (&m)[NQ], float (&l)[NQ],
    const int (&qpos_hi)[NQ], const int (&qpos_lo)[NQ], float alibi_l2 = 0.f) {
  // alibi_l2: this lane's query head's ALiBi slope times log2(e) (0: none); the bias of key k for the
  // query at position p is slope * (k - p)
  using V = typename Vec8<T>::type;
  const int g4 = (threadIdx.x & 63) >> 4;
  const int kidx0 = base + 8 * g4;     // key index of P element j is kidx0 + j
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
      s0 = mfma16(f.kfa[s], qf[n][s], s0);
      s1 = mfma16(f.kfb[s], qf[n][s], s1);
    }
    const int hi = min(qpos_hi[n], seq_len - 1);
    const int lo = qpos_lo[n];           // first key inside the sliding window (0 without one)
    float p[8];
    float tmax = NEG_BIG;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float sv = (j < 4 ? s0[j & 3] : s1[j & 3]) * scale_log2e;
      if (softcap > 0.f) sv = soft_cap_log2(sv, softcap);
      sv = fmaf(alibi_l2, (float)(kidx0 + j - qpos_hi[n]), sv);
      p[j] = (kidx0 + j <= hi && kidx0 + j >= lo) ? sv : NEG_BIG;
      tmax = fmaxf(tmax, p[j]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m[n], tmax);
    const float alpha = fast_exp2(m[n] - m_new);
    float psum = 0.f;
    V pf;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // (a row may have no visible key in a group below its window: its running max is still the
      // sentinel then, and the masked scores must not turn into exp2(0))
      const float e = p[j] <= 0.5f * NEG_BIG ? 0.f : fast_exp2(p[j] - m_new);
      psum += e;
      pf[j] = (T)e;
    }
    l[n] = l[n] * alpha + psum;
    m[n] = m_new;
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) {
      acc[n][dt] *= alpha;
      acc[n][dt] = mfma16(f.vf[dt], pf, acc[n][dt]);
    }
  }
}

template <typename T, int D, int NQ, bool KV8>
__device__ __forceinline__ void attend_group(
    const void* __restrict__ k_cache, const void* __restrict__ v_cache, const int* __restrict__ bt_row, int kvh,
    int Hkv, int block_size, int base, int seq_len, float scale_log2e, float softcap,
    const typename Vec8<T>::type (&qf)[NQ][D / 32], f32x4 (&acc)[NQ][D / 16], float (&m)[NQ], float (&l)[NQ],
    const int (&qpos_hi)[NQ], const int (&qpos_lo)[NQ], float k_scale, float v_scale, float alibi_l2 = 0.f) {
  KVFrag<T, D> f;
  load_kv_group<T, D, KV8>(f, k_cache, v_cache, bt_row, kvh, Hkv, block_size, base, seq_len, k_scale, v_scale);
  attend_loaded<T, D, NQ>(f, base, seq_len, scale_log2e, softcap, qf, acc, m, l, qpos_hi, qpos_lo, alibi_l2);
}

// ---------------------------------------------------------------------------
// Decode: one query token per sequence.  Workgroup = (kv-split, kv-head x
// q-tile, sequence); its 4 waves stride over the split's 32-key groups and are
// merged through LDS.  The 16 MFMA columns carry the GQA group's query heads.
// ---------------------------------------------------------------------------
template <typename T, int D, bool KV8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void paged_decode_kernel(
    T* __restrict__ out, float* __restrict__ part_o, float* __restrict__ part_ml, const T* __restrict__ q,
    long q_stride, const void* __restrict__ k_cache, const void* __restrict__ v_cache,
    const int* __restrict__ block_table, int bt_stride, const int* __restrict__ seq_lens, float scale_log2e,
    int Hq, int Hkv, int block_size, int num_splits, long out_stride, const float* __restrict__ k_scales,
    const float* __restrict__ v_scales, int window, float softcap, const float* __restrict__ sinks,
    const float* __restrict__ alibi) {
  // sinks[Hq] (natural-log units): an extra score per query head that takes part in the softmax's
  // denominator only (attention sinks); alibi[Hq]: slopes of a linear position bias.  Either may be null.
  using V = typename Vec8<T>::type;
  constexpr int LDS_ROW = D + 4;
  __shared__ float s_o[4][16][LDS_ROW];
  __shared__ float s_m[4][16];
  __shared__ float s_l[4][16];

  const int split = blockIdx.x, b = blockIdx.z;
  const int G = Hq / Hkv;
  const int q_tiles = (G + 15) >> 4;
  const int kvh = blockIdx.y / q_tiles, qt = blockIdx.y - kvh * q_tiles;
  const int qh0 = kvh * G + qt * 16;
  const int nq = min(16, G - qt * 16);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r16 = lane & 15, g4 = lane >> 4;
  const int seq_len = seq_lens[b];
  const float k_scale = (KV8 && k_scales != nullptr) ? k_scales[kvh] : 1.f;
  const float v_scale = (KV8 && v_scales != nullptr) ? v_scales[kvh] : 1.f;

  // sliding window: the query (position seq_len - 1) sees the last `window` keys only
  const int key_lo = window > 0 ? max(0, seq_len - window) : 0;
  const int g_first = key_lo >> 5;
  const int n_groups = ((seq_len + 31) >> 5) - g_first;
  const int gps = (n_groups + num_splits - 1) / num_splits;
  const int g_begin = g_first + split * gps;
  const int g_end = min(g_begin + gps, g_first + n_groups);

  V qf[1][D / 32];
#pragma unroll
  for (int s = 0; s < D / 32; ++s)
    qf[0][s] = (r16 < nq) ? ld8<T>(q + b * q_stride + (long)(qh0 + r16) * D + 32 * s + 8 * g4) : zero8<T>();

  f32x4 acc[1][D / 16];
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt) acc[0][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[1] = {NEG_BIG}, l[1] = {0.f};
  const int qpos_hi[1] = {seq_len - 1};
  const int qpos_lo[1] = {key_lo};
  const int* bt_row = block_table + (long)b * bt_stride;
  const float alibi_l2 = (alibi != nullptr && r16 < nq) ? alibi[qh0 + r16] * LOG2E_F : 0.f;

  // no register double-buffering here: it halves the occupancy (202 VGPRs) and
  // measured 3-8% slower than letting 4 resident workgroups per CU hide the latency
  for (int g = g_begin + wid; g < g_end; g += 4)
    attend_group<T, D, 1, KV8>(k_cache, v_cache, bt_row, kvh, Hkv, block_size, g * 32, seq_len,
                               scale_log2e * k_scale, softcap, qf, acc, m, l, qpos_hi, qpos_lo, 1.f, 1.f, alibi_l2);

  float lsum = l[0];
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (g4 == 0) { s_m[wid][r16] = m[0]; s_l[wid][r16] = lsum; }
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt)
    *reinterpret_cast<f32x4*>(&s_o[wid][r16][16 * dt + 4 * g4]) = acc[0][dt];
  __syncthreads();

  for (int e = threadIdx.x; e < 16 * D; e += 256) {
    const int qi = e / D, d = e - qi * D;
    if (qi >= nq) continue;
    const float m0 = s_m[0][qi], m1 = s_m[1][qi], m2 = s_m[2][qi], m3 = s_m[3][qi];
    const float mm = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    const float a0 = fast_exp2(m0 - mm), a1 = fast_exp2(m1 - mm), a2 = fast_exp2(m2 - mm), a3 = fast_exp2(m3 - mm);
    const float o = (s_o[0][qi][d] * a0 + s_o[1][qi][d] * a1 + s_o[2][qi][d] * a2 + s_o[3][qi][d] * a3) * v_scale;
    float lt = s_l[0][qi] * a0 + s_l[1][qi] * a1 + s_l[2][qi] * a2 + s_l[3][qi] * a3;
    const int qh = qh0 + qi;
    if (num_splits == 1) {
      if (sinks != nullptr) lt += fast_exp2(sinks[qh] * LOG2E_F - mm);       // (with splits: added by the merge kernel)
      out[b * out_stride + (long)qh * D + d] = (T)(lt > 0.f ? o / lt : 0.f);
    } else {
      const long pidx = ((long)b * Hq + qh) * num_splits + split;
      part_o[pidx * D + d] = o;
      if (d == 0) { part_ml[2 * pidx] = mm; part_ml[2 * pidx + 1] = lt; }
    }
  }
}

// Short contexts, large batches: one WAVE per (sequence, KV head, 16-query tile), four sequences per
// workgroup.  A context of a few hundred keys is 5-10 groups of 32; giving it four waves leaves
// most of them one group each and then pays an LDS exchange, a barrier and a cross-wave combine
// per sequence.  Here a wave walks all groups of its sequence and writes its rows itself: no LDS,
// no barrier, and consecutive sequences of a workgroup keep the load queues of the CU busy.
// (three waves per SIMD: at four, 128 registers, the q fragments spill and are reloaded inside the key loop)
template <typename T, int D, bool KV8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void paged_decode_wave_kernel(
    T* __restrict__ out, const T* __restrict__ q, long q_stride, const void* __restrict__ k_cache,
    const void* __restrict__ v_cache, const int* __restrict__ block_table, int bt_stride,
    const int* __restrict__ seq_lens, float scale_log2e, int B, int Hq, int Hkv, int block_size, long out_stride,
    const float* __restrict__ k_scales, const float* __restrict__ v_scales, int window, float softcap,
    const float* __restrict__ sinks, const float* __restrict__ alibi) {
  using V = typename Vec8<T>::type;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wid;
  if (b >= B) return;
  const int G = Hq / Hkv;
  const int q_tiles = (G + 15) >> 4;
  const int kvh = blockIdx.y / q_tiles, qt = blockIdx.y - kvh * q_tiles;
  const int qh0 = kvh * G + qt * 16;
  const int nq = min(16, G - qt * 16);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int seq_len = seq_lens[b];
  const float k_scale = (KV8 && k_scales != nullptr) ? k_scales[kvh] : 1.f;
  const float v_scale = (KV8 && v_scales != nullptr) ? v_scales[kvh] : 1.f;

  V qf[1][D / 32];
#pragma unroll
  for (int s = 0; s < D / 32; ++s)
    qf[0][s] = (r16 < nq) ? ld8<T>(q + b * q_stride + (long)(qh0 + r16) * D + 32 * s + 8 * g4) : zero8<T>();
  f32x4 acc[1][D / 16];
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt) acc[0][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[1] = {NEG_BIG}, l[1] = {0.f};
  const int qpos_hi[1] = {seq_len - 1};
  const int qpos_lo[1] = {window > 0 ? max(0, seq_len - window) : 0};
  const int* bt_row = block_table + (long)b * bt_stride;
  const int n_groups = (seq_len + 31) >> 5;
  const float alibi_l2 = (alibi != nullptr && r16 < nq) ? alibi[qh0 + r16] * LOG2E_F : 0.f;
  for (int g = qpos_lo[0] >> 5; g < n_groups; ++g)
    attend_group<T, D, 1, KV8>(k_cache, v_cache, bt_row, kvh, Hkv, block_size, g * 32, seq_len,
                               scale_log2e * k_scale, softcap, qf, acc, m, l, qpos_hi, qpos_lo, 1.f, 1.f, alibi_l2);
  float lsum = l[0];
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (r16 >= nq) return;
  if (sinks != nullptr) lsum += fast_exp2(sinks[qh0 + r16] * LOG2E_F - m[0]);
  const float inv = lsum > 0.f ? v_scale / lsum : 0.f;
  T* orow = out + b * out_stride + (long)(qh0 + r16) * D + 4 * g4;
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt) {
    typename Vec4<T>::type o4;
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] = (T)(acc[0][dt][r] * inv);
    *reinterpret_cast<typename Vec4<T>::type*>(orow + 16 * dt) = o4;
  }
}

template <typename T, int D>
__global__ void decode_reduce_kernel(T* __restrict__ out, const float* __restrict__ part_o,
                                     const float* __restrict__ part_ml, int Hq, int num_splits, long out_stride,
                                     const float* __restrict__ sinks) {
  const int qh = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  const long base = ((long)b * Hq + qh) * num_splits;
  float mm = NEG_BIG;
  for (int s = 0; s < num_splits; ++s) mm = fmaxf(mm, part_ml[2 * (base + s)]);
  float o = 0.f, lt = 0.f;
  for (int s = 0; s < num_splits; ++s) {
    const float a = fast_exp2(part_ml[2 * (base + s)] - mm);
    lt += part_ml[2 * (base + s) + 1] * a;
    o += part_o[(base + s) * D + d] * a;
  }
  if (sinks != nullptr) lt += fast_exp2(sinks[qh] * LOG2E_F - mm);
  out[b * out_stride + (long)qh * D + d] = (T)(lt > 0.f ? o / lt : 0.f);
}

// ---------------------------------------------------------------------------
// Prefill / chunked prefill: sequence z owns query rows qsl[z]..qsl[z+1], the
// last q_len positions of its seq_len tokens; all keys come from the paged
// cache (the chunk's own K/V were written before this kernel runs).
// Workgroup = 128 query tokens of one head; each wave owns 32 tokens as two
// 16-row MFMA column tiles that share every K / V^T fragment it loads.
// ---------------------------------------------------------------------------
template <typename T, int D, bool KV8>
__global__ __launch_bounds__(256) void paged_prefill_kernel(
    T* __restrict__ out, const T* __restrict__ q, long q_stride, const void* __restrict__ k_cache,
    const void* __restrict__ v_cache, const int* __restrict__ block_table, int bt_stride,
    const int* __restrict__ query_start_loc, const int* __restrict__ seq_lens, float scale_log2e, int Hq, int Hkv,
    int block_size, long out_stride, const float* __restrict__ k_scales, const float* __restrict__ v_scales,
    int window, float softcap, const float* __restrict__ sinks, const float* __restrict__ alibi) {
  using V = typename Vec8<T>::type;
  constexpr int NQ = 2;
  const int z = blockIdx.z, h = blockIdx.y;
  const int q0 = query_start_loc[z];
  const int q_len = query_start_loc[z + 1] - q0;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r16 = lane & 15, g4 = lane >> 4;
  const int t_wave = blockIdx.x * 128 + wid * 32;       // first token of this wave within the chunk
  if (t_wave >= q_len) return;
  const int seq_len = seq_lens[z];
  const int ctx = seq_len - q_len;
  const int kvh = h / (Hq / Hkv);
  const float k_scale = (KV8 && k_scales != nullptr) ? k_scales[kvh] : 1.f;
  const float v_scale = (KV8 && v_scales != nullptr) ? v_scales[kvh] : 1.f;

  V qf[NQ][D / 32];
  int qpos_hi[NQ], qpos_lo[NQ];
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    const int tok = t_wave + 16 * n + r16;
    qpos_hi[n] = ctx + tok;
    qpos_lo[n] = window > 0 ? max(0, ctx + tok - window + 1) : 0;
#pragma unroll
    for (int s = 0; s < D / 32; ++s)
      qf[n][s] = (tok < q_len) ? ld8<T>(q + (long)(q0 + tok) * q_stride + (long)h * D + 32 * s + 8 * g4)
                               : zero8<T>();
  }
  f32x4 acc[NQ][D / 16];
  float m[NQ], l[NQ];
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    m[n] = NEG_BIG; l[n] = 0.f;
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) acc[n][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int* bt_row = block_table + (long)z * bt_stride;
  const int last_key = min(seq_len - 1, ctx + min(t_wave + 31, q_len - 1));
  const int n_groups = (last_key >> 5) + 1;
  const int g_first = window > 0 ? max(0, ctx + t_wave - window + 1) >> 5 : 0;     // the wave's earliest window edge
  const float alibi_l2 = alibi != nullptr ? alibi[h] * LOG2E_F : 0.f;
  for (int g = g_first; g < n_groups; ++g)
    attend_group<T, D, NQ, KV8>(k_cache, v_cache, bt_row, kvh, Hkv, block_size, g * 32, seq_len,
                                scale_log2e * k_scale, softcap, qf, acc, m, l, qpos_hi, qpos_lo, 1.f, 1.f, alibi_l2);
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    float lsum = l[n];
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (sinks != nullptr) lsum += fast_exp2(sinks[h] * LOG2E_F - m[n]);
    const float inv = lsum > 0.f ? v_scale / lsum : 0.f;
    const int tok = t_wave + 16 * n + r16;
    if (tok < q_len) {
      T* orow = out + (long)(q0 + tok) * out_stride + (long)h * D + 4 * g4;
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) {
        typename Vec4<T>::type o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = (T)(acc[n][dt][r] * inv);
        *reinterpret_cast<typename Vec4<T>::type*>(orow + 16 * dt) = o4;
      }
    }
  }
}


// ---------------------------------------------------------------------------
// Prefill, LDS-staged (D = 128): workgroup = 128 query tokens of one head,
// 4 waves x 32 query columns on v_mfma_f32_32x32x16.  K (64 keys x 256 B) and
// V^T (128 rows x 128 B) tiles are staged through double-buffered LDS with
// register prefetch of the next tile (global loads issued before the MFMA
// phase, LDS writes after it, one barrier per tile) and XOR-swizzled 16-byte
// chunks so every ds_read_b128 is bank-conflict free.
//
// Swapped products again: S^T[key][q] = K.Q^T, O^T[d][q] += V^T.P^T.  The K
// rows fed to the MFMA are permuted (bits 2 and 3 of the row index swapped) so
// that accumulator registers 8s..8s+7 of a lane are 8 *consecutive* keys
// 16s + 8*(lane>>5) + j - the natural B-operand layout of the PV product.
// ---------------------------------------------------------------------------
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// 16 bytes per lane from global memory straight into LDS: the wave's 64 pieces land at lds_wave_base + 16 * lane
typedef __attribute__((address_space(3))) void prefill_lds_void;
typedef __attribute__((address_space(1))) void prefill_glb_void;
__device__ __forceinline__ void prefill_dma16(const void* g, void* lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the builtin exists in the device pass only)
  __builtin_amdgcn_global_load_lds((prefill_glb_void*)const_cast<void*>(g), (prefill_lds_void*)lds_wave_base, 16, 0, 0);
#endif
}

#define PREFILL_LDS_D 128
#define PREFILL_LDS_OCC 2
#define PREFILL_LDS_NAME paged_prefill_lds_kernel
#define PREFILL_LDS_BIAS 0
#define PREFILL_LDS_EXTRA_PARAMS
#include "attention_prefill_lds.inc"
#undef PREFILL_LDS_NAME
#undef PREFILL_LDS_BIAS
#undef PREFILL_LDS_EXTRA_PARAMS
#define PREFILL_LDS_NAME paged_prefill_lds_bias_kernel
#define PREFILL_LDS_BIAS 1
#define PREFILL_LDS_EXTRA_PARAMS , const float *__restrict__ sinks, const float *__restrict__ alibi
#include "attention_prefill_lds.inc"
#undef PREFILL_LDS_NAME
#undef PREFILL_LDS_BIAS
#undef PREFILL_LDS_EXTRA_PARAMS
#undef PREFILL_LDS_D
#undef PREFILL_LDS_OCC

// head size 64 (Llama-3.2-1B, gpt-oss, ...): half the registers and LDS of the kernel above
#define PREFILL_LDS_D 64
#define PREFILL_LDS_OCC 2
#define PREFILL_LDS_NAME paged_prefill_lds64_kernel
#define PREFILL_LDS_BIAS 0
#define PREFILL_LDS_EXTRA_PARAMS
#include "attention_prefill_lds.inc"
#undef PREFILL_LDS_NAME
#undef PREFILL_LDS_BIAS
#undef PREFILL_LDS_EXTRA_PARAMS
#define PREFILL_LDS_NAME paged_prefill_lds64_bias_kernel
#define PREFILL_LDS_BIAS 1
#define PREFILL_LDS_EXTRA_PARAMS , const float *__restrict__ sinks, const float *__restrict__ alibi
#include "attention_prefill_lds.inc"
#undef PREFILL_LDS_NAME
#undef PREFILL_LDS_BIAS
#undef PREFILL_LDS_EXTRA_PARAMS
#undef PREFILL_LDS_D
#undef PREFILL_LDS_OCC

// head size 256 (Gemma-2 / 3, padded odd sizes): 128 KB of LDS and ~330 registers, one workgroup per CU
#define PREFILL_LDS_D 256
#define PREFILL_LDS_OCC 1
#define PREFILL_LDS_NAME paged_prefill_lds256_kernel
#define PREFILL_LDS_BIAS 0
#define PREFILL_LDS_EXTRA_PARAMS
#include "attention_prefill_lds.inc"
#undef PREFILL_LDS_NAME
#undef PREFILL_LDS_BIAS
#undef PREFILL_LDS_EXTRA_PARAMS
#define PREFILL_LDS_NAME paged_prefill_lds256_bias_kernel
#define PREFILL_LDS_BIAS 1
#define PREFILL_LDS_EXTRA_PARAMS , const float *__restrict__ sinks, const float *__restrict__ alibi
#include "attention_prefill_lds.inc"
#undef PREFILL_LDS_NAME
#undef PREFILL_LDS_BIAS
#undef PREFILL_LDS_EXTRA_PARAMS
#undef PREFILL_LDS_D
#undef PREFILL_LDS_OCC

template <typename T, int D, bool KV8>
int launch_decode(void* out, float* ws, const void* q, long q_stride, const void* kc, const void* vc, const int* bt,
                  int bt_stride, const int* seq_lens, float scale, int B, int Hq, int Hkv, int block_size,
                  int num_splits, long out_stride, const float* k_scales, const float* v_scales, int window,
                  float softcap, const float* sinks, const float* alibi, hipStream_t s) {
  const int G = Hq / Hkv;
  const int q_tiles = (G + 15) / 16;
  float* part_o = ws;
  float* part_ml = ws ? ws + (long)B * Hq * num_splits * D : nullptr;
  const float sl2 = scale * 1.4426950408889634f;
  if (num_splits == 0) {      // one wave per sequence
    hipLaunchKernelGGL((paged_decode_wave_kernel<T, D, KV8>), dim3((B + 3) / 4, Hkv * q_tiles), dim3(256), 0, s,
                       (T*)out, (const T*)q, q_stride, kc, vc, bt, bt_stride, seq_lens, sl2, B, Hq, Hkv, block_size,
                       out_stride, k_scales, v_scales, window, softcap, sinks, alibi);
    HIP_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL((paged_decode_kernel<T, D, KV8>), dim3(num_splits, Hkv * q_tiles, B), dim3(256), 0, s, (T*)out,
                     part_o, part_ml, (const T*)q, q_stride, kc, vc, bt, bt_stride, seq_lens,
                     sl2, Hq, Hkv, block_size, num_splits, out_stride, k_scales, v_scales, window, softcap, sinks,
                     alibi);
  HIP_LAUNCH_CHECK();
  if (num_splits > 1) {
    hipLaunchKernelGGL((decode_reduce_kernel<T, D>), dim3(Hq, B), dim3(D), 0, s, (T*)out, part_o, part_ml, Hq,
                       num_splits, out_stride, sinks);
    HIP_LAUNCH_CHECK();
  }
  return 0;
}

template <typename T, int D, bool KV8>
int launch_prefill(void* out, const void* q, long q_stride, const void* kc, const void* vc, const int* bt,
                   int bt_stride, const int* qsl, const int* seq_lens, float scale, int num_seqs, int max_q_len,
                   int Hq, int Hkv, int block_size, long out_stride, const float* k_scales, const float* v_scales,
                   int window, float softcap, const float* sinks, const float* alibi, hipStream_t s) {
  const float sl2 = scale * 1.4426950408889634f;
  const int tiles = (max_q_len + 127) / 128;
  if constexpr (D == 128 || D == 64 || D == 256) {
    static const bool lds_kernel = [] {          // VDAMD_PREFILL_LDS=0: measure the fragment-direct kernel instead
      const char* e = getenv("VDAMD_PREFILL_LDS");
      return e == nullptr || atoi(e) != 0;
    }();
    // longer chunks: LDS-staged 32x32 MFMA kernel (it addresses the cache by shifts: power-of-two blocks)
    if (max_q_len > 32 && lds_kernel && (block_size & (block_size - 1)) == 0) {
      static const int remap = [] {
        const char* e = getenv("VDAMD_PREFILL_XCD_REMAP");
        return e == nullptr ? 1 : atoi(e);
      }();
      const long total = (long)tiles * Hq * num_seqs;
      if (total > (1L << 30)) return -3;
      const int grid = remap ? (int)((total + 7) / 8 * 8) : (int)total;
#define LDS_LAUNCH(PLAIN, BIAS)                                                                                  \
  if (sinks != nullptr || alibi != nullptr)                                                                      \
    hipLaunchKernelGGL((BIAS<T, KV8>), dim3(grid), dim3(256), 0, s, (T*)out, (const T*)q, q_stride, kc, vc, bt,  \
                       bt_stride, qsl, seq_lens, sl2, Hq, Hkv, block_size, out_stride, tiles, (int)total, remap, \
                       k_scales, v_scales, window, softcap, sinks, alibi);                                       \
  else                                                                                                           \
    hipLaunchKernelGGL((PLAIN<T, KV8>), dim3(grid), dim3(256), 0, s, (T*)out, (const T*)q, q_stride, kc, vc, bt, \
                       bt_stride, qsl, seq_lens, sl2, Hq, Hkv, block_size, out_stride, tiles, (int)total, remap, \
                       k_scales, v_scales, window, softcap)
      if constexpr (D == 128) { LDS_LAUNCH(paged_prefill_lds_kernel, paged_prefill_lds_bias_kernel); }
      else if constexpr (D == 64) { LDS_LAUNCH(paged_prefill_lds64_kernel, paged_prefill_lds64_bias_kernel); }
      else { LDS_LAUNCH(paged_prefill_lds256_kernel, paged_prefill_lds256_bias_kernel); }
#undef LDS_LAUNCH
      HIP_LAUNCH_CHECK();
      return 0;
    }
  }
  hipLaunchKernelGGL((paged_prefill_kernel<T, D, KV8>), dim3(tiles, Hq, num_seqs), dim3(256), 0, s, (T*)out,
                     (const T*)q, q_stride, kc, vc, bt, bt_stride, qsl, seq_lens, sl2, Hq, Hkv,
                     block_size, out_stride, k_scales, v_scales, window, softcap, sinks, alibi);
  HIP_LAUNCH_CHECK();
  return 0;
}

}  // namespace

VDAMD_API int vdamd_max_splits() { return MAX_SPLITS; }

// workspace (floats) needed for B sequences: partial O + (m, l) per split
VDAMD_API long vdamd_decode_workspace_floats(int B, int Hq, int D) {
  return (long)B * Hq * MAX_SPLITS * (D + 2);
}

// kv_fp8 != 0: k_cache / v_cache hold OCP e4m3 bytes in the same layouts; k_scales / v_scales
// (one float per KV head, or null for 1.0) multiply the stored values back to their range.
VDAMD_API int vdamd_paged_attention_decode(void* out, float* workspace, const void* q, long q_stride,
                                           const void* k_cache, const void* v_cache, const int* block_table,
                                           int bt_stride, const int* seq_lens, float scale, int B, int Hq, int Hkv,
                                           int D, int block_size, int num_splits, long out_stride, int dtype,
                                           int kv_fp8, const float* k_scales, const float* v_scales, int window,
                                           float softcap, const float* sinks, const float* alibi, hipStream_t s) {
  if (B == 0) return 0;
  // num_splits == 0 selects the one-wave-per-sequence kernel (short contexts, large batches)
  if (Hq % Hkv != 0 || block_size % 8 != 0 || num_splits < 0 || num_splits > MAX_SPLITS) return -1;
  if (num_splits > 1 && workspace == nullptr) return -3;
#define CASE(TY, DD)                                                                                             \
  do {                                                                                                           \
    if (kv_fp8)                                                                                                  \
      return launch_decode<TY, DD, true>(out, workspace, q, q_stride, k_cache, v_cache, block_table, bt_stride,   \
                                         seq_lens, scale, B, Hq, Hkv, block_size, num_splits, out_stride,         \
                                         k_scales, v_scales, window, softcap, sinks, alibi, s);                            \
    return launch_decode<TY, DD, false>(out, workspace, q, q_stride, k_cache, v_cache, block_table, bt_stride,    \
                                        seq_lens, scale, B, Hq, Hkv, block_size, num_splits, out_stride, nullptr, \
                                        nullptr, window, softcap, sinks, alibi, s);                                      \
  } while (0)
  if (dtype == DT_BF16) {
    if (D == 128) CASE(bf16_t, 128);
    if (D == 64) CASE(bf16_t, 64);
    if (D == 96) CASE(bf16_t, 96);
    if (D == 256) CASE(bf16_t, 256);
  }
  if (dtype == DT_F16) {
    if (D == 128) CASE(f16_t, 128);
    if (D == 64) CASE(f16_t, 64);
    if (D == 96) CASE(f16_t, 96);
    if (D == 256) CASE(f16_t, 256);
  }
#undef CASE
  return -2;
}

VDAMD_API int vdamd_paged_attention_prefill(void* out, const void* q, long q_stride, const void* k_cache,
                                            const void* v_cache, const int* block_table, int bt_stride,
                                            const int* query_start_loc, const int* seq_lens, float scale,
                                            int num_seqs, int max_q_len, int Hq, int Hkv, int D, int block_size,
                                            long out_stride, int dtype, int kv_fp8, const float* k_scales,
                                            const float* v_scales, int window, float softcap, const float* sinks,
                                            const float* alibi, hipStream_t s) {
  if (num_seqs == 0 || max_q_len == 0) return 0;
  if (Hq % Hkv != 0 || block_size % 8 != 0) return -1;
#define CASE(TY, DD)                                                                                             \
  do {                                                                                                           \
    if (kv_fp8)                                                                                                  \
      return launch_prefill<TY, DD, true>(out, q, q_stride, k_cache, v_cache, block_table, bt_stride,             \
                                          query_start_loc, seq_lens, scale, num_seqs, max_q_len, Hq, Hkv,         \
                                          block_size, out_stride, k_scales, v_scales, window, softcap, sinks, alibi, s);   \
    return launch_prefill<TY, DD, false>(out, q, q_stride, k_cache, v_cache, block_table, bt_stride,              \
                                         query_start_loc, seq_lens, scale, num_seqs, max_q_len, Hq, Hkv,          \
                                         block_size, out_stride, nullptr, nullptr, window, softcap, sinks, alibi, s);    \
  } while (0)
  if (dtype == DT_BF16) {
    if (D == 128) CASE(bf16_t, 128);
    if (D == 64) CASE(bf16_t, 64);
    if (D == 96) CASE(bf16_t, 96);
    if (D == 256) CASE(bf16_t, 256);
  }
  if (dtype == DT_F16) {
    if (D == 128) CASE(f16_t, 128);
    if (D == 64) CASE(f16_t, 64);
    if (D == 96) CASE(f16_t, 96);
    if (D == 256) CASE(f16_t, 256);
  }
#undef CASE
  return -2;
}
