// Multi-head latent attention (DeepSeek-V2 / V3) over a LATENT paged cache, gfx950.
//
// With the up-projections absorbed into the query and the output (q_lat = q_nope . W_UK, out = o_lat . W_UV^T),
// every query head of a token attends over ONE shared key per position,
//     key   = [ RMSNorm(c_kv) (RANK values) | rotated k_rope (ROPE values) ]      DK = RANK + ROPE
//     value =   RMSNorm(c_kv)                                                      DV = RANK
// so the cache holds DK + DV values per token and layer instead of heads x (DK' + DV') expanded ones
// (DeepSeek-V3: 1088 against 49152).  The two forms of the latent are stored in the layouts the other
// attention kernels use, which make them direct MFMA operands from global memory:
//     k_cache[num_blocks][block_size][DK]                    key rows
//     v_cache[num_blocks][DV/16][block_size/8][16][8]        V^T in fragment order
//
// One workgroup = one token's tile of 16 query heads (x one KV split).  What makes this shape different from
// grouped-query attention is the width: a wave that held all of Q (DK/32 fragments), all of a 32-key K group
// and the whole O^T accumulator (DV/16 tiles) would need ~500 registers.  The four waves of a workgroup split
// BOTH reductions instead:
//   * wave w multiplies its quarter of the DK dimension (Q and K fragments of its k-steps only) and leaves a
//     partial S^T tile pair in LDS; after one barrier every wave adds the four partials - 8 floats per lane,
//     same lane in every wave, conflict-free - and all four run the identical online softmax on the sum;
//   * wave w owns DV/4 columns of the output: its V^T fragments and accumulators are a quarter as well.
// No byte of the cache is read twice by a workgroup, 2 waves per SIMD fit (<= 256 registers), and the exchange
// (2 KiB per group, double-buffered so one barrier per 32 keys suffices) is small next to the 70 KB of cache a
// group streams.  The next group's fragments are requested before the current group is multiplied.
//
// Rows are tokens, not sequences: a prefill chunk, a decode batch and a speculative verification are the same
// launch with `token_seq[t]` naming the block-table row and `seq_lens[t]` = position + 1 the causal limit of
// token t.  (A long prompt therefore streams its keys once per query token - from L2 for the most part, since
// neighbouring tokens run side by side; the expanded-cache prefill kernel reads them once per 128 tokens.)
//
// FlashMLA / vLLM's MLA backends are the CUDA counterparts; the reference framework reaches them through the
// vLLM image it runs (/root/reference/Dockerfile:1).
#include "common.h"

namespace {

constexpr float MLA_NEG_BIG = -1.0e30f;
constexpr int MLA_MAX_SPLITS = 64;

// An e4m3 cache (one byte per value, same layouts, scale 1: the normalised latent and the rotated key part are
// O(1) numbers) halves the bytes a decode step streams; fragments are widened in registers on their way in.
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type mla_fp8x8_to(uint2 w) {
  union { uint32_t u[4]; typename Vec8<T>::type v; } r;
  if constexpr (__is_same(T, bf16_t)) {
    r.u[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.x, 1.f, false));
    r.u[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.x, 1.f, true));
    r.u[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.y, 1.f, false));
    r.u[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w.y, 1.f, true));
  } else {
    r.u[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.x, 1.f, false));
    r.u[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.x, 1.f, true));
    r.u[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.y, 1.f, false));
    r.u[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.y, 1.f, true));
  }
  // (two wait states behind the packed converts: their results are not safe to read earlier, and the
  // compiler's hazard table does not know it - see fp8x8_to in attention.hip)
  asm volatile("s_nop 1" : "+v"(r.u[0]), "+v"(r.u[1]), "+v"(r.u[2]), "+v"(r.u[3]));
  return r.v;
}

// 8 consecutive cache values from element offset `off`, as they are loaded (e4m3: 8 bytes, widened only where
// they are used - a fragment requested one key group ahead must not be waited for at the request)
template <typename T, bool KV8> struct MlaRaw { using type = typename Vec8<T>::type; };
template <typename T> struct MlaRaw<T, true> { using type = uint2; };

template <typename T, bool KV8>
__device__ __forceinline__ typename MlaRaw<T, KV8>::type mla_ld(const void* cache, long off) {
  if constexpr (KV8) {
    return *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(cache) + off);
  } else {
    return ld8<T>(reinterpret_cast<const T*>(cache) + off);
  }
}

template <typename T, bool KV8>
__device__ __forceinline__ typename Vec8<T>::type mla_widen(typename MlaRaw<T, KV8>::type raw) {
  if constexpr (KV8) {
    return mla_fp8x8_to<T>(raw);
  } else {
    return raw;
  }
}

// one value into the cache at element offset `off`
template <typename T, bool KV8>
__device__ __forceinline__ void mla_st(void* cache, long off, float v) {
  if constexpr (KV8) {
    const float c = fminf(fmaxf(v, -448.f), 448.f);
    reinterpret_cast<uint8_t*>(cache)[off] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(c, 0.f, 0, false) & 0xff);
  } else {
    reinterpret_cast<T*>(cache)[off] = (T)v;
  }
}

template <typename T, int DK, int DV, bool KV8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void mla_attention_kernel(
    T* __restrict__ out, long out_stride, float* __restrict__ part_o, float* __restrict__ part_ml,
    const T* __restrict__ q, long q_stride, const void* __restrict__ k_cache, const void* __restrict__ v_cache,
    const int* __restrict__ block_table, int bt_stride, const int* __restrict__ token_seq,
    const int* __restrict__ seq_lens, float scale_log2e, int Hq, int block_size, int num_splits) {
  using V = typename Vec8<T>::type;
  constexpr int KSTEPS = DK / 32;              // 32-wide steps of the QK^T reduction
  constexpr int KS = (KSTEPS + 3) / 4;         // ... per wave (the last waves may run short: zero Q there)
  constexpr int VT = DV / 64;                  // 16-row tiles of O^T per wave
  static_assert(DK % 32 == 0 && DV % 64 == 0, "unsupported latent width");
  __shared__ float s_part[2][4][64][8];

  const int split = blockIdx.x, qt = blockIdx.y, t = blockIdx.z;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int qh0 = qt * 16;
  const int nq = min(16, Hq - qh0);
  const int seq_len = seq_lens[t];
  const int* bt_row = block_table + (long)(token_seq != nullptr ? token_seq[t] : t) * bt_stride;

  const int n_groups = (seq_len + 31) >> 5;
  const int gps = (n_groups + num_splits - 1) / num_splits;
  const int g_begin = split * gps;
  const int g_end = min(g_begin + gps, n_groups);

  // this wave's k-steps: global step wid * KS + s, clamped for the address and zeroed in Q when past the end
  V qf[KS];
  int kstep_off[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int sg = wid * KS + s;
    kstep_off[s] = 32 * min(sg, KSTEPS - 1) + 8 * g4;
    V z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (T)0.f;
    qf[s] = (sg < KSTEPS && r16 < nq) ? ld8<T>(q + t * q_stride + (long)(qh0 + r16) * DK + kstep_off[s]) : z;
  }
  f32x4 acc[VT];
#pragma unroll
  for (int i = 0; i < VT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = MLA_NEG_BIG, l = 0.f;

  const int slab = block_size * 16;            // elements of one 16-row slab of a V^T block
  using R = typename MlaRaw<T, KV8>::type;
  struct Frag { R ka[KS], kb[KS], vf[VT]; };
  auto load = [&](Frag& f, int g) {
    const int base = g * 32;
    // K rows of the two interleaved S^T tiles: tile 0 row i <-> key 8 (i >> 2) + (i & 3), tile 1 the same + 4
    const int key0 = base + 8 * (r16 >> 2) + (r16 & 3);
    const int ka = min(key0, seq_len - 1), kb = min(key0 + 4, seq_len - 1);
    const long ra = ((long)bt_row[ka / block_size] * block_size + (ka % block_size)) * DK;
    const long rb = ((long)bt_row[kb / block_size] * block_size + (kb % block_size)) * DK;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f.ka[s] = mla_ld<T, KV8>(k_cache, ra + kstep_off[s]);
      f.kb[s] = mla_ld<T, KV8>(k_cache, rb + kstep_off[s]);
    }
    // V^T: this lane's 8 keys are base + 8 g4 .. + 7 (inside one block: block_size % 8 == 0)
    const int vkey = min(base + 8 * g4, seq_len - 1) & ~7;
    const long vo = (long)bt_row[vkey / block_size] * (DV / 16) * slab + ((vkey % block_size) >> 3) * 128 + 8 * r16;
#pragma unroll
    for (int i = 0; i < VT; ++i) f.vf[i] = mla_ld<T, KV8>(v_cache, vo + (long)(wid * VT + i) * slab);
  };

  Frag cur, nxt;
  if (g_begin < g_end) load(cur, g_begin);
  for (int g = g_begin; g < g_end; ++g) {
    load(nxt, min(g + 1, g_end - 1));          // (past the end: the last group again, never used)
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      s0 = mfma16(mla_widen<T, KV8>(cur.ka[s]), qf[s], s0);
      s1 = mfma16(mla_widen<T, KV8>(cur.kb[s]), qf[s], s1);
    }
    float* mine = &s_part[g & 1][wid][lane][0];
    *reinterpret_cast<f32x4*>(mine) = s0;
    *reinterpret_cast<f32x4*>(mine + 4) = s1;
    __syncthreads();
    // (every wave adds the four partials in the same order, its own included: the softmax state that follows is
    // then bit-identical in all four, which the split-KV merge relies on - only wave 0 reports m and l)
    s0 = *reinterpret_cast<const f32x4*>(&s_part[g & 1][0][lane][0]);
    s1 = *reinterpret_cast<const f32x4*>(&s_part[g & 1][0][lane][4]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      s0 += *reinterpret_cast<const f32x4*>(&s_part[g & 1][w][lane][0]);
      s1 += *reinterpret_cast<const f32x4*>(&s_part[g & 1][w][lane][4]);
    }
    // online softmax; key index of element j is base + 8 g4 + j
    const int kidx0 = g * 32 + 8 * g4;
    float p[8];
    float tmax = MLA_NEG_BIG;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float sv = (j < 4 ? s0[j & 3] : s1[j & 3]) * scale_log2e;
      p[j] = (kidx0 + j < seq_len) ? sv : MLA_NEG_BIG;
      tmax = fmaxf(tmax, p[j]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax);
    const float alpha = fast_exp2(m - m_new);
    float psum = 0.f;
    V pf;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float e = p[j] <= 0.5f * MLA_NEG_BIG ? 0.f : fast_exp2(p[j] - m_new);
      psum += e;
      pf[j] = (T)e;
    }
    l = l * alpha + psum;
    m = m_new;
#pragma unroll
    for (int i = 0; i < VT; ++i) {
      acc[i] *= alpha;
      acc[i] = mfma16(mla_widen<T, KV8>(cur.vf[i]), pf, acc[i]);
    }
    cur = nxt;
  }

  float lsum = l;
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (r16 >= nq) return;
  const int qh = qh0 + r16;
  if (num_splits == 1) {
    const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
    T* orow = out + t * out_stride + (long)qh * DV + 4 * g4;
#pragma unroll
    for (int i = 0; i < VT; ++i) {
      typename Vec4<T>::type o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (T)(acc[i][r] * inv);
      *reinterpret_cast<typename Vec4<T>::type*>(orow + 16 * (wid * VT + i)) = o;
    }
  } else {
    const long pidx = ((long)t * Hq + qh) * num_splits + split;
#pragma unroll
    for (int i = 0; i < VT; ++i)
      *reinterpret_cast<f32x4*>(part_o + pidx * DV + 16 * (wid * VT + i) + 4 * g4) = acc[i];
    if (wid == 0 && g4 == 0) { part_ml[2 * pidx] = m; part_ml[2 * pidx + 1] = lsum; }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Several query tiles per workgroup over ONE stream of cache fragments.  The kernel above requests every cached
// row once per 16 query rows; wherever more rows share a sequence, that is the L2's bandwidth, not HBM's:
//   * BY_TOKEN (prefill chunks, verification): the NQ tiles are NQ consecutive tokens of one sequence - the token
//     whose position is a multiple of NQ (or the first of its sequence in the step) leads a workgroup, the
//     workgroups of its followers exit at once; every tile masks with its own causal limit;
//   * otherwise (decode, many heads per rank): the NQ tiles are NQ head tiles of one token.
// Same split of the two reductions over the four waves; one wave per SIMD (the NQ accumulator sets and query
// fragments take the registers the second wave had), all NQ partial score tiles exchanged behind one barrier.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int DK, int DV, int NQ, bool BY_TOKEN, bool KV8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mla_attention_multi_kernel(
    T* __restrict__ out, long out_stride, float* __restrict__ part_o, float* __restrict__ part_ml,
    const T* __restrict__ q, long q_stride, const void* __restrict__ k_cache, const void* __restrict__ v_cache,
    const int* __restrict__ block_table, int bt_stride, const int* __restrict__ token_seq,
    const int* __restrict__ seq_lens, float scale_log2e, int T_, int Hq, int block_size, int num_splits) {
  using V = typename Vec8<T>::type;
  constexpr int KSTEPS = DK / 32;
  constexpr int KS = (KSTEPS + 3) / 4;
  constexpr int VT = DV / 64;
  // [buffer][wave][tile][half][lane][4]: a lane's f32x4 sits next to its neighbours' (conflict-free b128 accesses)
  __shared__ float s_part[2][4][NQ][2][64][4];

  const int split = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  // ---- this workgroup's tiles: (token, first head, visible keys); a tile with 0 keys is absent
  int tok[NQ], qh0[NQ], slen[NQ];
  // BY_TOKEN: leaders are (mostly) every NQ-th token, and consecutive workgroup ids go round the 8 XCDs: with
  // t0 = id they would all land on two of them.  Walk the tokens residue class by residue class instead
  // (ids 0 .. G-1 -> tokens 0, NQ, 2 NQ, ...; the next G ids -> 1, NQ + 1, ...).
  int t0 = blockIdx.z;
  if constexpr (BY_TOKEN) {
    const int G = (T_ + NQ - 1) / NQ;
    t0 = (int)(blockIdx.z % G) * NQ + (int)(blockIdx.z / G);
    if (t0 >= T_) return;
  }
  const int seq_row = token_seq != nullptr ? token_seq[t0] : t0;
  if constexpr (BY_TOKEN) {
    const int pos0 = seq_lens[t0] - 1;
    if (t0 > 0 && token_seq[t0 - 1] == seq_row && pos0 % NQ != 0) return;        // a follower: its leader covers it
    bool live = true;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int t = t0 + n;
      live = live && t < T_ && (n == 0 || (token_seq[t] == seq_row && (seq_lens[t] - 1) % NQ != 0));
      tok[n] = live ? t : t0;
      qh0[n] = blockIdx.y * 16;
      slen[n] = live ? seq_lens[t] : 0;
    }
  } else {
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      tok[n] = t0;
      qh0[n] = (blockIdx.y * NQ + n) * 16;
      slen[n] = qh0[n] < Hq ? seq_lens[t0] : 0;
    }
  }
  int seq_len = 0;
#pragma unroll
  for (int n = 0; n < NQ; ++n) seq_len = max(seq_len, slen[n]);
  const int* bt_row = block_table + (long)seq_row * bt_stride;

  const int n_groups = (seq_len + 31) >> 5;
  const int gps = (n_groups + num_splits - 1) / num_splits;
  const int g_begin = split * gps;
  const int g_end = min(g_begin + gps, n_groups);

  V qf[NQ][KS];
  int kstep_off[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) kstep_off[s] = 32 * min(wid * KS + s, KSTEPS - 1) + 8 * g4;
#pragma unroll
  for (int n = 0; n < NQ; ++n)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      V z;
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = (T)0.f;
      const bool have = wid * KS + s < KSTEPS && slen[n] > 0 && qh0[n] + r16 < Hq;
      qf[n][s] = have ? ld8<T>(q + tok[n] * q_stride + (long)(qh0[n] + r16) * DK + kstep_off[s]) : z;
    }
  f32x4 acc[NQ][VT];
  float m[NQ], l[NQ];
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    m[n] = MLA_NEG_BIG;
    l[n] = 0.f;
#pragma unroll
    for (int i = 0; i < VT; ++i) acc[n][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int slab = block_size * 16;
  using R = typename MlaRaw<T, KV8>::type;
  struct Frag { R ka[KS], kb[KS], vf[VT]; };
  auto load = [&](Frag& f, int g) {
    const int base = g * 32;
    const int key0 = base + 8 * (r16 >> 2) + (r16 & 3);
    const int ka = min(key0, seq_len - 1), kb = min(key0 + 4, seq_len - 1);
    const long ra = ((long)bt_row[ka / block_size] * block_size + (ka % block_size)) * DK;
    const long rb = ((long)bt_row[kb / block_size] * block_size + (kb % block_size)) * DK;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f.ka[s] = mla_ld<T, KV8>(k_cache, ra + kstep_off[s]);
      f.kb[s] = mla_ld<T, KV8>(k_cache, rb + kstep_off[s]);
    }
    const int vkey = min(base + 8 * g4, seq_len - 1) & ~7;
    const long vo = (long)bt_row[vkey / block_size] * (DV / 16) * slab + ((vkey % block_size) >> 3) * 128 + 8 * r16;
#pragma unroll
    for (int i = 0; i < VT; ++i) f.vf[i] = mla_ld<T, KV8>(v_cache, vo + (long)(wid * VT + i) * slab);
  };

  Frag cur, nxt;
  if (g_begin < g_end) load(cur, g_begin);
  for (int g = g_begin; g < g_end; ++g) {
    load(nxt, min(g + 1, g_end - 1));
    V ka[KS], kb[KS], vf[VT];                    // (an e4m3 cache: widened once, shared by the NQ tiles)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      ka[s] = mla_widen<T, KV8>(cur.ka[s]);
      kb[s] = mla_widen<T, KV8>(cur.kb[s]);
    }
#pragma unroll
    for (int i = 0; i < VT; ++i) vf[i] = mla_widen<T, KV8>(cur.vf[i]);
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        s0 = mfma16(ka[s], qf[n][s], s0);
        s1 = mfma16(kb[s], qf[n][s], s1);
      }
      *reinterpret_cast<f32x4*>(&s_part[g & 1][wid][n][0][lane][0]) = s0;
      *reinterpret_cast<f32x4*>(&s_part[g & 1][wid][n][1][lane][0]) = s1;
    }
    __syncthreads();
    const int kidx0 = g * 32 + 8 * g4;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      f32x4 s0 = *reinterpret_cast<const f32x4*>(&s_part[g & 1][0][n][0][lane][0]);
      f32x4 s1 = *reinterpret_cast<const f32x4*>(&s_part[g & 1][0][n][1][lane][0]);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        s0 += *reinterpret_cast<const f32x4*>(&s_part[g & 1][w][n][0][lane][0]);
        s1 += *reinterpret_cast<const f32x4*>(&s_part[g & 1][w][n][1][lane][0]);
      }
      float p[8];
      float tmax = MLA_NEG_BIG;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sv = (j < 4 ? s0[j & 3] : s1[j & 3]) * scale_log2e;
        p[j] = (kidx0 + j < slen[n]) ? sv : MLA_NEG_BIG;
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
        const float e = p[j] <= 0.5f * MLA_NEG_BIG ? 0.f : fast_exp2(p[j] - m_new);
        psum += e;
        pf[j] = (T)e;
      }
      l[n] = l[n] * alpha + psum;
      m[n] = m_new;
#pragma unroll
      for (int i = 0; i < VT; ++i) {
        acc[n][i] *= alpha;
        acc[n][i] = mfma16(vf[i], pf, acc[n][i]);
      }
    }
    cur = nxt;
  }

#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    float lsum = l[n];
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    const int qh = qh0[n] + r16;
    if (slen[n] <= 0 || qh >= Hq) continue;
    if (num_splits == 1) {
      const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
      T* orow = out + tok[n] * out_stride + (long)qh * DV + 4 * g4;
#pragma unroll
      for (int i = 0; i < VT; ++i) {
        typename Vec4<T>::type o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)(acc[n][i][r] * inv);
        *reinterpret_cast<typename Vec4<T>::type*>(orow + 16 * (wid * VT + i)) = o;
      }
    } else {
      const long pidx = ((long)tok[n] * Hq + qh) * num_splits + split;
#pragma unroll
      for (int i = 0; i < VT; ++i)
        *reinterpret_cast<f32x4*>(part_o + pidx * DV + 16 * (wid * VT + i) + 4 * g4) = acc[n][i];
      if (wid == 0 && g4 == 0) { part_ml[2 * pidx] = m[n]; part_ml[2 * pidx + 1] = lsum; }
    }
  }
}

// Merge of the KV splits: out = sum_s o_s 2^(m_s - M) / sum_s l_s 2^(m_s - M).
template <typename T, int DV>
__global__ void mla_reduce_kernel(T* __restrict__ out, long out_stride, const float* __restrict__ part_o,
                                  const float* __restrict__ part_ml, int Hq, int num_splits) {
  const int t = blockIdx.y, qh = blockIdx.x;
  const long p0 = ((long)t * Hq + qh) * num_splits;
  float mm = MLA_NEG_BIG;
  for (int s = 0; s < num_splits; ++s) mm = fmaxf(mm, part_ml[2 * (p0 + s)]);
  float lt = 0.f;
  for (int s = 0; s < num_splits; ++s) lt += part_ml[2 * (p0 + s) + 1] * fast_exp2(part_ml[2 * (p0 + s)] - mm);
  const float inv = lt > 0.f ? 1.f / lt : 0.f;
  for (int d = threadIdx.x; d < DV; d += blockDim.x) {
    float o = 0.f;
    for (int s = 0; s < num_splits; ++s) o += part_o[(p0 + s) * DV + d] * fast_exp2(part_ml[2 * (p0 + s)] - mm);
    out[t * out_stride + (long)qh * DV + d] = (T)(o * inv);
  }
}

// One token's latent row into both cache forms.  x[T][DK] (row stride in elements); slot < 0: skipped.
template <typename T, bool KV8>
__global__ void mla_cache_write_kernel(const T* __restrict__ x, long x_stride, void* __restrict__ k_cache,
                                       void* __restrict__ v_cache, const int* __restrict__ slot_mapping, int DK, int DV,
                                       int block_size) {
  const int t = blockIdx.x;
  const int slot = slot_mapping[t];
  if (slot < 0) return;
  const long blk = slot / block_size;
  const int off = slot % block_size;
  const T* row = x + t * x_stride;
  const long krow = ((long)blk * block_size + off) * DK;
  const long slab = (long)block_size * 16;
  const long vblk = blk * (DV / 16) * slab + (off >> 3) * 128 + (off & 7);
  for (int d = threadIdx.x; d < DK; d += blockDim.x) {
    const float v = (float)row[d];
    mla_st<T, KV8>(k_cache, krow + d, v);
    if (d < DV) mla_st<T, KV8>(v_cache, vblk + (d >> 4) * slab + (d & 15) * 8, v);
  }
}

// Everything between the projections and the attention of one token, in one launch (one workgroup per token):
//   * RMSNorm of the latent's first RANK values, rotation of its last ROPE values (half-apart pairs, fp32
//     [cos | sin] table row of the token's position), and the resulting row written into both cache forms;
//   * the absorbed query assembled per head: q_full[t][h] = [ q_lat[h][t] (RANK) | rotated q[t][h][nope:] (ROPE) ].
// Replaces a norm, two slice copies, the rotary kernel, two concatenations and the cache write.
template <typename T, bool KV8>
__global__ __launch_bounds__(256) void mla_prep_kernel(
    T* __restrict__ q_full, const T* __restrict__ q, long q_tok_stride, int q_head_stride, int nope,
    const T* __restrict__ q_lat, long q_lat_head_stride, long q_lat_tok_stride, const T* __restrict__ latent,
    long latent_stride, const T* __restrict__ norm_w, float eps, const float* __restrict__ cos_sin,
    const int* __restrict__ positions, void* __restrict__ k_cache, void* __restrict__ v_cache,
    const int* __restrict__ slot_mapping, int H, int RANK, int ROPE, int block_size) {
  __shared__ float red[17];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int DK = RANK + ROPE, half = ROPE / 2;
  const float* cs = cos_sin + (long)positions[t] * ROPE;
  const T* lat = latent + t * latent_stride;
  float ss = 0.f;
  for (int d = tid; d < RANK; d += 256) { const float v = (float)lat[d]; ss += v * v; }
  const float rinv = rsqrtf(block_sum(ss, red) / (float)RANK + eps);
  const int slot = slot_mapping != nullptr ? slot_mapping[t] : -1;
  if (slot >= 0) {
    const long blk = slot / block_size;
    const int off = slot % block_size;
    const long krow = (blk * block_size + off) * DK;
    const long slab = (long)block_size * 16;
    const long vblk = blk * (RANK / 16) * slab + (off >> 3) * 128 + (off & 7);
    for (int d = tid; d < RANK; d += 256) {
      const float v = (float)(T)((float)lat[d] * rinv * (float)norm_w[d]);      // rounded as the norm kernel rounds
      mla_st<T, KV8>(k_cache, krow + d, v);
      mla_st<T, KV8>(v_cache, vblk + (d >> 4) * slab + (d & 15) * 8, v);
    }
    for (int i = tid; i < half; i += 256) {
      const float x1 = (float)lat[RANK + i], x2 = (float)lat[RANK + half + i];
      mla_st<T, KV8>(k_cache, krow + RANK + i, x1 * cs[i] - x2 * cs[half + i]);
      mla_st<T, KV8>(k_cache, krow + RANK + half + i, x2 * cs[i] + x1 * cs[half + i]);
    }
  }
  T* qo = q_full + (long)t * H * DK;
  for (int e = tid; e < H * RANK; e += 256) {
    const int h = e / RANK, d = e - h * RANK;
    qo[(long)h * DK + d] = q_lat[h * q_lat_head_stride + t * q_lat_tok_stride + d];
  }
  for (int e = tid; e < H * half; e += 256) {
    const int h = e / half, i = e - h * half;
    const T* qr = q + t * q_tok_stride + (long)h * q_head_stride + nope;
    const float x1 = (float)qr[i], x2 = (float)qr[half + i];
    qo[(long)h * DK + RANK + i] = (T)(x1 * cs[i] - x2 * cs[half + i]);
    qo[(long)h * DK + RANK + half + i] = (T)(x2 * cs[i] + x1 * cs[half + i]);
  }
}

template <typename T, int DK, int DV, bool KV8>
int launch_mla(void* out, long out_stride, void* part_o, void* part_ml, const void* q, long q_stride,
               const void* k_cache, const void* v_cache, const int* block_table, int bt_stride, const int* token_seq,
               const int* seq_lens, float scale, int T_, int Hq, int block_size, int num_splits, int mode,
               hipStream_t st) {
  const int q_tiles = (Hq + 15) / 16;
  const float sl2 = scale * 1.4426950408889634f;
  constexpr int NQ = 4;
  if (mode == 1 && token_seq != nullptr) {          // NQ consecutive tokens of a sequence share the cache stream
    // (ids walk the tokens residue class by residue class: (T_ + NQ - 1) / NQ ids per class)
    mla_attention_multi_kernel<T, DK, DV, NQ, true, KV8><<<dim3(num_splits, q_tiles, (T_ + NQ - 1) / NQ * NQ), 256, 0, st>>>(
        (T*)out, out_stride, (float*)part_o, (float*)part_ml, (const T*)q, q_stride, k_cache, v_cache, block_table,
        bt_stride, token_seq, seq_lens, sl2, T_, Hq, block_size, num_splits);
  } else if (mode == 2) {                           // NQ head tiles of a token share it
    mla_attention_multi_kernel<T, DK, DV, NQ, false, KV8><<<dim3(num_splits, (q_tiles + NQ - 1) / NQ, T_), 256, 0, st>>>(
        (T*)out, out_stride, (float*)part_o, (float*)part_ml, (const T*)q, q_stride, k_cache, v_cache, block_table,
        bt_stride, token_seq, seq_lens, sl2, T_, Hq, block_size, num_splits);
  } else {
    mla_attention_kernel<T, DK, DV, KV8><<<dim3(num_splits, q_tiles, T_), 256, 0, st>>>(
        (T*)out, out_stride, (float*)part_o, (float*)part_ml, (const T*)q, q_stride, k_cache, v_cache, block_table,
        bt_stride, token_seq, seq_lens, sl2, Hq, block_size, num_splits);
  }
  HIP_LAUNCH_CHECK();
  if (num_splits > 1) {
    mla_reduce_kernel<T, DV><<<dim3(Hq, T_), 128, 0, st>>>((T*)out, out_stride, (const float*)part_o,
                                                          (const float*)part_ml, Hq, num_splits);
    HIP_LAUNCH_CHECK();
  }
  return 0;
}

template <typename T>
int dispatch_mla(int DK, int DV, int kv_fp8, void* out, long out_stride, void* part_o, void* part_ml, const void* q,
                 long q_stride, const void* k_cache, const void* v_cache, const int* block_table, int bt_stride,
                 const int* token_seq, const int* seq_lens, float scale, int T_, int Hq, int block_size,
                 int num_splits, int mode, hipStream_t st) {
#define MLA_CASE(DK_, DV_)                                                                                        \
  if (DK == DK_ && DV == DV_) {                                                                                   \
    if (kv_fp8)                                                                                                   \
      return launch_mla<T, DK_, DV_, true>(out, out_stride, part_o, part_ml, q, q_stride, k_cache, v_cache,       \
                                           block_table, bt_stride, token_seq, seq_lens, scale, T_, Hq, block_size, \
                                           num_splits, mode, st);                                                 \
    return launch_mla<T, DK_, DV_, false>(out, out_stride, part_o, part_ml, q, q_stride, k_cache, v_cache,        \
                                          block_table, bt_stride, token_seq, seq_lens, scale, T_, Hq, block_size, \
                                          num_splits, mode, st);                                                  \
  }
  MLA_CASE(576, 512) MLA_CASE(96, 64) MLA_CASE(320, 256)
#undef MLA_CASE
  return -4;
}

}  // namespace

// out[T][Hq][DV] = softmax(scale * q[T][Hq][DK] . K^T) . V over the latent cache.  token_seq[T] (may be null:
// token t uses block-table row t) and seq_lens[T] (keys 0 .. seq_lens[t] - 1 are visible to token t).
// part_o / part_ml: num_splits * T * Hq * (DV | 2) floats when num_splits > 1.
// mode 0: one workgroup per (token, 16 heads); 1: four consecutive tokens of a sequence per workgroup (needs
// token_seq; positions inside a sequence must be consecutive); 2: four head tiles of a token per workgroup.
// Supported (DK, DV): (576, 512) DeepSeek-V2 / V3, (320, 256), (96, 64).
VDAMD_API int vdamd_mla_attention(void* out, long out_stride, void* part_o, void* part_ml, const void* q,
                                  long q_stride, const void* k_cache, const void* v_cache, const int* block_table,
                                  int bt_stride, const int* token_seq, const int* seq_lens, float scale, int T_,
                                  int Hq, int DK, int DV, int block_size, int num_splits, int mode, int dtype,
                                  int kv_fp8, void* stream) {
  if (T_ <= 0) return 0;
  if (block_size % 8 || num_splits < 1 || num_splits > MLA_MAX_SPLITS) return -1;
  if (num_splits > 1 && (part_o == nullptr || part_ml == nullptr)) return -2;
  if (q_stride % 8 || out_stride % 4) return -5;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16)
    return dispatch_mla<bf16_t>(DK, DV, kv_fp8, out, out_stride, part_o, part_ml, q, q_stride, k_cache, v_cache,
                                block_table, bt_stride, token_seq, seq_lens, scale, T_, Hq, block_size, num_splits,
                                mode, st);
  if (dtype == DT_F16)
    return dispatch_mla<f16_t>(DK, DV, kv_fp8, out, out_stride, part_o, part_ml, q, q_stride, k_cache, v_cache,
                               block_table, bt_stride, token_seq, seq_lens, scale, T_, Hq, block_size, num_splits,
                               mode, st);
  return -3;
}

// kv_fp8: the caches hold OCP e4m3 bytes (values saturated at +-448, scale 1) instead of elements of the model type.
VDAMD_API int vdamd_mla_cache_write(const void* x, long x_stride, void* k_cache, void* v_cache,
                                    const int* slot_mapping, int T_, int DK, int DV, int block_size, int dtype,
                                    int kv_fp8, void* stream) {
  if (T_ <= 0) return 0;
  if (block_size % 8 || DV % 16 || DV > DK) return -1;
  hipStream_t st = (hipStream_t)stream;
#define MLA_CW(T__, KV8__)                                                                                   \
  mla_cache_write_kernel<T__, KV8__><<<T_, 256, 0, st>>>((const T__*)x, x_stride, k_cache, v_cache, slot_mapping, \
                                                         DK, DV, block_size)
  if (dtype == DT_BF16) { if (kv_fp8) MLA_CW(bf16_t, true); else MLA_CW(bf16_t, false); }
  else if (dtype == DT_F16) { if (kv_fp8) MLA_CW(f16_t, true); else MLA_CW(f16_t, false); }
  else return -3;
#undef MLA_CW
  HIP_LAUNCH_CHECK();
  return 0;
}

// q_full[T][H][RANK + ROPE] and the latent cache rows of T tokens (see mla_prep_kernel).  q[T][H][nope + ROPE]
// with the given token / head strides; q_lat[H][T][RANK] with the given strides; latent[T][RANK + ROPE];
// slot_mapping may be null (no cache bound: only q_full is produced).
VDAMD_API int vdamd_mla_prep(void* q_full, const void* q, long q_tok_stride, int q_head_stride, int nope,
                             const void* q_lat, long q_lat_head_stride, long q_lat_tok_stride, const void* latent,
                             long latent_stride, const void* norm_w, float eps, const float* cos_sin,
                             const int* positions, void* k_cache, void* v_cache, const int* slot_mapping, int T_,
                             int H, int RANK, int ROPE, int block_size, int dtype, int kv_fp8, void* stream) {
  if (T_ <= 0) return 0;
  if (RANK % 16 || ROPE % 2 || (slot_mapping != nullptr && block_size % 8)) return -1;
  hipStream_t st = (hipStream_t)stream;
#define MLA_PREP(T__, KV8__)                                                                                     \
  mla_prep_kernel<T__, KV8__><<<T_, 256, 0, st>>>((T__*)q_full, (const T__*)q, q_tok_stride, q_head_stride, nope,  \
                                                  (const T__*)q_lat, q_lat_head_stride, q_lat_tok_stride,         \
                                                  (const T__*)latent, latent_stride, (const T__*)norm_w, eps,     \
                                                  cos_sin, positions, k_cache, v_cache, slot_mapping, H, RANK,    \
                                                  ROPE, block_size)
  if (dtype == DT_BF16) { if (kv_fp8) MLA_PREP(bf16_t, true); else MLA_PREP(bf16_t, false); }
  else if (dtype == DT_F16) { if (kv_fp8) MLA_PREP(f16_t, true); else MLA_PREP(f16_t, false); }
  else return -3;
#undef MLA_PREP
  HIP_LAUNCH_CHECK();
  return 0;
}
